import numpy as np, sys
sys.path.insert(0, '/root/repo')
import jstsp19_amd as J
rng = np.random.default_rng(1)
c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
N = Gr = 64; G2 = 512; batch = 16
A = (c(N, Gr) / 8).astype(np.complex64)
GA = (A.astype(np.complex128).conj().T @ A.astype(np.complex128)).astype(np.complex64)
# a "coherent" case: Tc = A X so that A^H Tc = G_A X has sums of mostly positive terms on the diagonal part
X = c(batch, Gr, G2)
Tc = np.einsum("na,tag->tng", A.astype(np.complex128), X).astype(np.complex64)
ref = np.einsum("na,tng->tag", A.astype(np.complex128).conj(), Tc.astype(np.complex128))
Res, P1 = J.gradient_head(Tc, A, GA, None)
Res = np.asarray(Res).astype(np.complex128)
rel = ((Res - ref) * ref.conj()).real / np.abs(ref) ** 2
print("f16 head : signed mean rel err %.3e  rms %.3e" % (rel.mean(), np.sqrt(np.mean(np.abs(Res - ref) ** 2 / np.abs(ref) ** 2))))
# fp32 rounding of the exact result for comparison
r32 = ref.astype(np.complex64).astype(np.complex128)
rel = ((r32 - ref) * ref.conj()).real / np.abs(ref) ** 2
print("fp32 round: signed mean rel err %.3e  rms %.3e" % (rel.mean(), np.sqrt(np.mean(np.abs(r32 - ref) ** 2 / np.abs(ref) ** 2))))
# the library's fp32-MFMA product for the same thing: correlate with B = I?  use synthesize-like call: jstsp_correlate needs B; skip
import torch
# P1 check (second product)
p1ref = np.einsum("ab,tbg->tag", GA.astype(np.complex128), Res)
P1 = np.asarray(P1).astype(np.complex128)
rel = ((P1 - p1ref) * p1ref.conj()).real / np.abs(p1ref) ** 2
print("f16 P1   : signed mean rel err %.3e  rms %.3e" % (rel.mean(), np.sqrt(np.mean(np.abs(P1 - p1ref) ** 2 / np.abs(p1ref) ** 2))))
# the fp32-MFMA chain for the same product: correlate(K = Tc, A, B = I) = A^H Tc I
I = np.eye(G2, dtype=np.complex64)
out = np.asarray(J.correlate(Tc, A, I)).astype(np.complex128)
rel = ((out - ref) * ref.conj()).real / np.abs(ref) ** 2
print("fp32 cgemm: signed mean rel err %.3e  rms %.3e" % (rel.mean(), np.sqrt(np.mean(np.abs(out - ref) ** 2 / np.abs(ref) ** 2))))
# operands that ARE f16 numbers (no split error at all): what the f16 MFMA accumulation itself does
A16 = (np.round(c(N, Gr).real * 64) / 64 + 1j * np.round(c(N, Gr).imag * 64) / 64).astype(np.complex64)
T16 = (np.round(c(batch, N, G2).real * 64) / 64 + 1j * np.round(c(batch, N, G2).imag * 64) / 64).astype(np.complex64)
ref16 = np.einsum("na,tng->tag", A16.astype(np.complex128).conj(), T16.astype(np.complex128))
R16, _ = J.gradient_head(T16, A16, GA, None)
R16 = np.asarray(R16).astype(np.complex128)
print("f16-exact operands: rms rel err %.3e, exact entries %.3f" % (np.sqrt(np.mean(np.abs(R16 - ref16) ** 2 / np.maximum(np.abs(ref16), 1e-30) ** 2)),
      np.mean(R16 == ref16.astype(np.complex64).astype(np.complex128))))
