#!/usr/bin/env python3
"""BASELINE configs[4] shape for proposed_algorithm_angles: Nt=256, Nr=64, L=16, T=256 ->
N=64, M=65536, Gr=64, G2=4096 (B is 2 GiB per pilot set: pilots shared by the batch here).
Functional run with invariants (the float64 oracle would need ~30 TFLOP per trial on the host)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, M, Gr, G2 = 64, 65536, 64, 4096
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device=dev), torch.randn(*s, generator=g, device=dev))
cm = J.colmajor
A = cm(rnd(N, Gr) / np.sqrt(N))
B = cm(rnd(G2, M) / np.sqrt(G2))                      # shared pilots
S0 = torch.zeros(batch, Gr, G2, dtype=torch.complex64, device=dev)
idx = torch.randint(0, Gr * G2, (batch, 40), generator=g, device=dev)
S0.view(batch, -1).scatter_(1, idx, rnd(batch, 40))
Om = (torch.rand(batch, N, M, generator=g, device=dev) < 0.125).float()
X = torch.empty(batch, N, M, dtype=torch.complex64, device=dev)
for t in range(batch):
    X[t] = A @ S0[t] @ B
subY = Om * (X + 0.05 * rnd(batch, N, M))
del X
indx = (torch.argsort(S0.transpose(1, 2).reshape(batch, -1).abs(), dim=1, descending=True, stable=True) + 1).to(torch.int32)
fro2 = (subY.abs() ** 2).sum(dim=(1, 2)).double().cpu().numpy()
tY = 1.0 / fro2; tS = np.full(batch, 1e-3); rho = np.full(batch, 0.2)
t0 = time.perf_counter()
S, Y, ce = J.proposed_algorithm_angles(cm(subY), cm(Om), indx, A, B, 20, tY, tS, rho, "approximate", None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("config-5 shape, batch %d, 20 iterations: %.2f s; workspace %.1f GiB" % (batch, dt, J.default_context(0).workspace_bytes() / 2**30))
assert torch.isfinite(torch.view_as_real(S)).all() and torch.isfinite(torch.view_as_real(Y)).all()
ce = ce.cpu().numpy(); assert np.all(np.isinf(ce[:, 0, 2])) and np.all(np.isfinite(ce[:, 1:, :]))
for t in range(batch):
    allowed = set((indx[t, :10 + 5 * 20] - 1).cpu().numpy().tolist())
    nz = set(np.flatnonzero(S[t].cpu().numpy().reshape(-1, order="F")).tolist())
    assert nz <= allowed
S1, _, _ = J.proposed_algorithm_angles(cm(subY[:1]), cm(Om[:1]), indx[:1], A, B, 20, tY[:1], tS[:1], rho[:1], "approximate", None, want_ce=False)
print("batched vs single rel diff: %.2e" % float((S1[0] - S[0]).abs().max() / S[0].abs().max()))
err = float((S - S0).abs().max() / S0.abs().max())
print("recovery: max |S - S0| / max |S0| after 20 iterations = %.3f ; ce(20,:) = %s" % (err, ce[0, -1]))

# ---- the same solve with pilots as the reference's drivers build them: a block-Toeplitz dictionary (proposed_hbf.m:36-42: block ld is
# block 0 delayed by ld columns), block height Gt = 256, L = 16.  G_B = B B^H is then assembled from its first block row (1 / 16 of the
# product); the iteration is the same three-kernel one.
B0 = rnd(256, M) / np.sqrt(G2)
Bt = torch.empty(G2, M, dtype=torch.complex64, device=dev)
for ld in range(16):
    Bt[256 * ld:256 * (ld + 1), ld:] = B0[:, :M - ld]
    if ld:
        Bt[256 * ld:256 * (ld + 1), :ld] = rnd(256, ld) / np.sqrt(G2)
Bt = cm(Bt)
del B0
J.proposed_algorithm_angles(cm(subY[:1]), cm(Om[:1]), indx[:1], A, Bt, 2, tY[:1], tS[:1], rho[:1], "approximate", None, want_ce=False)   # (workspace growth)
torch.cuda.synchronize(); t0 = time.perf_counter()
St, _, _ = J.proposed_algorithm_angles(cm(subY), cm(Om), indx, A, Bt, 20, tY, tS, rho, "approximate", None)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("config-5 shape with block-Toeplitz pilots (block height %d found), batch %d, 20 iterations: %.2f s, finite %s" % (
    J.default_context(0).last_dictionary_block(), batch, dt, bool(torch.isfinite(torch.view_as_real(St)).all())))
del Bt, St

# ---- VAMP at the same shape (the drivers' call: vamp(vec(Y_hbf*B_hbf'), kron((B_hbf*B_hbf').', A_hbf), 1, L), plot_errorVSsnr.m:79-80,100):
# Gb = B_hbf B_hbf' has order G2 = 4096 - eigen-decomposition through csrc/eig_large.hip - and is shared by the batch here.
del subY, Om, B, S, Y
torch.cuda.empty_cache()
T_hbf = 8192                                           # round(T / (Nr / Mr)) * Nt = 32 * 256
Bh = cm(rnd(G2, T_hbf) / np.sqrt(T_hbf))
Gb = cm(Bh @ Bh.conj().T)
Gb = cm(0.5 * (Gb + Gb.conj().T))
del Bh
Xs = torch.zeros(batch, Gr, G2, dtype=torch.complex64, device=dev)
Xs.view(batch, -1).scatter_(1, idx, 3 * rnd(batch, 40))
Yv = torch.empty(batch, N, G2, dtype=torch.complex64, device=dev)
for t in range(batch):
    Yv[t] = A @ Xs[t] @ Gb + 0.05 * rnd(N, G2)
for nit in (3, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Xv = J.vamp_kron(cm(Yv), A, Gb, 1.0, 40, nit=nit)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(torch.view_as_real(Xv)).all())
    e = float((Xv - Xs).abs().max() / Xs.abs().max())
    print("vamp_kron Na=64 Gr=64 G2=4096, batch %d, %d iterations: %.2f s, finite %s, max |X - X0| / max |X0| = %.3f" % (batch, nit, dt, ok, e))
