#!/usr/bin/env python3
"""OMP step kernel forms (JSTSP_OMP_REG=0/1, read at first use): hashes of the outputs on dense and Kronecker dictionaries."""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
rng = np.random.default_rng(3)
c = lambda *s: (rng.standard_normal(s) + 1j * rng.standard_normal(s)).astype(np.complex64)
dev = torch.device("cuda:0")
h = lambda *a: hashlib.sha1(b"".join(x.cpu().numpy().tobytes() for x in a)).hexdigest()[:12]
out = []
save = {}
for meas, size_d, m, batch in ((1024, 1024, 24, 1), (1024, 1024, 24, 8), (1536, 700, 20, 3), (300, 512, 30, 5), (64, 128, 64, 2)):
    A = (c(meas, size_d) / np.sqrt(meas)).astype(np.complex64)
    x0 = np.zeros((batch, size_d), np.complex64)
    for t in range(batch):
        x0[t, rng.choice(size_d, 6, replace=False)] = c(6)
    y = (x0 @ A.T + 0.01 * c(batch, meas)).astype(np.complex64)
    x, idx, _, tg = J.OMP(J.colmajor(torch.from_numpy(A).to(dev)), torch.from_numpy(y).to(dev), m)
    out.append("dense %4d x %4d m %2d batch %d: x %s idx %s tg %s" % (meas, size_d, m, batch, h(x), h(idx), h(tg)))
    save["x%d" % len(out)] = x.cpu().numpy(); save["i%d" % len(out)] = idx.cpu().numpy(); save["t%d" % len(out)] = tg.cpu().numpy()
for N, M, Gr, G2, m, batch in ((16, 64, 16, 64, 24, 1), (8, 200, 12, 40, 12, 4)):
    Af, Bf = (c(N, Gr) / 4).astype(np.complex64), (c(G2, M) / 8).astype(np.complex64)
    S = np.zeros((batch, Gr, G2), np.complex64)
    for t in range(batch):
        S[t].reshape(-1)[rng.choice(Gr * G2, 5, replace=False)] = c(5)
    Y = np.stack([Af @ S[t] @ Bf for t in range(batch)]).astype(np.complex64)
    yv = np.stack([Y[t].reshape(-1, order="F") for t in range(batch)])
    r = J.omp_kron(J.colmajor(torch.from_numpy(Af).to(dev)), J.colmajor(torch.from_numpy(Bf).to(dev)), torch.from_numpy(yv).to(dev), m)
    out.append("kron N %d M %d m %d batch %d: %s" % (N, M, m, batch, h(*[z for z in r if torch.is_tensor(z)])))
print("\n".join(out))
if len(sys.argv) > 1:
    np.savez(sys.argv[1], **save)
