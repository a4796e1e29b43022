#!/usr/bin/env python3
"""One order-n Hermitian eigen-decomposition through vamp_kron's setup (csrc/eig_large.hip), timed."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device=dev), torch.randn(*s, generator=g, device=dev))
cm = J.colmajor
A = cm(rnd(64, 64) / 8)
Bh = cm(rnd(n, 2 * n) / np.sqrt(2 * n))
Gb = cm(Bh @ Bh.conj().T); Gb = cm(0.5 * (Gb + Gb.conj().T)); del Bh
Xs = torch.zeros(1, 64, n, dtype=torch.complex64, device=dev)
idx = torch.randint(0, 64 * n, (1, 40), generator=g, device=dev)
Xs.view(1, -1).scatter_(1, idx, 3 * rnd(1, 40))
Yv = (A @ Xs[0] @ Gb + 0.05 * rnd(64, n))[None]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Xv = J.vamp_kron(cm(Yv), A, Gb, 1.0, 40, nit=3)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("order %d: vamp_kron 3 iterations %.3f s, max |X - X0| / max |X0| = %.3f" % (n, dt, float((Xv - Xs).abs().max() / Xs.abs().max())), flush=True)
