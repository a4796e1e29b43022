#!/usr/bin/env python3
"""configs[0] OMP at one problem, 20 calls (for the kernel sequence)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
rng = np.random.default_rng(16)
c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
Phi = np.kron((c(64, 64) / 8).T, c(16, 16) / 4).astype(np.complex64)
dev = torch.device("cuda:0")
Phi_d = J.colmajor(torch.from_numpy(Phi).to(dev))
x0 = np.zeros((1, 1024), np.complex64); x0[0, rng.choice(1024, 6, replace=False)] = c(6)
y_d = torch.from_numpy((x0 @ Phi.T + 0.01 * c(1, 1024)).astype(np.complex64)).to(dev)
for rep in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    J.OMP(Phi_d, y_d, 24, want_target=False); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("%.3f ms" % (dt * 1e3))
