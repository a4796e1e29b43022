#!/usr/bin/env python3
"""BASELINE configs[2] shapes (128 x 128): sparse_admm + svt / mc_svt, batched, device-resident timing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device=dev), torch.randn(*s, generator=g, device=dev))
idx = torch.arange(n, device=dev, dtype=torch.float64)
D = torch.exp(-2j * np.pi * idx[:, None] * idx[None, :] / n).to(torch.complex64) / np.sqrt(n)
Sp = torch.zeros(batch, n, n, dtype=torch.complex64, device=dev)
Sp[:, ::17, ::13] = rnd(batch, len(range(0, n, 17)), len(range(0, n, 13)))
H = D @ Sp @ D.conj().T
OH = H + 0.05 * rnd(batch, n, n)
Om = (torch.rand(batch, n, n, generator=g, device=dev) < 0.125).float()
cm = J.colmajor
for name, fn in [("svt", lambda: J.svt(cm(OH), np.full(batch, 1.0))),
                 ("mc_svt x20", lambda: J.mc_svt(cm(Om * OH), cm(Om), 20, np.full(batch, 0.5), np.full(batch, 0.1))),
                 ("mc_admm x20", lambda: J.mc_admm(cm(H), cm(Om * OH), cm(Om), 20, np.full(batch, 0.5), np.full(batch, 0.1))),
                 ("sparse_admm x100", lambda: J.sparse_admm(cm(H), cm(OH), cm(D), cm(D), 100))]:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-18s batch %d: %.3f s  (%.1f problems/s)" % (name, batch, dt, batch / dt))
