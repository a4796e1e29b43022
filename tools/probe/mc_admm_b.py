#!/usr/bin/env python3
"""mc_admm x 20 at the configs[2] shape, 1024 trials (for the kernel table)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
batch, n = 1024, 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device=dev), torch.randn(*s, generator=g, device=dev))
idx = torch.arange(n, device=dev, dtype=torch.float64)
D = (torch.exp(-2j * np.pi * idx[:, None] * idx[None, :] / n) / np.sqrt(n)).to(torch.complex64)
Sp = torch.zeros(batch, n, n, dtype=torch.complex64, device=dev)
Sp[:, ::17, ::13] = rnd(batch, len(range(0, n, 17)), len(range(0, n, 13)))
H = D @ Sp @ D.conj().T
OH = H + 0.05 * rnd(batch, n, n)
Om = (torch.rand(batch, n, n, generator=g, device=dev) < 0.125).float()
cm = J.colmajor
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    J.mc_admm(cm(H), cm(Om * OH), cm(Om), 20, np.full(batch, 0.5), np.full(batch, 0.1)); torch.cuda.synchronize()
    print("mc_admm x20 batch %d: %.3f s" % (batch, time.perf_counter() - t0))
