#!/usr/bin/env python3
"""Warm-started lambda_max at order 128 (sparse_admm's error curve, configs[2]): how many warm attempts converge and in how many steps."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
n, batch, Imax = 128, 1024, 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1283)
rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device=dev), torch.randn(*s, generator=g, device=dev))
idx = torch.arange(n, device=dev, dtype=torch.float64)
D = (torch.exp(-2j * np.pi * idx[:, None] * idx[None, :] / n) / np.sqrt(n)).to(torch.complex64)
Sp = torch.zeros(batch, n, n, dtype=torch.complex64, device=dev)
Sp[:, ::17, ::13] = rnd(batch, len(range(0, n, 17)), len(range(0, n, 13)))
H = D @ Sp @ D.conj().T
OH = H + 0.05 * rnd(batch, n, n)
cm = J.colmajor
ctx = J.default_context(0)
J.sparse_admm(cm(H), cm(OH), cm(D), cm(D), Imax); torch.cuda.synchronize()
c = (C.c_uint * 4)()
fn = ctx._lib.jstsp_debug_lanczos_counters
fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]; fn.restype = C.c_int
fn(ctx.handle, c)
tot = batch * Imax
print("order 128: %d lambda_max calls; warm attempts failed %d, verifications %d (mismatches %d), warm steps total %d = %.2f per call"
      % (tot, c[1], c[2], c[0], c[3], c[3] / tot))
