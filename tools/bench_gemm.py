#!/usr/bin/env python3
"""Micro-benchmark of the two dominant GEMMs (correlate K*B^H, synthesize W*B) at the
BASELINE configs[1] shape, device-resident, timed with HIP events inside the library."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
N, M, Gr, G2 = 64, 4096, 64, 512
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device=dev), torch.randn(*s, generator=g, device=dev))
K = J.colmajor(rnd(a.batch, N, M)); S = J.colmajor(rnd(a.batch, Gr, G2))
A = J.colmajor(rnd(N, Gr)); B = J.colmajor(rnd(a.batch, G2, M))
ctx = J.default_context(0)
for _ in range(2):
    J.correlate(K, A, B); J.synthesize(S, A, B)
torch.cuda.synchronize()
ctx.set_profiling(True)
for _ in range(a.reps):
    J.correlate(K, A, B); J.synthesize(S, A, B)
torch.cuda.synchronize()
fl = 8.0 * N * M * G2 * a.batch
for name in ("correlate", "synthesize"):
    n, ms = ctx.get_profile(name)
    print("%-10s %3d launches  %.3f ms avg  %.1f TFLOP/s (%.1f%% of 157.3)" % (name, n, ms / n, fl / (ms / n * 1e-3) / 1e12,
                                                                           fl / (ms / n * 1e-3) / 1e12 / 1.573))

# accuracy of both GEMM paths against float64 on 4 problems
def err(x, ref):
    return float((x.to(torch.complex128) - ref).abs().max() / ref.abs().max())
nchk = min(4, a.batch)
Kd, Sd, Ad, Bd = (x.to(torch.complex128) for x in (K[:nchk], S[:nchk], A, B[:nchk]))
ref_c = Ad.conj().T @ Kd @ Bd.conj().transpose(1, 2)
ref_s = Ad @ Sd @ Bd
for h2 in ("0", "1"):
    os.environ["JSTSP_H2"] = h2
    c = J.correlate(K[:nchk].contiguous() if False else J.colmajor(K[:nchk]), A, J.colmajor(B[:nchk]))
    s = J.synthesize(J.colmajor(S[:nchk]), A, J.colmajor(B[:nchk]))
    torch.cuda.synchronize()
    print("JSTSP_H2=%s  correlate max rel err %.2e   synthesize max rel err %.2e" % (h2, err(c, ref_c), err(s, ref_s)))
