#!/usr/bin/env python3
"""Round 6, VERDICT item 8: are the Jacobi kernels of eig3.hip / eig_large.hip safe to compile WITH packed fp32 (v_pk_fma_f32 ...)?

The whole library is built with `-target-feature -packed-fp32-ops` because v_pk_fma_f32 with op_sel returned run-to-run different
bits beside MFMA-heavy waves (tools/probe/pk_fp32_probe.hip, build.py).  That was observed in the Lanczos kernel; the order-65..128
Jacobi (jacobi128_kernel, VALU-issue-bound, 85 % of mc_admm) never runs beside an MFMA wave of its OWN solver - but another
context of the same process may.  This probe runs svt and mc_admm x 20 at 128 x 128 x 1024 twenty times while a second context keeps
the matrix pipe busy (a queue of big split-f16 correlate launches), compares every run's bits with the first, and times mc_admm
alone.  Usage:  JSTSP_PK_FP32_FILES=eig3.hip,eig_large.hip python jstsp19_amd/build.py; python tools/probe/pk_jacobi_race.py
"""
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
Hc, OHc, Omc = cm(H), cm(Om * OH), cm(Om)
tau, rho = np.full(batch, 0.5), np.full(batch, 0.1)

# the MFMA load: K B^H at the configs[1] shape on a SECOND context (own stream), 32 trials per launch, launched ahead
ctx2 = J.Context(0)
Kb = cm(rnd(32, 64, 4096)); Bb = cm(rnd(32, 512, 4096)); Ab = cm(rnd(64, 64))
torch.cuda.synchronize()


def mfma_load(k):
    for _ in range(k):
        J.correlate(Kb, Ab, Bb, ctx=ctx2)


def run():
    X = J.svt(OHc, np.full(batch, 0.5))
    Xa, ce = J.mc_admm(Hc, OHc, Omc, 20, tau, rho)
    torch.cuda.synchronize()
    return X.cpu().numpy().tobytes(), Xa.cpu().numpy().tobytes(), ce.cpu().numpy().tobytes()


ref = run()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    J.mc_admm(Hc, OHc, Omc, 20, tau, rho); torch.cuda.synchronize()
    print("mc_admm x20 batch %d alone: %.3f s" % (batch, time.perf_counter() - t0), flush=True)
bad = 0
for rep in range(20):
    mfma_load(60)                       # queued on ctx2's stream: runs beside the solver below
    r = run()
    ctx2.synchronize()
    d = [a != b for a, b in zip(r, ref)]
    if any(d):
        bad += 1
        print("run %d differs from the first: svt %s, mc_admm X %s, ce %s" % (rep, *d), flush=True)
print("runs that differ beside the MFMA load: %d of 20" % bad)
