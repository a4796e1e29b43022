#!/usr/bin/env python3
"""Diagnostics of the warm-started lambda_max kernel (eig2.hip) at the bench workload: how many warm attempts converge, how many
steps they take, and what the kernel costs alone (a sequence of slowly rotating 64 x 64 Grams through
jstsp_lambda_max_sequence_c32, warm against cold).   python tools/probe/lanczos_stats.py [batch]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import jstsp19_amd as J
from jstsp19_amd.system_model import SweepParams
from bench import make_inputs

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
inp = make_inputs(p, list(range(batch)), torch.device("cuda:0"))
ctx = J.default_context(0)
for imax in (10, 100):
    J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], imax, inp["tau_Y"], inp["tau_Z"], inp["rho"], "approximate")
    torch.cuda.synchronize()
    c = (C.c_uint * 4)()
    fn = ctx._lib.jstsp_debug_lanczos_counters
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    fn.restype = C.c_int
    fn(ctx.handle, c)
    tot = 3 * batch * imax
    conv = tot - 3 * batch - c[1]          # (the first call of every matrix is cold)
    print("Imax %d: %d lambda_max calls; warm attempts failed %d, verifications %d (mismatches %d), steps per converged warm attempt %.2f"
          % (imax, tot, c[1], c[2], c[0], c[3] / max(1, conv)))

# the kernel alone
rng = np.random.default_rng(0)
n, nb, steps = 64, 768, 60
X = (rng.standard_normal((nb, n, 256)) + 1j * rng.standard_normal((nb, n, 256))).astype(np.complex64)
X[:, :, :6] *= 8.0                                     # a few strong directions, as the iterates have
D = (rng.standard_normal((nb, n, 256)) + 1j * rng.standard_normal((nb, n, 256))).astype(np.complex64)
G = np.empty((steps, nb, n, n), np.complex64)
for s in range(steps):
    Y = X + 0.01 * s * D
    G[s] = Y @ np.conj(np.swapaxes(Y, 1, 2))
for env in ({"JSTSP_LANCZOS_WARM": "0"}, {}, {"JSTSP_LANCZOS_VERIFY": "0"}):
    os.environ.update(env)
    J.lambda_max_sequence(G[:2])
    t0 = time.perf_counter()
    lam = J.lambda_max_sequence(G)
    dt = time.perf_counter() - t0
    for k in env:
        os.environ.pop(k)
    ref = np.linalg.eigvalsh(G[-1].astype(np.complex128))[:, -1]
    print(env or "default", "%.1f ms for %d x %d matrices (host arrays: includes the upload)" % (dt * 1e3, steps, nb),
          "max rel err last step %.2e" % np.max(np.abs(lam[-1] - ref) / ref))
