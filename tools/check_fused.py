#!/usr/bin/env python3
"""JSTSP_FUSED=1 (one pass over the dictionary per iteration) against the default three-kernel path on the same trials at
the BASELINE configs[1] shape: S, Y, convergence_error after a few iterations, then timing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J
from jstsp19_amd.system_model import SweepParams, build_trials

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3, 10]
p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
inp = build_trials(p, 0, batch, seed=1)
args = lambda I: (inp["subY"], inp["Omega"], inp["A"], inp["B"], I, inp["tau_Y"].numpy(), inp["tau_Z"].numpy(), inp["rho"].numpy(),
                  "approximate")
def run(I, fused):
    os.environ["JSTSP_FUSED"] = "1" if fused else "0"
    S, Y, ce = J.proposed_algorithm(*args(I))
    torch.cuda.synchronize()
    return S.cpu().numpy(), Y.cpu().numpy(), ce.cpu().numpy()
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))
for I in iters:
    S0, Y0, c0 = run(I, False)
    S1, Y1, c1 = run(I, True)
    fin = np.isfinite(c0) & np.isfinite(c1)
    print("Imax %3d: rel dS %.3e  rel dY %.3e  max rel dce %.3e  nan(S) %d" %
          (I, rel(S1, S0), rel(Y1, Y0), float(np.max(np.abs(c1[fin] - c0[fin]) / np.maximum(np.abs(c0[fin]), 1e-30))),
           int(np.isnan(S1).sum())), flush=True)
if len(sys.argv) > 3:
    I = int(sys.argv[3])
    for fused in (False, True):
        run(I, fused)
        t0 = time.perf_counter(); run(I, fused); dt = time.perf_counter() - t0
        print("fused=%d: %d trials x %d iterations in %.3f s  (%.3f ms / iteration)" % (fused, batch, I, dt, 1e3 * dt / I))
