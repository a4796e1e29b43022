#!/usr/bin/env python3
"""One device-resident proposed_algorithm call at the BASELINE configs[1] shape (for rocprofv3): args batch Imax [reps] [native].
`native`: the reference's own driver shape (plot_errorVSsnr.m:8-23: Nt = 4, Nr = 32, L = 4, T = 35, Mr = 4)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J
from jstsp19_amd.system_model import SweepParams, build_trials
batch, Imax = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
p = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4, snr_db=5.0) if "native" in sys.argv[4:] else SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
inp = build_trials(p, 0, batch, seed=1)
for r in range(reps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S, Y, ce = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], Imax, inp["tau_Y"].numpy(), inp["tau_Z"].numpy(),
                                    inp["rho"].numpy(), "approximate")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("run %d: %.3f s, %.3f ms/iteration, nan %d" % (r, dt, 1e3 * dt / Imax, int(torch.isnan(S.abs()).sum())))
