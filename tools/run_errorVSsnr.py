#!/usr/bin/env python3
"""The plot_errorVSsnr.m sweep (reference-native parameters, :8-25) on the HIP path: proposed_algorithm,
proposed_algorithm_angles, LS and VAMP baselines; prints the mean capped NMSE per SNR point."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jstsp19_amd.montecarlo import run_sweep
from jstsp19_amd.system_model import SweepParams

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=64)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--builder", default="hip", choices=["hip", "torch"])
ap.add_argument("--config3", action="store_true", help="BASELINE configs[3]: Nt=Nr=64, Nrf=8, K=64, L=8, 10 SNR points")
a = ap.parse_args()
base = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4)              # plot_errorVSsnr.m:8-23
snrs = list(range(-15, 16, 3))                                  # :24
if a.config3:
    base = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8)           # the configs[1] shape (N=64, M=4096, Gr=64, G2=512)
    snrs = list(range(-15, 15, 3))                              # 10 points
t0 = time.perf_counter()
out = run_sweep(base, snrs, a.trials, Imax=100, batch=a.batch, baselines=True, numOfnz=100, builder=a.builder)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("SNR(dB)  proposed  proposed+angles  LS        VAMP      (%d trials/point, %s input builder, %.1f s)" % (a.trials, a.builder, dt))
for s, row in zip(snrs, out.tolist()):
    print("%6d   %.5f   %.5f          %.5f   %.5f" % (s, *row))
