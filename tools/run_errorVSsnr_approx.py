#!/usr/bin/env python3
"""The plot_errorVSsnr_approx.m sweep (reference-native parameters, :8-20) on the HIP path: Algorithm 1
(proposed_algorithm 'std') against Algorithm 2 ('approximate') for Imax in {10, 30, 50}, inputs from
wideband_hybBF_comm_system_training, S = pinv(A)*Y*pinv(B); prints the mean capped NMSE per (Imax, SNR)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jstsp19_amd.montecarlo import run_approx_sweep
from jstsp19_amd.system_model import TrainingParams

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=50)              # maxMCRealizations, :17
ap.add_argument("--batch", type=int, default=50)
a = ap.parse_args()
base = TrainingParams(Nt=4, Nr=32, L=4, T=70, ratio=0.75)       # :8-19
snrs = list(range(-15, 16, 5))                                  # :15
imax = [10, 30, 50]                                             # :19
t0 = time.perf_counter()
out = run_approx_sweep(base, snrs, imax, a.trials, batch=a.batch)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("capped NMSE, %d trials/point, %.1f s" % (a.trials, dt))
print("Imax  SNR(dB)  Algorithm 1 (std)  Algorithm 2 (approximate)")
for i, im in enumerate(imax):
    for s, row in zip(snrs, out[i].tolist()):
        print("%4d  %6d   %.6f           %.6f" % (im, s, row[0], row[1]))
