#!/usr/bin/env python3
"""Which fp32 rounding is the NMSE of proposed_algorithm sensitive to?  (CPU only, test infrastructure.)

The float64 port (oracle/cpu_port.cpp) is run on full-size trials (BASELINE configs[1] shape, torch builder on the CPU) with
JSTSP_PORT_ROUND masks that round ONE group of intermediate arrays to float32 where the HIP path holds them in fp32, and the
per-trial NMSE is compared with the pure float64 run.  tools/parity_tail.py measured rms |dNMSE| 3.7e-7 (max 1.95e-6 over
2560 trials) for EVERY HIP variant - split-f16 or strict fp32 MFMA, fused or three kernels - so the source is common to
all of them; this finds it.

    python tools/precision_study.py [--trials 16] [--masks 1,2,4,...]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {1: "X,K stored fp32", 2: "V1,V2 stored fp32", 4: "V,S stored fp32", 8: "Tc = K B^H output fp32", 16: "Res terms fp32 (A^H Tc, G_A V G_B, difference)",
         32: "R res fp32", 64: "Xs = A S B output fp32", 128: "G_A, G_B fp32", 256: "Y fp32", 512: "W = A S fp32",
         2048: "K B^H accumulation jitter", 4096: "G_A V G_B accumulation jitter", 8192: "A S B accumulation jitter", 16384: "G_A jitter (one-off)", 32768: "G_B jitter (one-off)",
         65536: "B to 22 bits in K B^H and (A S) B, G_B from exact B", 131072: "B to 22 bits everywhere (consistent)",
         262144: "G_B to 22 bits", 1048576: "rho, 1/rho, rho/(rho+1), tau/rho each rounded to fp32 on its own (TrialParams)", 4194304: "1/(Omega + 2 rho) stored fp32",
         8388608: "tau_S/rho, tau_Y/rho rounded to fp32", 16777216: "1/rho rounded to fp32", 33554432: "rho/(rho+1) rounded to fp32",
         67108864: "G_B alone stored fp32 (G_A float64)", 134217728: "svt operator I - Q as two f16 per entry (the pass's fragments)",
         2097152: "rho rounded to fp32, everything derived from it in float64 (a consistent problem with another rho)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=16)
    ap.add_argument("--masks", type=str, default="1,2,4,8,16,32,64,128,256,512,1023")
    ap.add_argument("--jitter", type=float, default=3e-7)
    ap.add_argument("--snr-db", type=float, default=5.0)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from jstsp19_amd.system_model import SweepParams
    from torch_builder import build_inputs, draw_trials
    from oracle import build_cpu_port as bp
    from oracle import solvers as O
    lib = bp.load()
    p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=a.snr_db)
    t0 = time.time()
    inp = build_inputs(p, draw_trials(p, list(range(a.trials)), device="cpu"))
    h = {k: inp[k].numpy() for k in ("subY", "Omega", "B")}
    A = inp["A"].numpy()
    zb = inp["Zbar"].numpy().astype(np.complex128)
    hyp = [inp[k].numpy() for k in ("tau_Y", "tau_Z", "rho")]
    print("inputs built in %.1f s" % (time.time() - t0), flush=True)

    def run(mask, jit=0.0):
        os.environ["JSTSP_PORT_ROUND"] = str(mask)
        os.environ["JSTSP_PORT_JITTER"] = repr(jit)
        S, _, _, _ = bp.proposed_algorithm(lib, h["subY"], h["Omega"], A, h["B"], 100, *hyp, want_ce=False, threads=a.threads)
        os.environ.pop("JSTSP_PORT_ROUND"); os.environ.pop("JSTSP_PORT_JITTER")
        return S, np.array([O.nmse_capped(S[t], zb[t]) for t in range(a.trials)])

    t0 = time.time()
    S0, n0 = run(0)
    print("float64 reference: %.1f s, mean NMSE %.4f" % (time.time() - t0, n0.mean()), flush=True)
    for m in [int(x, 0) for x in a.masks.split(",")]:
        S, n = run(m, a.jitter)
        d = n - n0
        ds = max(np.max(np.abs(S[t] - S0[t])) / np.max(np.abs(S0[t])) for t in range(a.trials))
        label = " + ".join(v for k, v in NAMES.items() if m & k)
        print("mask %5d  rms dNMSE %.2e  max %.2e  max rel dS %.2e   %s" % (m, np.sqrt(np.mean(d ** 2)), np.abs(d).max(), ds, label), flush=True)


if __name__ == "__main__":
    main()
