#!/usr/bin/env python3
"""PCIe-inclusive rate of the JSTSP_HOST path (numpy arrays in, numpy arrays out) at BASELINE
configs[1] — for DESIGN.md; never the headline value."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J
from jstsp19_amd.system_model import SweepParams
from bench import make_inputs
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
inp = make_inputs(p, list(range(batch)), torch.device("cuda:0"))
h = {k: inp[k].cpu().numpy() for k in ("subY", "Omega", "B")}
A = inp["A"].cpu().numpy()
for rep in range(2):
    t0 = time.perf_counter()
    S, Y, ce = J.proposed_algorithm(h["subY"], h["Omega"], A, h["B"], 100, inp["tau_Y"], inp["tau_Z"], inp["rho"], "approximate")
    dt = time.perf_counter() - t0
    print("host path: batch %d  %.3f s  %.1f estimates/s (includes numpy re-layout + H2D of %.2f GiB + D2H)" % (
        batch, dt, batch / dt, (h["B"].nbytes + h["subY"].nbytes + h["Omega"].nbytes) / 2**30))
