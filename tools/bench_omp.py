#!/usr/bin/env python3
"""OMP.m on the Kronecker dictionary kron(B.', A) at the BASELINE configs[1] shape (N=64, M=4096, Gr=64, G2=512,
per-trial pilots), batched and device-resident: every OMP iteration is one pass of the correlation kernel."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J
from jstsp19_amd.system_model import SweepParams, build_trials
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 48
p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
inp = build_trials(p, 0, batch)
y = inp["subY"].transpose(1, 2).reshape(batch, -1)              # column-major vec
for h2, gram in (("0", "0"), ("1", "0"), ("1", "1")):
    os.environ["JSTSP_H2"] = h2
    os.environ["JSTSP_OMP_GRAM"] = gram
    x, idx = J.omp_kron(inp["A"], inp["B"], y, m); torch.cuda.synchronize()
    t0 = time.perf_counter(); x, idx = J.omp_kron(inp["A"], inp["B"], y, m); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("JSTSP_H2=%s JSTSP_OMP_GRAM=%s  omp_kron m=%d batch %d: %.3f s  (%.0f estimates/s)  idx[0,:6]=%s" % (h2, gram, m, batch, dt, batch / dt, idx[0, :6].tolist()))
    if h2 == "0":
        ref, xref = idx.clone(), x.clone()
    else:
        print("   index sets identical to the first run:", bool(torch.equal(ref, idx)),
              " max |x - x_first| / max|x| = %.2e" % float((x - xref).abs().max() / xref.abs().max()))
