#!/usr/bin/env python3
"""vamp / vamp_kron at the drivers' size (plot_errorVSsnr.m:79-80,100: 512 x 512 dense Phi per trial), 100 iterations, timed."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
from jstsp19_amd.system_model import SweepParams, build_trials
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
p = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4, snr_db=6.0)
inp = build_trials(p, 0, nt, seed=616, with_hbf=True)
Bh = inp["B_hbf"].cpu().numpy().astype(np.complex128); Yh = inp["Y_hbf"].cpu().numpy().astype(np.complex128)
A = inp["A_hbf"].cpu().numpy().astype(np.complex128)
Gb = Bh @ Bh.conj().transpose(0, 2, 1); Ym = Yh @ Bh.conj().transpose(0, 2, 1)
dev = torch.device("cuda:0")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.complex64)).to(dev)
Phi = tt(np.stack([np.kron(Gb[t].T, A) for t in range(nt)])); y = tt(np.stack([Ym[t].flatten("F") for t in range(nt)]))
Ym, A, Gb = tt(Ym), tt(A), tt(Gb)
cm = J.colmajor
for name, fn in (("vamp dense 512 x 512", lambda: J.vamp(y, cm(Phi), 1.0, 100, nit=100)),
                 ("vamp_kron", lambda: J.vamp_kron(cm(Ym), cm(A), cm(Gb), 1.0, 100, nit=100))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-22s %d trials x 100 iterations: %.3f s" % (name, nt, dt))
