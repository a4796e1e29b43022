#!/usr/bin/env python3
"""dense vamp at the drivers' size: single call against the batched call's first problem (relative difference)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import jstsp19_amd as J
from jstsp19_amd.system_model import SweepParams, build_trials
nt, numOfnz = 3, 100
p = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4, snr_db=6.0)
inp = build_trials(p, 0, nt, seed=616, with_hbf=True)
Bh = inp["B_hbf"].cpu().numpy().astype(np.complex128); Yh = inp["Y_hbf"].cpu().numpy().astype(np.complex128)
A = inp["A_hbf"].cpu().numpy().astype(np.complex128)
Gb = Bh @ Bh.conj().transpose(0, 2, 1); Ym = Yh @ Bh.conj().transpose(0, 2, 1)
Phi = np.stack([np.kron(Gb[t].T, A) for t in range(nt)]); y = np.stack([Ym[t].flatten("F") for t in range(nt)])
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for nit in (2, 5, 12):
    xb = np.asarray(J.vamp(y, Phi, 1.0, numOfnz, nit=nit))
    print(nit, ["%.2e" % rel(np.asarray(J.vamp(y[t], Phi[t], 1.0, numOfnz, nit=nit)), xb[t]) for t in range(nt)])
import hashlib
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
print("single nit=5 twice:", h(np.asarray(J.vamp(y[0], Phi[0], 1.0, numOfnz, nit=5))), h(np.asarray(J.vamp(y[0], Phi[0], 1.0, numOfnz, nit=5))))
print("batched nit=5 twice:", h(np.asarray(J.vamp(y, Phi, 1.0, numOfnz, nit=5))), h(np.asarray(J.vamp(y, Phi, 1.0, numOfnz, nit=5))))
x1 = np.asarray(J.vamp(y[0], Phi[0], 1.0, numOfnz, nit=5)); xb = np.asarray(J.vamp(y, Phi, 1.0, numOfnz, nit=5))[0]
print("single then batched:", "%.2e" % rel(x1, xb))
