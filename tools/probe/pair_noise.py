#!/usr/bin/env python3
"""The rounding noise of the two-trials-per-workgroup kernel against the per-trial kernels (experiments build: JSTSP_EXPERIMENTS_LIB=1,
JSTSP_HGEMM_PAIR=0 / 1 given on the command line of two runs): proposed_algorithm with one pilot set, N = 64, G2 = 2048, M = 4096, 32
trials, 10 iterations; S, Y of three trials against the float64 oracle.   usage: JSTSP_EXPERIMENTS_LIB=1 JSTSP_HGEMM_PAIR=0 python pair_noise.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
from oracle import solvers as O
rng = np.random.default_rng(4747)
r = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
N, M, Gr, G2, b, Imax = 64, 4096, 64, 2048, 32, 10
A, B = r(N, Gr) / np.sqrt(N), r(G2, M) / np.sqrt(G2)
Om = (rng.random((b, N, M)) < 0.25).astype(float)
S0 = np.zeros((b, Gr, G2), complex)
for t in range(b):
    S0[t, rng.integers(0, Gr, 6), rng.integers(0, G2, 6)] = r(6)
subY = Om * (A @ S0 @ B + 0.05 * r(b, N, M))
fro2 = (np.abs(subY) ** 2).sum((1, 2))
tY, tZ, rho = 1.0 / fro2, np.full(b, 1e-2), np.full(b, 0.25)
S, Y, ce = J.proposed_algorithm(subY, Om, A, B, Imax, tY, tZ, rho, "approximate")
rel = lambda a, c: float(np.max(np.abs(a - c)) / np.max(np.abs(c)))
es, ey = [], []
for t in (0, 5, 11, 17, 23, 31):
    So, Yo, _ = O.proposed_algorithm(subY[t], Om[t], A, B, Imax, float(tY[t]), float(tZ[t]), float(rho[t]), "approximate", want_ce=False)
    es.append(rel(S[t], So)); ey.append(rel(Y[t], Yo))
print("JSTSP_HGEMM_PAIR=%s: rel S max %.3e mean %.3e   rel Y max %.3e mean %.3e" % (os.environ.get("JSTSP_HGEMM_PAIR", "(default)"), max(es), np.mean(es),
                                                                                  max(ey), np.mean(ey)))
