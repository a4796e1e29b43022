#!/usr/bin/env python3
"""BASELINE configs[0] on the GPU: OMP.m:16-24 with the dense 1024 x 1024 dictionary Phi = kron(B.', A) (Nt=Nr=16, Nrf=4, K=16,
L=4: A 16 x 16, B 64 x 64), m = 24 atoms - one problem (the reference's case) and a batch of independent problems with one
shared dictionary.  Per OMP iteration the dense path reads the dictionary once: 8 * meas * size_d bytes,
which is what the printed bandwidth is computed from (kernel time: profile this script with tools/prof_cmd.sh)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J

rng = np.random.default_rng(16)
c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
A = (c(16, 16) / 4).astype(np.complex64)
B = (c(64, 64) / 8).astype(np.complex64)
Phi = np.kron(B.T, A).astype(np.complex64)                      # 1024 x 1024 (meas x size_d)
m = 24
dev = torch.device("cuda:0")
Phi_d = J.colmajor(torch.from_numpy(Phi).to(dev))
for batch in (1, 64, 1024):
    x0 = np.zeros((batch, 1024), np.complex64)
    for t in range(batch):
        x0[t, rng.choice(1024, 6, replace=False)] = c(6)
    y = (x0 @ Phi.T + 0.01 * c(batch, 1024)).astype(np.complex64)
    y_d = torch.from_numpy(y).to(dev)
    r = J.OMP(Phi_d, y_d, m, want_target=False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        r = J.OMP(Phi_d, y_d, m, want_target=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    byts = 8.0 * 1024 * 1024 * m                                  # dictionary bytes per problem set and call (shared dictionary: read once per iteration)
    print("OMP dense 1024 x 1024, m = %d, batch %4d: %.3f ms per call, %.1f problems/s; dictionary traffic %.1f MB per call -> %.1f GB/s if read once per iteration"
          % (m, batch, dt * 1e3, batch / dt, byts / 1e6, byts / dt / 1e9))
