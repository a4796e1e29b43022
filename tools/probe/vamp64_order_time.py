"""Cost of the float64 VAMP (csrc/vamp64.hip) against the order of the delay factor's Gram: one problem, 10 iterations; the float64 Jacobi
of the Gram dominates (~n^3: 0.06 / 0.21 / 1.76 s at 512 / 1024 / 2048 on MI355X)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
rng = np.random.default_rng(1)
r = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
for n in (512, 1024, 2048):
    Na = 32
    A = r(Na, Na) / np.sqrt(Na)
    Bh = r(n, 2 * n) / np.sqrt(n)
    Gb = Bh @ Bh.conj().T
    Y = r(1, Na, n)
    t0 = time.perf_counter(); X = J.vamp_kron(Y, A, Gb[None], 1.0, 100, nit=10); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); X = J.vamp_kron(Y, A, Gb[None], 1.0, 100, nit=10); dt2 = time.perf_counter() - t0
    print("float64 vamp_kron, Gb order %d, 1 problem, 10 iterations: %.2f s (second call %.2f s)" % (n, dt, dt2), flush=True)
