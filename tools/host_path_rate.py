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

# ---- the same call as the MEX gateway makes it: MATLAB's interleaved complex DOUBLES in host memory (jstsp_proposed_algorithm_c64)
if len(sys.argv) > 2 and sys.argv[2] == "c64":
    import ctypes as C
    from jstsp19_amd import _lib
    lib, ctx = _lib.load(), J.default_context(0)
    N, M = h["subY"].shape[1:]
    Gr, G2 = A.shape[1], h["B"].shape[1]
    colm = lambda a, dt: np.ascontiguousarray(np.swapaxes(np.asarray(a), -1, -2).astype(dt))
    sy, om, b, a64 = colm(h["subY"], np.complex128), colm(h["Omega"], np.float64), colm(h["B"], np.complex128), colm(A, np.complex128)
    S = np.empty(batch * Gr * G2, np.complex128); Y = np.empty(batch * N * M, np.complex128); ce = np.empty(batch * 300, np.float64)
    ty, ts, rh = (np.ascontiguousarray(np.asarray(inp[k]), dtype=np.float64) for k in ("tau_Y", "tau_Z", "rho"))
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    for rep in range(2):
        t0 = time.perf_counter()
        _lib.check(lib.jstsp_proposed_algorithm_c64(ctx.handle, N, M, Gr, G2, batch, p(sy), p(om), p(a64), 0, p(b), G2 * M, 100, dp(ty), dp(ts), dp(rh),
                                                    0, None, p(S), p(Y), p(ce), 0), "c64")
        dt = time.perf_counter() - t0
        print("host path, complex doubles (the MEX gateway's call): batch %d  %.3f s  %.1f estimates/s (H2D of %.2f GiB of doubles)" % (
            batch, dt, batch / dt, (sy.nbytes + om.nbytes + b.nbytes) / 2**30))
    S32 = np.swapaxes(np.asarray(S).reshape(batch, G2, Gr), 1, 2)
    print("   equals the complex64 host call: %s" % bool(np.array_equal(S32.astype(np.complex64), np.asarray(J.proposed_algorithm(h["subY"], h["Omega"], A, h["B"], 100, inp["tau_Y"], inp["tau_Z"], inp["rho"], "approximate")[0]))))
