"""Build oracle/_cpu/libjstsp_cpu_port.so from oracle/cpu_port.cpp (TEST INFRASTRUCTURE: the CPU baseline of bench.py).

x86-64-v4 (AVX-512) is what both this container's and the GPU box's host CPUs provide; `native=True` (bench.py on the GPU
box, into a scratch directory) lets gcc tune for the machine the baseline is timed on."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpu_port.cpp")
OUT_DIR = os.path.join(HERE, "_cpu")
LIB = os.path.join(OUT_DIR, "libjstsp_cpu_port.so")


def build(native=False, out=None, verbose=False):
    out = out or LIB
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(SRC) and not native:
        return out
    cmd = ["g++", "-O3", "-march=native" if native else "-march=x86-64-v4", "-fopenmp", "-std=c++17", "-shared", "-fPIC",
           "-o", out, SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed for cpu_port.cpp:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", out)
    return out


def load(path=None):
    """ctypes handle with the prototype of jstsp_cpu_proposed_algorithm set."""
    import ctypes as C
    lib = C.CDLL(path or build())
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.jstsp_cpu_proposed_algorithm.restype = C.c_int
    lib.jstsp_cpu_proposed_algorithm.argtypes = [C.c_int] * 5 + [dp, dp, dp, dp, C.c_longlong, C.c_int, dp, dp, dp, ip, C.c_int,
                                                 dp, dp, dp, C.c_int]
    return lib


def proposed_algorithm(lib, subY, Omega, A, B, Imax, tau_Y, tau_S, rho, indx_S=None, want_ce=True, threads=0):
    """Batched call on numpy arrays shaped (batch, rows, cols) (any memory order; copied to column-major interleaved
    complex128).  B: (batch, G2, M) or (G2, M) shared.  Returns S (batch, Gr, G2), Y (batch, N, M), ce (batch, Imax, 3),
    threads used."""
    import ctypes as C
    import numpy as np
    subY = np.asarray(subY)
    batch, N, M = subY.shape
    Gr, G2 = A.shape[1], B.shape[-2]
    colm = lambda x, dt: np.ascontiguousarray(np.swapaxes(np.asarray(x, dtype=dt), -1, -2))   # (.., cols, rows) C-order = column-major
    sY, Om, Ac, Bc = colm(subY, np.complex128), colm(Omega, np.float64), colm(A, np.complex128), colm(B, np.complex128)
    strideB = 0 if Bc.ndim == 2 else G2 * M
    S = np.empty((batch, G2, Gr), np.complex128); Y = np.empty((batch, M, N), np.complex128)
    ce = np.empty((batch, 3, Imax), np.float64)
    vec = lambda v: np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float64), (batch,)))
    tY, tS, rh = vec(tau_Y), vec(tau_S), vec(rho)
    idx = None if indx_S is None else np.ascontiguousarray(np.asarray(indx_S, np.int32).reshape(batch, Gr * G2))
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    p = lambda a: a.ctypes.data_as(dp)
    used = lib.jstsp_cpu_proposed_algorithm(N, M, Gr, G2, batch, p(sY), p(Om), p(Ac), p(Bc), strideB, Imax, p(tY), p(tS), p(rh),
                                            None if idx is None else idx.ctypes.data_as(ip), 1 if want_ce else 0, p(S), p(Y),
                                            p(ce), threads)
    if used < 1:
        raise RuntimeError("jstsp_cpu_proposed_algorithm: bad argument")
    return np.swapaxes(S, 1, 2), np.swapaxes(Y, 1, 2), np.swapaxes(ce, 1, 2), used


if __name__ == "__main__":
    build(verbose=True)
