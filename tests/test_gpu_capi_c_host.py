"""The C ABI from a plain C program (tests/capi/host_example.c: gcc, no Python / torch in that process), host
memory in and out exactly as the MEX gateway calls it; outputs checked against the float64 oracle on the same
LCG-generated inputs."""
import os
import subprocess
import sys

import numpy as np
import pytest
from conftest import check_below, ce_rel, TOL_S, TOL_CE, TOL_NMSE  # noqa: E402

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _LCG:
    def __init__(self):
        self.s = 0x2545F4914F6CDD1D

    def uniform(self):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) & (2 ** 64 - 1)
        return ((self.s >> 11) + 0.5) / 9007199254740992.0

    def c32(self, n, scale):
        out = np.empty(n, dtype=np.complex64)
        for i in range(n):
            re = np.float32(scale * (2.0 * self.uniform() - 1.0))
            im = np.float32(scale * (2.0 * self.uniform() - 1.0))
            out[i] = complex(re, im)
        return out


def test_c_host_program_matches_oracle(tmp_path):
    from oracle import solvers as O
    exe, out = str(tmp_path / "host_example"), str(tmp_path / "out.bin")
    libdir = os.path.join(ROOT, "jstsp19_amd", "csrc")
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "capi", "host_example.c"),
                    "-o", exe, "-L", libdir, "-ljstsp_mi355x", "-lm", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([exe, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "gfx950" in r.stdout

    N, M, Gr, G2, Imax, m = 12, 40, 10, 18, 25, 5
    g = _LCG()
    A = g.c32(N * Gr, 1.0 / np.sqrt(N)).reshape(Gr, N).T               # column-major fill
    B = g.c32(G2 * M, 1.0 / np.sqrt(G2)).reshape(M, G2).T
    subY = g.c32(N * M, 1.0).reshape(M, N).T.copy()
    Om = np.empty((M, N), dtype=np.float32)
    sy = subY.T.copy()
    for i in range(N * M):
        o = np.float32(1.0 if g.uniform() < 0.4 else 0.0)
        Om.flat[i] = o
        sy.flat[i] = sy.flat[i] * o
    Om, subY = Om.T, sy.T
    raw = np.fromfile(out, dtype=np.uint8)
    off = 0

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a
    S = take(np.complex64, Gr * G2).reshape(G2, Gr).T
    Y = take(np.complex64, N * M).reshape(M, N).T
    ce = take(np.float64, Imax * 3).reshape(3, Imax).T
    xh = take(np.complex64, Gr * G2)
    idx = take(np.int32, m)

    So, Yo, ceo = O.proposed_algorithm(subY.astype(complex), Om.astype(float), A.astype(complex), B.astype(complex), Imax,
                                       0.02, 0.01, 0.35, "approximate")
    check_below("c_host.S", rel_err(S, So), TOL_S); check_below("c_host.Y", rel_err(Y, Yo), TOL_S)
    check_below("c_host.ce", ce_rel(ce, ceo), TOL_CE)
    xo, io, _, _ = O.omp_literal(np.kron(B.astype(complex).T, A.astype(complex)), subY.astype(complex).reshape(-1, order="F"), m)
    assert np.array_equal(idx, io) and rel_err(xh, xo) < 1e-4
