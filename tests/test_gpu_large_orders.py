"""Orders above 128 (csrc/eig_large.hip: the LDS-resident Jacobi kernels stop at 128; above that the Hermitian
eigen-decomposition is the library's two-sided block Jacobi - 128 x 128 sub-problems on the existing kernel, their
unitaries applied by batched GEMMs): svt / mc_svt of inputs whose BOTH dimensions exceed 128, and VAMP on the Kronecker
dictionary with G2 = L*Gt > 128 (towards BASELINE configs[4]).  Checked against the float64 oracle."""
import numpy as np
import pytest

import jstsp19_amd as J
from conftest import rel_err

pytestmark = pytest.mark.gpu


def _lowrank(rng, batch, R, C, r, noise):
    H = (rng.standard_normal((batch, R, r)) + 1j * rng.standard_normal((batch, R, r))) @ \
        (rng.standard_normal((batch, r, C)) + 1j * rng.standard_normal((batch, r, C)))
    return H + noise * (rng.standard_normal((batch, R, C)) + 1j * rng.standard_normal((batch, R, C)))


@pytest.mark.parametrize("shape", [(160, 192), (200, 136), (130, 131), (320, 400), (512, 640), (700, 520)])
def test_svt_both_dimensions_above_128(shape):
    from oracle import solvers as O
    rng = np.random.default_rng(3)
    R, C = shape
    Y = 0.2 * _lowrank(rng, 3, R, C, 5, 1.5)               # sigma: 5 of about 35..80, the rest between 0.3 and 8
    tau = np.array([0.2, 2.0, 20.0])                       # below, inside and above the noise part of the spectrum
    # (the SVT is formed from the fp32 Gram of the short side: singular values are resolved to about 1e-7 sigma_max^2 / sigma)
    X = np.asarray(J.svt(Y, tau))
    for t in range(3):
        ref = O.svt(Y[t], tau[t])
        assert rel_err(X[t], ref) < 3e-5, (t, rel_err(X[t], ref))
    assert np.count_nonzero(np.asarray(J.svt(np.zeros((R, C), complex), 0.1))) == 0      # svt.m:7-12 on the zero matrix


def test_mc_svt_and_mc_admm_above_128():
    from oracle import solvers as O
    rng = np.random.default_rng(4)
    n = 144
    H = _lowrank(rng, 2, n, n, 4, 0.0)
    Om = (rng.random((2, n, n)) < 0.5).astype(float)
    OH = Om * H
    X = np.asarray(J.mc_svt(OH, Om, 8, 2.0, 0.3))
    Xa, ce = J.mc_admm(H, OH, Om, 8, 2.0, 0.3)
    for t in range(2):
        assert rel_err(X[t], O.mc_svt(OH[t], Om[t], 8, 2.0, 0.3)) < 1e-4
        Xo, ceo = O.mc_admm(H[t], OH[t], Om[t], 8, 2.0, 0.3)
        assert rel_err(np.asarray(Xa)[t], Xo) < 1e-4
        np.testing.assert_allclose(np.asarray(ce)[t], np.ravel(ceo), rtol=2e-3)


def test_vamp_kron_with_a_large_delay_factor():
    """Phi = kron(Gb.', Af) with Gb of order 160 (> 128): the first iterations follow the float64 restatement (later ones
    are chaotic in any precision, as for the small orders: tests/test_gpu_baselines.py)."""
    from oracle import vamp as V
    rng = np.random.default_rng(6)
    Na, Gr, G2, T = 16, 24, 160, 200
    Af = (rng.standard_normal((Na, Gr)) + 1j * rng.standard_normal((Na, Gr))) / np.sqrt(2 * Na)
    Bh = (rng.standard_normal((G2, T)) + 1j * rng.standard_normal((G2, T))) / np.sqrt(2 * T)
    Gb = Bh @ Bh.conj().T
    X0 = np.zeros((Gr, G2), complex)
    ix = rng.choice(Gr * G2, 12, replace=False)
    X0.flat[ix] = 3 * (rng.standard_normal(12) + 1j * rng.standard_normal(12))
    Y = Af @ X0 @ Gb + 0.05 * (rng.standard_normal((Na, G2)) + 1j * rng.standard_normal((Na, G2)))
    for nit, tol in ((1, 2e-5), (4, 2e-4)):
        out = np.asarray(J.vamp_kron(Y, Af, Gb, 1.0, 12, nit=nit))
        ref = V.vamp_kron(Y, Af, Gb, 1.0, 12, nit=nit)
        assert rel_err(out, ref) < tol, (nit, rel_err(out, ref))


def test_svt_of_an_input_with_an_exactly_diagonal_gram():
    """An exactly diagonal Gram: nothing to rotate, the decomposition is the input (rounds 1-2: rocSOLVER's cheevd returned NaN
    eigenvectors here)."""
    from oracle import solvers as O
    n, C = 136, 200
    Y = np.zeros((n, C), complex)
    Y[np.arange(n), np.arange(n)] = 1.0 + 0.05 * np.arange(n)
    X = np.asarray(J.svt(Y, 2.5))
    assert np.isfinite(X).all() and rel_err(X, O.svt(Y, 2.5)) < 1e-6


def test_proposed_algorithm_and_sparse_admm_above_128():
    from oracle import solvers as O
    rng = np.random.default_rng(8)
    N, M, Gr, G2, Imax = 136, 150, 136, 20, 6
    A = (rng.standard_normal((N, Gr)) + 1j * rng.standard_normal((N, Gr))) / np.sqrt(2 * N)
    B = (rng.standard_normal((G2, M)) + 1j * rng.standard_normal((G2, M))) / np.sqrt(2 * G2)
    S0 = np.zeros((Gr, G2), complex)
    ix = rng.choice(Gr * G2, 10, replace=False)
    S0.flat[ix] = rng.standard_normal(10) + 1j * rng.standard_normal(10)
    Om = (rng.random((N, M)) < 0.4).astype(float)
    subY = Om * (A @ S0 @ B + 0.05 * (rng.standard_normal((N, M)) + 1j * rng.standard_normal((N, M))))
    S, Y, ce = J.proposed_algorithm(subY, Om, A, B, Imax, 0.02, 0.01, 0.4, "approximate")
    So, Yo, ceo = O.proposed_algorithm(subY, Om, A, B, Imax, 0.02, 0.01, 0.4, "approximate")
    assert rel_err(S, So) < 3e-4 and rel_err(Y, Yo) < 3e-4
    np.testing.assert_allclose(np.asarray(ce)[1:], ceo[1:], rtol=3e-3)
    # sparse_admm.m on 144 x 144 with unitary DFT dictionaries
    n = 144
    F = np.fft.fft(np.eye(n)) / np.sqrt(n)
    H = _lowrank(rng, 1, n, n, 3, 0.0)[0]
    OH = (rng.random((n, n)) < 0.6) * H
    Ss, ces = J.sparse_admm(H, OH, F, F, 6)
    Sso, ceso = O.sparse_admm(H, OH, F, F, 6)
    assert rel_err(Ss, Sso) < 3e-4


def test_library_has_no_vendor_lapack_dependency():
    """No rocSOLVER / rocBLAS behind the C ABI any more: neither linked nor dlopen'ed (the order > 128 path is csrc/eig_large.hip)."""
    import os
    import subprocess
    out = subprocess.run(["ldd", J.LIB_PATH], capture_output=True, text=True).stdout
    assert "rocsolver" not in out and "rocblas" not in out
    src = os.path.join(os.path.dirname(J.LIB_PATH))
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")):
            assert "dlopen" not in open(os.path.join(src, f)).read(), f


@pytest.mark.parametrize("cu_mask", ["1", "0"])
def test_eigen_decomposition_of_order_1024_through_the_kronecker_vamp(cu_mask, monkeypatch):
    """A delay factor of order 1024 (16 blocks of 64): first VAMP iterations against the float64 oracle - they use the
    eigenvectors and eigenvalues of Gb directly (U^H (.) U, d = la x lb^2).  Both schedules of the block Jacobi: its chain of
    sub-problems on 32 reserved compute units beside the panel products (streams with a compute-unit mask; the default when a
    round has at most 32 sub-problems), and on plain side streams (JSTSP_BJ_MASK=0; what a runtime without masks gets)."""
    from oracle import vamp as V
    monkeypatch.setenv("JSTSP_BJ_MASK", cu_mask)
    rng = np.random.default_rng(16)
    Na, Gr, G2, T = 16, 16, 1024, 1400
    Af = (rng.standard_normal((Na, Gr)) + 1j * rng.standard_normal((Na, Gr))) / np.sqrt(2 * Na)
    Bh = (rng.standard_normal((G2, T)) + 1j * rng.standard_normal((G2, T))) / np.sqrt(2 * T)
    Gb = Bh @ Bh.conj().T
    X0 = np.zeros((Gr, G2), complex)
    X0.flat[rng.choice(Gr * G2, 30, replace=False)] = 3 * (rng.standard_normal(30) + 1j * rng.standard_normal(30))
    Y = Af @ X0 @ Gb + 0.05 * (rng.standard_normal((Na, G2)) + 1j * rng.standard_normal((Na, G2)))
    for nit, tol in ((2, 1e-4), (4, 1e-3)):
        out = np.asarray(J.vamp_kron(Y, Af, Gb, 1.0, 30, nit=nit))
        ref = V.vamp_kron(Y, Af, Gb, 1.0, 30, nit=nit)
        assert rel_err(out, ref) < tol, (nit, rel_err(out, ref))


def test_nmse_spectral_above_128():
    from oracle import solvers as O
    rng = np.random.default_rng(12)
    Zb = _lowrank(rng, 2, 150, 170, 4, 0.1)
    S = Zb + 0.3 * _lowrank(rng, 2, 150, 170, 2, 0.05)
    out = np.asarray(J.nmse_spectral(S, Zb))
    for t in range(2):
        assert abs(out[t] - O.nmse_capped(S[t], Zb[t])) < 2e-6 * max(1.0, out[t])


def test_svt_above_128_with_more_sub_problems_than_reserved_units():
    """40 matrices of 150 x 170: 80 pair sub-problems per round - more than the 32 compute units the masked streams set aside,
    so the block Jacobi runs its pipeline on plain streams without an environment switch (csrc/eig_large.hip)."""
    from oracle import solvers as O
    rng = np.random.default_rng(21)
    Y = 0.2 * _lowrank(rng, 40, 150, 170, 4, 1.0)
    tau = np.linspace(0.2, 6.0, 40)
    X = np.asarray(J.svt(Y, tau))
    for t in (0, 13, 39):
        ref = O.svt(Y[t], tau[t])
        assert rel_err(X[t], ref) < 3e-5, (t, rel_err(X[t], ref))
