"""Kernel-level parity: the MFMA correlation / synthesis GEMMs, svt, spectral NMSE, mc_*."""
import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _rand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.mark.parametrize("N,M,Gr,G2,batch,shared", [
    (32, 140, 32, 16, 1, True),        # reference-native
    (64, 256, 64, 128, 3, False),      # tile-aligned, per-trial dictionaries
    (7, 13, 5, 9, 2, True),            # ragged
    (64, 4096, 64, 512, 2, False),     # BASELINE config 2 shape
    (33, 65, 70, 130, 2, False),       # one past the tile in every dimension
])
def test_correlate_and_synthesize_match_numpy(N, M, Gr, G2, batch, shared):
    """A^H K B^H and A S B against float64 numpy.  Inputs are asymmetric complex random
    matrices, so a transposed / conjugated / re-im-swapped tile cannot pass."""
    import jstsp19_amd as J
    rng = np.random.default_rng(N * 1000 + M)
    K = _rand(rng, batch, N, M)
    S = _rand(rng, batch, Gr, G2)
    A = _rand(rng, N, Gr) if shared else _rand(rng, batch, N, Gr)
    B = _rand(rng, G2, M) if shared else _rand(rng, batch, G2, M)
    Ab = np.broadcast_to(A, (batch, N, Gr)); Bb = np.broadcast_to(B, (batch, G2, M))
    ref_c = np.conj(np.swapaxes(Ab, 1, 2)) @ K @ np.conj(np.swapaxes(Bb, 1, 2))
    ref_s = Ab @ S @ Bb
    out_c = J.correlate(K, A, B)
    out_s = J.synthesize(S, A, B)
    # fp32 MFMA accumulation over k: error ~ 1e-7 * sqrt(k) * |a||b|
    assert rel_err(out_c, ref_c) < 5e-6
    assert rel_err(out_s, ref_s) < 5e-6


def test_correlate_device_tensors_equal_host_path():
    import torch
    import jstsp19_amd as J
    rng = np.random.default_rng(3)
    K, A, B = _rand(rng, 4, 32, 140), _rand(rng, 32, 32), _rand(rng, 4, 16, 140)
    host = J.correlate(K, A, B)
    cm = lambda a: J.colmajor(torch.from_numpy(a.astype(np.complex64)).cuda())
    dev = J.correlate(cm(K), cm(A), cm(B))
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy(), host)


def test_svt_matches_golden_and_known_answers():
    import jstsp19_amd as J
    from oracle import solvers as O
    g = load_golden("svt")
    for k in range(int(g["n"])):
        X = J.svt(g["Y%d" % k], float(g["tau%d" % k]))
        assert rel_err(X, g["X%d" % k]) < 1e-4
    # zero matrix -> zeros (svt.m:8-12 guard)
    assert np.all(J.svt(np.zeros((6, 9), complex), 0.5) == 0)
    # rank-1: svt(s u v^H, tau) = max(s - tau, 0) u v^H
    u = np.array([1, 2j, -1, 0.5]) / np.linalg.norm([1, 2, 1, 0.5])
    v = np.array([1j, 1, 1, -1, 2]) / np.linalg.norm([1, 1, 1, 1, 2])
    Y = 3.0 * np.outer(u, v.conj())
    assert rel_err(J.svt(Y, 1.0), 2.0 * np.outer(u, v.conj())) < 1e-5
    assert np.max(np.abs(J.svt(Y, 3.5))) < 1e-5
    # batched, different thresholds per problem
    rng = np.random.default_rng(5)
    Yb = _rand(rng, 3, 12, 40)
    taus = np.array([0.1, 2.0, 50.0])
    Xb = J.svt(Yb, taus)
    for t in range(3):
        assert rel_err(Xb[t], O.svt(Yb[t], taus[t])) < 1e-4 or np.max(np.abs(O.svt(Yb[t], taus[t]))) == 0


def test_nmse_spectral_matches_oracle():
    import jstsp19_amd as J
    from oracle import solvers as O
    rng = np.random.default_rng(9)
    Z = _rand(rng, 4, 32, 16)
    S = Z + np.array([0.01, 0.1, 1.0, 5.0])[:, None, None] * _rand(rng, 4, 32, 16)
    out = J.nmse_spectral(S, Z)
    ref = np.array([O.nmse_capped(S[t], Z[t]) for t in range(4)])
    np.testing.assert_allclose(out, ref, rtol=2e-5)
    assert out[3] == 1.0                                   # clipped (plot_errorVSsnr.m:139-141)


def test_mc_svt_and_mc_admm_match_golden():
    import jstsp19_amd as J
    g = load_golden("mc")
    X = J.mc_svt(g["OH"], g["Omega"], int(g["Imax"]), float(g["tau"]), float(g["rho"]))
    assert rel_err(X, g["X_svt"]) < 2e-4
    X2, ce = J.mc_admm(g["Htrue"], g["OH"], g["Omega"], int(g["Imax"]), float(g["tau"]), float(g["rho"]))
    assert rel_err(X2, g["X_admm"]) < 2e-4
    np.testing.assert_allclose(ce, g["ce_admm"], rtol=2e-3)


def test_svt_order_above_64_and_tall_matrices():
    """Gram eigenproblems of order 65..128 take the general Jacobi kernel (eigenvectors in LDS up to
    n = 98, in HBM above); tall inputs (rows > cols) decompose Z^H Z and apply Q from the right."""
    import jstsp19_amd as J
    from oracle import solvers as O
    rng = np.random.default_rng(12)
    for (r, c) in [(96, 150), (128, 128), (140, 100), (12, 9)]:
        low = _rand(rng, r, 4) @ _rand(rng, 4, c)
        Y = low + 0.2 * _rand(rng, r, c)
        tau = float(0.5 * np.linalg.svd(Y, compute_uv=False)[3])
        assert rel_err(J.svt(Y, tau), O.svt(Y, tau)) < 3e-4, (r, c)


def test_proposed_tall_measurement_matrix_unfused_path():
    """N > M: the SVT decomposes the M x M Gram and the ADMM updates run as separate kernels."""
    import jstsp19_amd as J
    from oracle import solvers as O
    rng = np.random.default_rng(13)
    N, M, Gr, G2 = 12, 9, 6, 5
    A = _rand(rng, N, Gr) / np.sqrt(N); B = _rand(rng, G2, M) / np.sqrt(G2)
    S0 = np.zeros((Gr, G2), complex); S0[1, 2] = 2 - 1j; S0[3, 0] = 1j
    Om = (rng.random((N, M)) < 0.6).astype(float)
    subY = Om * (A @ S0 @ B + 0.05 * _rand(rng, N, M))
    args = (subY, Om, A, B, 20, 0.01, 0.02, 0.3, "approximate")
    So, Yo, ceo = O.proposed_algorithm(*args)
    S, Y, ce = J.proposed_algorithm(*args)
    assert rel_err(S, So) < 2e-4 and rel_err(Y, Yo) < 2e-4
    np.testing.assert_allclose(ce[:, :2], ceo[:, :2], rtol=2e-3)


def test_svt_zero_row_follows_the_oracle_and_zero_matrix_hits_the_guard():
    """svt.m:7-12 returns zeros when a singular value is EXACTLY 0.  Only the all-zero input is exact on every LAPACK
    (numpy's gesdd gives 4e-16 for a zero first / middle row and 0.0 for a zero last row): the HIP path reproduces the
    guard for the all-zero matrix and otherwise removes the null component, which is what the oracle computes for a
    zero row in the middle."""
    import jstsp19_amd as J
    from oracle import solvers as O
    rng = np.random.default_rng(31)
    for shape, kind in [((12, 40), "row"), ((40, 12), "col"), ((100, 130), "row"), ((64, 300), "row")]:
        Y = _rand(rng, *shape)
        if kind == "row":
            Y[5] = 0
        else:
            Y[:, 3] = 0
        ref = O.svt(Y, 0.7)
        assert np.max(np.abs(ref)) > 0                       # the oracle's guard does not fire here
        assert rel_err(J.svt(Y, 0.7), ref) < 3e-4, shape
    for shape in [(6, 9), (9, 6), (70, 90), (64, 64)]:
        assert np.all(J.svt(np.zeros(shape, complex), 0.5) == 0)
    # mc_svt / mc_admm with a row that Omega never samples: the whole trajectory follows the oracle
    OH = _rand(rng, 16, 16)
    Om = (rng.random((16, 16)) < 0.5).astype(float); Om[4] = 0
    assert rel_err(J.mc_svt(Om * OH, Om, 12, 0.5, 0.2), O.mc_svt(Om * OH, Om, 12, 0.5, 0.2)) < 2e-4
    X, ce = J.mc_admm(OH, Om * OH, Om, 12, 0.5, 0.2)
    Xo, ceo = O.mc_admm(OH, Om * OH, Om, 12, 0.5, 0.2)
    assert rel_err(X, Xo) < 2e-4
    np.testing.assert_allclose(ce, ceo, rtol=2e-3)


def test_mc_svt_and_mc_admm_orders_above_64_warm_started_register_resident_jacobi():
    """Gram orders 65..128 take jacobi128_kernel (csrc/eig3.hip: G in LDS, eigenvector basis in registers); inside the
    mc_* loops the basis of the previous iteration warm-starts it (G <- U^H G U by two batched GEMMs).  Whole
    trajectories against the float64 oracle, square / wide / tall, batched; JSTSP_EIG128=0 (basis in HBM) agrees."""
    import os
    import jstsp19_amd as J
    from oracle import solvers as O
    rng = np.random.default_rng(52)
    for (r, c) in [(72, 80), (128, 128), (100, 70)]:
        low = _rand(rng, 2, r, 5) @ _rand(rng, 2, 5, c)
        H = low + 0.05 * _rand(rng, 2, r, c)
        Om = (rng.random((2, r, c)) < 0.6).astype(float)
        tau, rho = 2.0, 0.3
        X = J.mc_svt(Om * H, Om, 8, tau, rho)
        X2, ce = J.mc_admm(H, Om * H, Om, 8, tau, rho)
        for t in range(2):
            assert rel_err(X[t], O.mc_svt(Om[t] * H[t], Om[t], 8, tau, rho)) < 5e-4, (r, c)
            Xo, ceo = O.mc_admm(H[t], Om[t] * H[t], Om[t], 8, tau, rho)
            assert rel_err(X2[t], Xo) < 5e-4, (r, c)
            np.testing.assert_allclose(ce[t], ceo, rtol=2e-3)
    old = os.environ.get("JSTSP_EIG128")
    try:
        Y = _rand(rng, 96, 150)
        a = J.svt(Y, 3.0)
        os.environ["JSTSP_EIG128"] = "0"
        b = J.svt(Y, 3.0)
        assert rel_err(a, b) < 1e-5
    finally:
        if old is None:
            os.environ.pop("JSTSP_EIG128", None)
        else:
            os.environ["JSTSP_EIG128"] = old
