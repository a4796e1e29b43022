"""The warm-started lambda_max of the ADMM loops (csrc/eig2.hip: lanczos_lmax_kernel, round 5).

convergence_error(:,1:2) of proposed_algorithm.m:67,69 and the error curves of sparse_admm.m:32 / mc_admm.m:28 need
lambda_max of Gram matrices that barely move between ADMM iterations.  The kernel starts its Lanczos run from the same
matrix's Ritz vector of the previous iteration and stops on the residual of the Ritz pair (cold n-step run otherwise).
Checked here: (1) with JSTSP_LANCZOS_VERIFY=1 every returned value is the cold one - bit for bit what JSTSP_LANCZOS_WARM=0
returns - and no verification disagrees; (2) the default path agrees with the cold path to 2e-5 relative on every
iteration of every trial, S and Y bit-identical (the norms do not feed back); (3) the same for the order-128 kernel inside
sparse_admm / mc_admm; (4) a sequence whose two largest eigenvalues CROSS is followed (the verification catches what a
residual test alone cannot)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _cases():
    rng = np.random.default_rng(77)
    r = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    out = []
    for (N, M, Gr, G2, b) in [(64, 512, 64, 128, 6), (32, 140, 32, 16, 4), (100, 160, 40, 30, 3), (12, 40, 6, 10, 3)]:
        A, B = r(N, Gr) / np.sqrt(N), r(G2, M) / np.sqrt(G2)
        Om = (rng.random((b, N, M)) < 0.4).astype(float)
        out.append((Om * r(b, N, M), Om, A, B, 40, 0.01, 0.02, 0.3, "approximate"))
    return out


def test_verify_always_returns_the_cold_value_bit_for_bit_and_never_disagrees():
    import jstsp19_amd as J
    for args in _cases():
        with _env(JSTSP_LANCZOS_WARM=0):
            S0, Y0, ce0 = J.proposed_algorithm(*args)
        with _env(JSTSP_LANCZOS_VERIFY=1):
            S1, Y1, ce1 = J.proposed_algorithm(*args)
            mism = J.default_context(0).last_lanczos_mismatches()
        assert np.array_equal(S0, S1) and np.array_equal(Y0, Y1)
        assert np.array_equal(ce0, ce1)
        assert mism == 0


def test_default_warm_start_agrees_with_the_cold_run_on_every_iteration():
    import jstsp19_amd as J
    for args in _cases():
        with _env(JSTSP_LANCZOS_WARM=0):
            S0, Y0, ce0 = J.proposed_algorithm(*args)
        S1, Y1, ce1 = J.proposed_algorithm(*args)
        assert J.default_context(0).last_lanczos_mismatches() == 0
        assert np.array_equal(S0, S1) and np.array_equal(Y0, Y1)          # the norms do not feed back
        assert np.array_equal(ce0[..., 2], ce1[..., 2])
        np.testing.assert_allclose(ce1[..., :2], ce0[..., :2], rtol=2e-5)
        with _env(JSTSP_LANCZOS_VERIFY=0):                                # never verified: still the same values
            _, _, ce2 = J.proposed_algorithm(*args)
        np.testing.assert_allclose(ce2[..., :2], ce0[..., :2], rtol=2e-5)


def test_order_128_curves_of_sparse_admm_and_mc_admm():
    import torch
    import jstsp19_amd as J
    dev = torch.device("cuda:0")
    n, batch = 128, 24
    g = torch.Generator(device=dev); g.manual_seed(5)
    rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device=dev), torch.randn(*s, generator=g, device=dev))
    idx = torch.arange(n, device=dev, dtype=torch.float64)
    D = (torch.exp(-2j * np.pi * idx[:, None] * idx[None, :] / n) / np.sqrt(n)).to(torch.complex64)
    Sp = torch.zeros(batch, n, n, dtype=torch.complex64, device=dev)
    Sp[:, ::17, ::13] = rnd(batch, len(range(0, n, 17)), len(range(0, n, 13)))
    H = D @ Sp @ D.conj().T
    OH = H + 0.05 * rnd(batch, n, n)
    Om = (torch.rand(batch, n, n, generator=g, device=dev) < 0.125).float()
    cm = J.colmajor
    tau, rho = np.full(batch, 0.05), np.full(batch, 0.1)
    with _env(JSTSP_LANCZOS_WARM=0):
        S0, ce0 = J.sparse_admm(cm(H), cm(OH), cm(D), cm(D), 60)
        X0, cm0 = J.mc_admm(cm(H), cm(Om * OH), cm(Om), 25, tau, rho)
    S1, ce1 = J.sparse_admm(cm(H), cm(OH), cm(D), cm(D), 60)
    m1 = J.default_context(0).last_lanczos_mismatches()
    X1, cm1 = J.mc_admm(cm(H), cm(Om * OH), cm(Om), 25, tau, rho)
    m2 = J.default_context(0).last_lanczos_mismatches()
    torch.cuda.synchronize()
    assert torch.equal(S0, S1) and torch.equal(X0, X1)
    np.testing.assert_allclose(ce1.cpu().numpy(), ce0.cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(cm1.cpu().numpy(), cm0.cpu().numpy(), rtol=2e-5)
    assert m1 == 0 and m2 == 0


def _herm(U, lam):
    return (U * lam[..., None, :]) @ np.conj(np.swapaxes(U, -1, -2))


@pytest.mark.parametrize("n", [64, 128, 40])
def test_sequences_with_flat_clustered_and_crossing_spectra(n):
    """The kernel on spectra the solvers do not produce, through jstsp_lambda_max_sequence_c32: a slowly rotating basis with
    (a) a well separated top eigenvalue, (b) a cluster of five within 1e-3 at the top, (c) a flat spectrum, and (d) two
    top eigenvalues that CROSS exactly (fixed eigenvectors: the residual test alone would follow the wrong branch for ever -
    the periodic verification must put it back within its period)."""
    import jstsp19_amd as J
    rng = np.random.default_rng(3 + n)
    steps, batch = 80, 4
    Q0, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    K = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    K = (K - K.conj().T) * 2e-3                                  # generator of a slow rotation
    w, V = np.linalg.eig(K)
    base = np.linspace(0.05, 0.6, n)
    flat = 1.0 + 1e-4 * rng.standard_normal(n)
    G = np.empty((steps, batch, n, n), np.complex128)
    ref = np.empty((steps, batch))
    for s in range(steps):
        U = Q0 @ ((V * np.exp(w * s)[None, :]) @ np.linalg.inv(V))
        U, _ = np.linalg.qr(U)
        lam = np.tile(base, (batch, 1)) * (1.0 + 0.002 * s)
        lam[0, -1] = 1.0 + 0.01 * np.sin(0.2 * s)                                    # (a)
        lam[1, -5:] = 1.0 + 1e-3 * np.arange(5) / 4 + 0.003 * s                      # (b)
        lam[2, :] = flat * (1.0 + 0.001 * s)                                         # (c) flat
        lam[3, -1] = 1.0 - 0.004 * (s - 30)                                          # (d) crosses lam[3, -2] at s = 30
        lam[3, -2] = 1.0 + 0.004 * (s - 30)
        Us = np.broadcast_to(U, (batch, n, n)).copy()
        Us[3] = Q0                                                                    # fixed eigenvectors: an EXACT crossing
        G[s] = _herm(Us, lam)
        ref[s] = lam.max(axis=1)
    with _env(JSTSP_LANCZOS_VERIFY=8):
        got = J.lambda_max_sequence(G.astype(np.complex64))
    rel = np.abs(got - ref) / ref
    assert rel[:, :3].max() < 2e-5, rel[:, :3].max(axis=0)
    assert rel[0].max() < 5e-6                                                        # cold start
    # (d): exact before the crossing; behind it the old branch may be followed until the next verification of that matrix
    assert rel[:30, 3].max() < 2e-5
    late = np.nonzero(rel[:, 3] > 2e-5)[0]
    assert len(late) <= 8 and (len(late) == 0 or late.max() < 30 + 9), late
    with _env(JSTSP_LANCZOS_VERIFY=1):
        got1 = J.lambda_max_sequence(G.astype(np.complex64))
    assert (np.abs(got1 - ref) / ref).max() < 5e-6
