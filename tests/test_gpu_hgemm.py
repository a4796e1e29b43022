"""The split-f16 GEMM / Gram path (csrc/hgemm.hip) forced on at every shape (JSTSP_H2=2; by default it is
only chosen for big contractions): fp32-equivalent accuracy against float64, ragged tiles, two row tiles,
shared and per-trial dictionaries, operand scales far from 1, zeros — and the whole ADMM through it against
the golden fixtures with the same tolerances as the fp32 path."""
import os

import numpy as np
import pytest
from conftest import check_below, ce_rel, TOL_S, TOL_CE, TOL_NMSE  # noqa: E402

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_h2():
    old = os.environ.get("JSTSP_H2")
    os.environ["JSTSP_H2"] = "2"
    yield
    if old is None:
        del os.environ["JSTSP_H2"]
    else:
        os.environ["JSTSP_H2"] = old


def _rand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.mark.parametrize("N,M,Gr,G2,batch,shared", [
    (32, 140, 32, 16, 1, True),        # reference-native: everything smaller than one tile
    (64, 256, 64, 128, 3, False),      # tile-aligned, per-trial dictionaries
    (7, 13, 5, 9, 2, True),            # ragged, k shorter than one stage
    (33, 65, 70, 130, 2, False),       # one past the tile / stage in every dimension
    (100, 97, 40, 75, 2, True),        # two row tiles of the a operand
    (64, 1100, 64, 192, 9, False),     # batch not a multiple of the 8 XCDs, several folds of the accumulators
])
def test_correlate_and_synthesize_split_f16_match_numpy(force_h2, N, M, Gr, G2, batch, shared):
    import jstsp19_amd as J
    rng = np.random.default_rng(N * 1000 + M)
    K, S = _rand(rng, batch, N, M), _rand(rng, batch, Gr, G2)
    A = _rand(rng, N, Gr) if shared else _rand(rng, batch, N, Gr)
    B = _rand(rng, G2, M) if shared else _rand(rng, batch, G2, M)
    Ab, Bb = np.broadcast_to(A, (batch, N, Gr)), np.broadcast_to(B, (batch, G2, M))
    ref_c = np.conj(np.swapaxes(Ab, 1, 2)) @ K @ np.conj(np.swapaxes(Bb, 1, 2))
    ref_s = Ab @ S @ Bb
    assert rel_err(J.correlate(K, A, B), ref_c) < 5e-6
    assert rel_err(J.synthesize(S, A, B), ref_s) < 5e-6


def test_split_f16_is_scale_free_and_handles_zeros(force_h2):
    """The per-problem power-of-two scaling keeps the f16 pieces in range whatever the magnitudes are."""
    import jstsp19_amd as J
    rng = np.random.default_rng(5)
    K, A, B = _rand(rng, 3, 48, 200), _rand(rng, 48, 40), _rand(rng, 3, 72, 200)
    K[0] *= 3e7; K[1] *= 2e-9; B[2] *= 5e5; B[0] *= 1e-6         # far outside the f16 range unscaled
    ref = np.conj(A.T)[None] @ K @ np.conj(np.swapaxes(B, 1, 2))
    out = J.correlate(K, A, B)
    for t in range(3):
        assert rel_err(out[t], ref[t]) < 5e-6
    # strongly non-uniform magnitudes inside one problem: small entries keep their relative accuracy budget
    K2 = _rand(rng, 1, 48, 200)
    K2[0, :, :100] *= 1e-4
    ref2 = np.conj(A.T)[None] @ K2 @ np.conj(np.swapaxes(B[1:2], 1, 2))
    assert rel_err(J.correlate(K2, A, B[1:2]), ref2) < 5e-6
    assert np.all(J.correlate(np.zeros((1, 48, 200), complex), A, B[:1]) == 0)
    assert np.all(J.synthesize(np.zeros((1, 40, 72), complex), A, B[:1]) == 0)


def test_split_f16_matches_the_fp32_mfma_path(force_h2):
    """Same inputs through both matrix-pipe paths: they agree to fp32 round-off."""
    import jstsp19_amd as J
    rng = np.random.default_rng(8)
    K, A, B = _rand(rng, 4, 64, 512), _rand(rng, 64, 64), _rand(rng, 4, 128, 512)
    h = J.correlate(K, A, B)
    os.environ["JSTSP_H2"] = "0"
    f = J.correlate(K, A, B)
    assert rel_err(h, f) < 3e-6


@pytest.mark.parametrize("name", ["proposed_small", "proposed_small_lowsnr", "proposed_refnative"])
def test_proposed_through_split_f16_matches_golden(force_h2, name):
    """GEMMs, the G_B applies and the SVT / convergence-error Grams all on the split-f16 kernels."""
    import jstsp19_amd as J
    from test_gpu_parity_proposed import _check
    g = load_golden(name)
    out = J.proposed_algorithm(g["subY"], g["Omega"], g["A"], g["B"], int(g["Imax"]), float(g["tau_Y"]),
                               float(g["tau_Z"]), float(g["rho"]), "approximate")
    _check(out, g, "approximate", nmse_tol=1e-6 if name == "proposed_refnative" else 2e-6)
    if name != "proposed_small_lowsnr":
        out = J.proposed_algorithm_angles(g["subY"], g["Omega"], g["indx_S"], g["A"], g["B"], int(g["Imax"]),
                                          float(g["tau_Y"]), float(g["tau_Z"]), float(g["rho"]), "approximate", 100)
        _check(out, g, "angles", nmse_tol=1e-6 if name == "proposed_refnative" else 2e-6)


def test_batched_equals_single_through_split_f16(force_h2):
    import torch
    import jstsp19_amd as J
    g = load_golden("proposed_refnative")
    args = lambda b: (np.stack([g["subY"]] * b), np.stack([g["Omega"]] * b), g["A"], np.stack([g["B"]] * b), 30,
                      [float(g["tau_Y"])] * b, [float(g["tau_Z"])] * b, [float(g["rho"])] * b, "approximate")
    S1, Y1, ce1 = J.proposed_algorithm(*args(1))
    S9, Y9, ce9 = J.proposed_algorithm(*args(9))
    for t in range(9):
        assert np.array_equal(S9[t], S1[0]) and np.array_equal(Y9[t], Y1[0])


def test_std_type_through_split_f16_matches_oracle(force_h2):
    """'std' (Alg. 1: v = U\\(L\\k) as G_A^-1 (A^H K B^H) G_B^-1) with G_B = B B^H and both big contractions on the
    split-f16 kernels."""
    import jstsp19_amd as J
    from oracle import solvers as O
    rng = np.random.default_rng(3)
    N, M, Gr, G2 = 40, 96, 24, 64
    r = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    A, B = r(N, Gr) / np.sqrt(N), r(G2, M) / np.sqrt(M)
    Om = (rng.random((N, M)) < 0.5).astype(float)
    args = (Om * r(N, M), Om, A, B, 15, 0.02, 0.01, 0.4, "std")
    So, Yo, _ = O.proposed_algorithm(*args)
    S, Y, _ = J.proposed_algorithm(*args)
    assert rel_err(S, So) < 1e-4 and rel_err(Y, Yo) < 1e-4


def test_split_f16_random_shapes(force_h2):
    """Random small shapes (including 1-sized dimensions) through both orientations of the packed operand."""
    import jstsp19_amd as J
    rng = np.random.default_rng(2024)
    for _ in range(24):
        N, M, Gr, G2 = (int(rng.integers(1, 90)), int(rng.integers(1, 150)), int(rng.integers(1, 70)), int(rng.integers(1, 140)))
        batch, shared = int(rng.integers(1, 10)), bool(rng.integers(0, 2))
        K, S = _rand(rng, batch, N, M), _rand(rng, batch, Gr, G2)
        A = _rand(rng, N, Gr)
        B = _rand(rng, G2, M) if shared else _rand(rng, batch, G2, M)
        Bb = np.broadcast_to(B, (batch, G2, M))
        ref_c = np.conj(A.T)[None] @ K @ np.conj(np.swapaxes(Bb, 1, 2))
        ref_s = A[None] @ S @ Bb
        assert rel_err(J.correlate(K, A, B), ref_c) < 5e-6, (N, M, Gr, G2, batch, shared)
        assert rel_err(J.synthesize(S, A, B), ref_s) < 5e-6, (N, M, Gr, G2, batch, shared)


@pytest.mark.parametrize("N,M,Gr,G2", [(7, 13, 5, 9), (100, 130, 70, 33), (64, 200, 64, 96)])
def test_proposed_ragged_and_two_row_tiles_through_split_f16(force_h2, N, M, Gr, G2):
    """Edge tiles, odd sizes, N > 64 (two row tiles of the a operand; Grams fall back to the fp32 kernel there)."""
    import jstsp19_amd as J
    from oracle import solvers as O
    rng = np.random.default_rng(N + M)
    r = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    A, B = r(N, Gr) / np.sqrt(N), r(G2, M) / np.sqrt(G2)
    S0 = np.zeros((Gr, G2), complex); S0[1, 2] = 3 + 1j; S0[Gr - 1, G2 - 2] = -2j
    Om = (rng.random((N, M)) < 0.5).astype(float)
    subY = Om * (A @ S0 @ B + 0.05 * r(N, M))
    args = (subY, Om, A, B, 20, 0.01, 0.02, 0.3, "approximate")
    So, Yo, ceo = O.proposed_algorithm(*args)
    S, Y, ce = J.proposed_algorithm(*args)
    check_below("hgemm_shapes.S", rel_err(S, So), TOL_S); check_below("hgemm_shapes.Y", rel_err(Y, Yo), TOL_S)
    check_below("hgemm_shapes.ce", ce_rel(ce, ceo), TOL_CE)


@pytest.mark.parametrize("M,G2,batch", [(2048, 2048, 32), (4096, 4096, 17), (16384, 256, 4), (4160, 2050, 31)])
def test_shared_dictionary_pairs_of_trials_match_numpy(force_h2, M, G2, batch):
    """hgemm_pair_kernel (one dictionary for the batch, N = 64, at least 256 workgroups: BASELINE configs[4]'s form of the two
    contractions - two trials per workgroup against the same dictionary fragments): even and ODD batch (the last pair holds one
    trial), column counts and contraction lengths that end inside a tile / a 64-term stage pair (4160, 2050), the second-level sums of a 4096-term contraction, trials of very different scale in one pair, against float64 numpy -
    and each trial bit-identical to the same trial in another batch position (pair partner and half of the pair changed)."""
    import jstsp19_amd as J
    rng = np.random.default_rng(M + batch)
    N = Gr = 64
    K, S = _rand(rng, batch, N, M), _rand(rng, batch, Gr, G2)
    K[1] *= 1e-6; S[1] *= 1e5; K[2] *= 3e4                    # (scales are per trial: a pair shares nothing but the dictionary)
    A, B = _rand(rng, N, Gr) / 8, _rand(rng, G2, M)
    Cg, Xg = J.correlate(K, A, B), J.synthesize(S, A, B)
    for t in range(batch):
        ref_c = np.conj(A.T) @ K[t] @ np.conj(B.T)
        ref_s = A @ S[t] @ B
        check_below("hgemm_pair.correlate", rel_err(Cg[t], ref_c), 5e-6)
        check_below("hgemm_pair.synthesize", rel_err(Xg[t], ref_s), 5e-6)
    if batch >= 4:
        perm = np.roll(np.arange(batch), 1)                     # trial t moves to position t + 1: other partner, other half
        Cp, Xp = J.correlate(K[perm], A, B), J.synthesize(S[perm], A, B)
        assert np.array_equal(Cp, Cg[perm]) and np.array_equal(Xp, Xg[perm])


def test_proposed_shared_pilots_three_kernel_iteration_through_the_pair_kernel(force_h2):
    """The solver's own use of the pair kernel: (A S) B with the packed a operand and the V2 update in its epilogue
    (proposed_algorithm.m:58,61,65), three-kernel iteration (JSTSP_FUSED=0), N = 64, one pilot set for 8 trials, M = 8192
    (4 pairs x 64 column tiles) - against the float64 oracle trial by trial, and batched == single (the single solve takes the
    per-trial kernel)."""
    import jstsp19_amd as J
    from oracle import solvers as O
    rng = np.random.default_rng(58)
    r = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    N, M, Gr, G2, b, Imax = 64, 8192, 64, 128, 8, 8
    A, B = r(N, Gr) / np.sqrt(N), r(G2, M) / np.sqrt(G2)
    Om = (rng.random((b, N, M)) < 0.25).astype(float)
    S0 = np.zeros((b, Gr, G2), complex)
    for t in range(b):
        S0[t, rng.integers(0, Gr, 5), rng.integers(0, G2, 5)] = r(5)
    subY = Om * (A @ S0 @ B + 0.05 * r(b, N, M))
    fro2 = (np.abs(subY) ** 2).sum((1, 2))
    tY, tZ, rho = 1.0 / fro2, np.full(b, 1e-2), np.full(b, 0.25)
    old = os.environ.get("JSTSP_FUSED")
    os.environ["JSTSP_FUSED"] = "0"
    try:
        S, Y, ce = J.proposed_algorithm(subY, Om, A, B, Imax, tY, tZ, rho, "approximate")
        S1, Y1, _ = J.proposed_algorithm(subY[5:6], Om[5:6], A, B, Imax, tY[5:6], tZ[5:6], rho[5:6], "approximate")
    finally:
        if old is None:
            os.environ.pop("JSTSP_FUSED", None)
        else:
            os.environ["JSTSP_FUSED"] = old
    for t in (0, 5, 7):
        So, Yo, ceo = O.proposed_algorithm(subY[t], Om[t], A, B, Imax, float(tY[t]), float(tZ[t]), float(rho[t]), "approximate")
        check_below("hgemm_pair.S", rel_err(S[t], So), TOL_S); check_below("hgemm_pair.Y", rel_err(Y[t], Yo), TOL_S)
        check_below("hgemm_pair.ce", ce_rel(ce[t], ceo), TOL_CE)
    check_below("hgemm_pair.batched_vs_single.S", rel_err(S1[0], S[5]), TOL_S)


def test_proposed_shared_pilots_both_contractions_through_the_pair_kernel():
    """K B^H (second-level sums, K packed once per iteration) AND (A S) B through hgemm_pair_kernel inside the solver, at a shape
    the default switches give it to: N = 64, G2 = 2048 (no one-pass iteration beyond 512), M = 4096, 32 trials on one pilot set
    (16 pairs x 16 / 32 column tiles) - K B^H feeds the cancellation Res = A^H Tc - R v (proposed_algorithm.m:47), which is where
    accumulation noise would show.  Three trials against the float64 oracle after 10 iterations; the support-restricted variant too."""
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
    for t in (0, 17, 31):
        So, Yo, ceo = O.proposed_algorithm(subY[t], Om[t], A, B, Imax, float(tY[t]), float(tZ[t]), float(rho[t]), "approximate")
        check_below("hgemm_pair.full.S", rel_err(S[t], So), TOL_S); check_below("hgemm_pair.full.Y", rel_err(Y[t], Yo), TOL_S)
        check_below("hgemm_pair.full.ce", ce_rel(ce[t], ceo), TOL_CE)
        check_below("hgemm_pair.full.nmse", abs(O.nmse_capped(S[t], S0[t]) - O.nmse_capped(So, S0[t])), TOL_NMSE)


def test_side_stream_overlap_is_bit_identical_through_the_split_f16_grams(force_h2):
    """JSTSP_OVERLAP=1 runs the next SVT preparation and the norm chain on side streams.  Same kernels, same
    arithmetic: the results must be bit-identical to the single-stream run, also when the side-stream Grams (which
    read the per-iteration operand maxima) are still running while the main stream starts the next iteration —
    the maxima are double-buffered by iteration parity for that."""
    import jstsp19_amd as J
    rng = np.random.default_rng(77)
    r = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    N, M, Gr, G2, b = 64, 768, 64, 128, 24
    A, B = r(N, Gr) / np.sqrt(N), r(b, G2, M) / np.sqrt(G2)
    Om = (rng.random((b, N, M)) < 0.3).astype(float)
    S0 = np.zeros((b, Gr, G2), complex); S0[:, 3, 7] = 2 - 1j; S0[:, 40, 100] = 1j
    subY = Om * (A @ S0 @ B + 0.05 * r(b, N, M))
    args = (subY, Om, A, B, 25, 1e-3, 2e-2, 0.3, "approximate")
    old = os.environ.get("JSTSP_OVERLAP")
    try:
        os.environ["JSTSP_OVERLAP"] = "0"
        S_ref, Y_ref, ce_ref = J.proposed_algorithm(*args)
        os.environ["JSTSP_OVERLAP"] = "1"
        for _ in range(3):
            S1, Y1, ce1 = J.proposed_algorithm(*args)
            assert np.array_equal(S1, S_ref) and np.array_equal(Y1, Y_ref) and np.array_equal(ce1, ce_ref)
    finally:
        if old is None:
            os.environ.pop("JSTSP_OVERLAP", None)
        else:
            os.environ["JSTSP_OVERLAP"] = old


def test_public_correlate_and_synthesize_keep_fp32_accuracy_per_row_under_dynamic_range():
    """One row of K (and one row of B) 1e-8 below the rest of the problem: the output row / column it produces depends on
    nothing else, so it must come out with fp32 relative accuracy - the split-f16 scale is per problem, the public entry
    points equilibrate the non-contracted indices by exact powers of two first (csrc/api_misc.hip)."""
    import torch
    import jstsp19_amd as J
    N, M, Gr, G2, b = 64, 4096, 64, 512, 2
    g = torch.Generator(device="cuda"); g.manual_seed(99)
    rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device="cuda", dtype=torch.float64),
                                   torch.randn(*s, generator=g, device="cuda", dtype=torch.float64))
    A = torch.eye(N, dtype=torch.complex128, device="cuda")              # keeps the rows of K B^H apart in the output
    K, B, S = rnd(b, N, M), rnd(b, G2, M), rnd(b, Gr, G2)
    K[:, 5, :] *= 1e-8
    B[:, 17, :] *= 1e-8
    cm = lambda x: J.colmajor(x.to(torch.complex64))
    K32, B32, S32, A32 = cm(K), cm(B), cm(S), cm(A)
    ref = A.conj().T @ K32.to(torch.complex128) @ B32.to(torch.complex128).conj().transpose(1, 2)
    out = J.correlate(K32, A32, B32).to(torch.complex128)
    err = (out - ref).abs()
    row = err.amax(dim=2) / ref.abs().amax(dim=2)                         # per output row (n)
    col = err.amax(dim=1) / ref.abs().amax(dim=1)                         # per output column (g)
    assert float(row.max()) < 1e-6 and float(col.max()) < 1e-6, (float(row.max()), float(col.max()))
    assert float(row[:, 5].max()) < 1e-6 and float(col[:, 17].max()) < 1e-6
    # synthesize: rows of A S (here: of S) and columns of B
    S[:, 9, :] *= 1e-8
    Bc = rnd(b, G2, M); Bc[:, :, 100] *= 1e-8
    S32, Bc32 = cm(S), cm(Bc)
    ref = A @ S32.to(torch.complex128) @ Bc32.to(torch.complex128)
    out = J.synthesize(S32, A32, Bc32).to(torch.complex128)
    err = (out - ref).abs()
    assert float((err.amax(dim=2) / ref.abs().amax(dim=2)).max()) < 1e-6
    assert float((err.amax(dim=1) / ref.abs().amax(dim=1)).max()) < 1e-6
    torch.cuda.synchronize()


def test_gradient_head_products_are_accurate_to_the_fp32_output():
    """csrc/hsmall.hip through the C ABI: Res = A'*Tc - RV and P1 = GA*Res (proposed_algorithm.m:47-48) against float64.
    These sums live in v-space, where the iteration never forgets an error, so they must be good to
    the rounding of their fp32 OUTPUT - including entries far below the operand maximum, which the three-way f16 split only
    represents if small third parts survive: a wide dynamic range of Tc rows and of A entries is part of the test."""
    import jstsp19_amd as J
    rng = np.random.default_rng(42)
    c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    N = Gr = 64
    G2, batch = 128, 3
    A = (c(N, Gr) / 8).astype(np.complex64)
    A[:, ::5] *= 2.0 ** -9                                    # columns far below the maximum
    GA = (A.astype(np.complex128).conj().T @ A.astype(np.complex128)).astype(np.complex64)
    Tc = c(batch, N, G2).astype(np.complex64)
    Tc[:, ::3, :] *= 2.0 ** -11                                # rows 2000 times smaller than the rest
    RV = (0.3 * c(batch, Gr, G2)).astype(np.complex64)
    A64, GA64, T64, R64 = (x.astype(np.complex128) for x in (A, GA, Tc, RV))
    res_ref = np.einsum("na,tng->tag", A64.conj(), T64) - R64
    Res, P1 = J.gradient_head(Tc, A, GA, RV)
    Res, P1 = np.asarray(Res), np.asarray(P1)
    # error relative to the magnitude of the terms of each sum (row-wise bound: |A|^T |Tc|), a few fp32 ulps
    bound = np.einsum("na,tng->tag", np.abs(A64), np.abs(T64)) + np.abs(R64)
    err = np.abs(Res - res_ref) / bound
    assert err.max() < 1.5e-7, float(err.max())
    assert np.sqrt(np.mean(err ** 2)) < 3e-8, float(np.sqrt(np.mean(err ** 2)))
    p1_ref = np.einsum("ab,tbg->tag", GA64, Res.astype(np.complex128))      # second product from the kernel's own Res
    b2 = np.einsum("ab,tbg->tag", np.abs(GA64), np.abs(Res.astype(np.complex128)))
    e2 = np.abs(P1 - p1_ref) / b2
    assert e2.max() < 1.5e-7 and np.sqrt(np.mean(e2 ** 2)) < 3e-8, (float(e2.max()), float(np.sqrt(np.mean(e2 ** 2))))
    # the small columns / rows came through with their own relative accuracy (not just that of the maximum)
    small = np.abs(res_ref[:, ::5, :])
    assert np.median(np.abs(Res[:, ::5, :] - res_ref[:, ::5, :]) / np.maximum(small, 1e-30)) < 1e-6
    with pytest.raises(J.JstspError):
        J.gradient_head(Tc[:, :32], A[:32], GA)               # N = 32 is not a shape of this kernel
