"""The dictionaries the reference's drivers build (errorVSsnr.m:36-47: the pilot frame delayed by ld samples under every
transmit steering vector) are block-Toeplitz, B(ld Gt + g, m) == B(g, m - ld) for m >= ld.  The fused pass probes that
(csrc/fused.hip: exact comparison of every entry) and then streams the first block only.  What must hold:
  * the probe finds the structure in what the input builders produce, at every block height, and only there;
  * JSTSP_TOEPLITZ=1 (compact HBM image, full LDS tile): results BIT-identical to the unstructured path (=0);
  * default (=2: block height 64 takes the window kernel, whose LDS tile is the window of block 0 and which applies the
    leading columns m < ld of block ld as fp32 corrections): bit-identical when those columns are zero, fp32-equivalent
    otherwise;
  * the leading columns (not covered by the property) are honoured on both;
  * a dictionary without the structure - one entry changed - takes the full image and is solved as before."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(inp, Imax, env=None, B=None, want_ce=True):
    import torch
    import jstsp19_amd as J
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    try:
        r = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"] if B is None else B, Imax,
                                 inp["tau_Y"].numpy(), inp["tau_Z"].numpy(), inp["rho"].numpy(), "approximate",
                                 want_ce=want_ce)
        torch.cuda.synchronize()
        ctx = J.default_context(0)
        gt, nfb = ctx.last_dictionary_block(), ctx.last_fused_fallbacks()
    finally:
        for k in env:
            os.environ.pop(k, None)
    return [None if x is None else x.cpu().numpy() for x in r], gt, nfb


def _same(r1, r0):
    for a, b in zip(r1, r0):
        if a is None or b is None:
            assert a is None and b is None
            continue
        assert a.tobytes() == b.tobytes(), float(np.nanmax(np.abs(a - b)))


def _close(r1, r0, tol=2e-5):
    """fp32-equivalent: S, Y relative to their maximum; convergence_error per entry"""
    for a, b in zip(r1[:2], r0[:2]):
        assert np.max(np.abs(a - b)) <= tol * np.max(np.abs(b)), float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    if r0[2] is not None:
        fin = np.isfinite(r0[2])
        assert np.array_equal(np.isfinite(r1[2]), fin)
        assert np.max(np.abs(r1[2][fin] - r0[2][fin]) / np.abs(r0[2][fin])) < 2e-4


def _params(Nt, L, T):
    from jstsp19_amd.system_model import SweepParams
    return SweepParams(Nt=Nt, Nr=64, L=L, T=T, Mr=8, snr_db=5.0)


@pytest.mark.parametrize("Nt,L,T,batch", [(16, 8, 32, 6),      # Gt = 16, G2 = 128, M = 512
                                          (32, 8, 16, 5),      # Gt = 32, G2 = 256, M = 512
                                          (32, 12, 24, 3),     # Gt = 32, G2 = 384, M = 768 (three 128-row groups)
                                          (64, 4, 8, 5),       # Gt = 64, G2 = 256
                                          (128, 4, 4, 3),      # Gt = 128, G2 = 512
                                          (16, 32, 32, 3)])    # Gt = 16, 32 delays: the halo spans a whole tile
def test_probe_finds_the_block_and_results_are_bit_identical(Nt, L, T, batch):
    from jstsp19_amd.system_model import build_trials
    p = _params(Nt, L, T)
    assert p.solver_shape[0] == 64 and p.solver_shape[3] == L * Nt
    inp = build_trials(p, 0, batch, seed=77)
    r1, gt1, n1 = _solve(inp, 12, {"JSTSP_TOEPLITZ": "1"})
    r0, gt0, n0 = _solve(inp, 12, {"JSTSP_TOEPLITZ": "0"})
    assert gt1 == Nt and gt0 == 0 and n1 == 0 and n0 == 0
    assert np.all(np.isfinite(r1[0]))
    _same(r1, r0)
    r2, gt2, n2 = _solve(inp, 12)
    assert gt2 == Nt and n2 == 0
    # default (2): G_B assembled in float64 from its first block row (round 4) and, for block height 64, the window kernel:
    # equal to the unstructured path to rounding, not bit for bit
    _close(r2, r0)


def test_headline_shape_shared_and_per_trial_pilots():
    from jstsp19_amd.system_model import build_trials
    p = _params(64, 8, 64)                                       # N = 64, M = 4096, G2 = 512: BASELINE.json configs[1]
    assert p.solver_shape == (64, 4096, 64, 512)
    inp = build_trials(p, 0, 9, seed=5)
    r1, gt1, _ = _solve(inp, 8, {"JSTSP_TOEPLITZ": "1"})
    r0, gt0, _ = _solve(inp, 8, {"JSTSP_TOEPLITZ": "0"})
    assert gt1 == 64 and gt0 == 0
    _same(r1, r0)
    r2, gt2, _ = _solve(inp, 8)
    assert gt2 == 64
    _close(r2, r0)
    # leading columns zero: the window kernel issues the same products on the same fragments - the same bits
    Bz = inp["B"].clone()
    for ld in range(1, 8):
        Bz[:, ld * 64:(ld + 1) * 64, :ld] = 0
    # (what still differs from the unstructured path is G_B, assembled from its first block row: fp32-equivalent, not bit-identical)
    z3, gt, _ = _solve(inp, 8, B=Bz)
    assert gt == 64
    z4, _, _ = _solve(inp, 8, {"JSTSP_TOEPLITZ": "0"}, B=Bz)
    _close(z3, z4)
    sh = build_trials(p, 0, 9, seed=5, shared_pilots=True)
    s0, _, _ = _solve(sh, 8, {"JSTSP_TOEPLITZ": "0"}, B=sh["B"][0])
    for env, cmp in (({"JSTSP_TOEPLITZ": "1"}, _same), (None, _close)):
        s1, gt, _ = _solve(sh, 8, env, B=sh["B"][0])
        assert gt == 64
        cmp(s1, s0)
    w0, _, _ = _solve(inp, 8, {"JSTSP_TOEPLITZ": "0"}, want_ce=False)
    for env, cmp in (({"JSTSP_TOEPLITZ": "1"}, _same), (None, _close)):   # the 2-output call (no convergence_error)
        w1, gt, _ = _solve(inp, 8, env, want_ce=False)
        assert gt == 64
        cmp(w1, w0)


def test_window_kernel_with_fewer_delays_and_modified_leading_columns():
    """Block height 64 with L = 2, 4, 6 delays (G2 = 128, 256, 384), leading columns replaced by arbitrary values."""
    import torch
    from jstsp19_amd.system_model import build_trials
    for L, T in ((2, 8), (4, 8), (6, 12)):
        p = _params(64, L, T)
        inp = build_trials(p, 0, 5, seed=21 + L)
        B = inp["B"].clone()
        g = torch.Generator(device=B.device); g.manual_seed(L)
        for ld in range(1, L):
            blk = B[:, ld * 64:(ld + 1) * 64, :ld]
            B[:, ld * 64:(ld + 1) * 64, :ld] = 0.5 * torch.complex(torch.randn(blk.shape, generator=g, device=B.device),
                                                                   torch.randn(blk.shape, generator=g, device=B.device))
        r0, _, _ = _solve(inp, 10, {"JSTSP_TOEPLITZ": "0"}, B=B)
        r1, gt1, _ = _solve(inp, 10, {"JSTSP_TOEPLITZ": "1"}, B=B)
        r2, gt2, _ = _solve(inp, 10, B=B)
        assert gt1 == 64 and gt2 == 64
        _same(r1, r0)
        _close(r2, r0)
        ref, _, _ = _solve(inp, 10, {"JSTSP_TOEPLITZ": "0"})
        assert np.max(np.abs(ref[0] - r0[0])) > 1e-3 * np.max(np.abs(r0[0]))      # (the changed columns do reach the result)


@pytest.mark.parametrize("T", [512, 544])                        # 16 tiles in 4 column ranges; 17 tiles in one
def test_gaussian_pilots_of_the_training_model(T):
    from jstsp19_amd.system_model import TrainingParams, build_trials_training
    p = TrainingParams(Nt=16, Nr=64, L=8, T=T, ratio=1.0)
    assert p.solver_shape == (64, T, 64, 128)
    inp = build_trials_training(p, 0, 4, seed=3)
    inp["tau_Y"], inp["tau_Z"] = inp["tau_X"], inp["tau_S"]      # (the approx driver's names: plot_errorVSsnr_approx.m:50-51)
    # (shorter frames do not take the split-f16 path at all: hgemm.hip use_hgemm)
    r1, gt1, _ = _solve(inp, 10, {"JSTSP_TOEPLITZ": "1"})
    r0, _, _ = _solve(inp, 10, {"JSTSP_TOEPLITZ": "0"})
    assert gt1 == 16
    _same(r1, r0)
    r2, gt2, _ = _solve(inp, 10)                                  # default: G_B assembled from its first block row
    assert gt2 == 16
    _close(r2, r0)


def test_leading_columns_are_free_and_one_changed_entry_ends_the_structure():
    import torch
    from jstsp19_amd.system_model import build_trials
    p = _params(16, 8, 32)
    inp = build_trials(p, 0, 4, seed=9)
    G2, M, Gt = 128, 512, 16
    B = inp["B"].clone()                                         # [batch, G2, M] view of column-major storage
    g = torch.Generator(device=B.device); g.manual_seed(1)
    for ld in range(1, 8):                                       # columns m < ld of block ld: anything goes
        blk = B[:, ld * Gt:(ld + 1) * Gt, :ld]
        B[:, ld * Gt:(ld + 1) * Gt, :ld] = 0.3 * torch.complex(torch.randn(blk.shape, generator=g, device=B.device),
                                                               torch.randn(blk.shape, generator=g, device=B.device))
    assert B.stride() == inp["B"].stride()
    r1, gt1, _ = _solve(inp, 10, {"JSTSP_TOEPLITZ": "1"}, B=B)
    r0, _, _ = _solve(inp, 10, {"JSTSP_TOEPLITZ": "0"}, B=B)
    assert gt1 == Gt
    _same(r1, r0)
    r4, gt4, _ = _solve(inp, 10, B=B)                                # default: G_B from its first block row + the leading columns
    assert gt4 == Gt
    _close(r4, r0)
    ref, _, _ = _solve(inp, 10, {"JSTSP_TOEPLITZ": "0"})
    assert np.max(np.abs(r1[0] - ref[0])) > 1e-3 * np.max(np.abs(ref[0]))     # (the changed columns do reach the result)
    # one entry of one trial's dictionary off by one ulp, deep inside: no structure any more, same results as without probe
    B2 = inp["B"].clone()
    v = torch.view_as_real(B2)
    v[2, 5 * Gt + 3, 300, 0] = torch.nextafter(v[2, 5 * Gt + 3, 300, 0], torch.tensor(10.0, device=B.device))
    r2, gt2, _ = _solve(inp, 10, B=B2)
    r3, _, _ = _solve(inp, 10, {"JSTSP_TOEPLITZ": "0"}, B=B2)
    assert gt2 == 0
    _same(r2, r3)
    # a random dictionary
    B3 = 0.1 * torch.complex(torch.randn(B.shape, generator=g, device=B.device), torch.randn(B.shape, generator=g, device=B.device))
    B3 = B3.permute(0, 2, 1).contiguous().permute(0, 2, 1)       # column-major like the builder's
    _, gt3, _ = _solve(inp, 4, B=B3)
    assert gt3 == 0


@pytest.mark.parametrize("L,T,batch,Imax", [(8, 1, 1, 2),        # one trial, one tile per... M = 64: 2 tiles, ONE pass
                                            (8, 2, 7, 3),        # batch not a multiple of the 8 XCDs, M = 128: 4 tiles in 4 ranges
                                            (2, 17, 3, 5),       # M = 1088: 34 tiles in 2 ranges
                                            (8, 11, 19, 4),      # M = 704: 22 tiles in 2 ranges, 19 trials
                                            (4, 5, 300, 3)])     # more trials than CUs
def test_window_kernel_at_odd_batch_sizes_and_frame_lengths(L, T, batch, Imax):
    from jstsp19_amd.system_model import build_trials
    import jstsp19_amd as J
    p = _params(64, L, T)
    inp = build_trials(p, 0, batch, seed=100 + batch)
    os.environ["JSTSP_H2"] = "2"                                 # (frames this short would not take the split-f16 path otherwise)
    try:
        r0, _, _ = _solve(inp, Imax, {"JSTSP_TOEPLITZ": "0"})
        r1, gt1, n1 = _solve(inp, Imax, {"JSTSP_TOEPLITZ": "1"})
        r2, gt2, n2 = _solve(inp, Imax)
        w0, _, _ = _solve(inp, Imax, {"JSTSP_TOEPLITZ": "0"}, want_ce=False)
        w2, gtw, _ = _solve(inp, Imax, want_ce=False)
    finally:
        os.environ.pop("JSTSP_H2", None)
    assert gt1 == 64 and gt2 == 64 and gtw == 64 and n1 == 0 and n2 == 0
    _same(r1, r0)
    _close(r2, r0)
    _close(w2, w0)


def _host_solve(inp, Imax, env, B=None, c64=False):
    """The JSTSP_HOST call through the C ABI (numpy arrays in and out: what a MEX gateway passes); c64: MATLAB's doubles."""
    import ctypes as C
    import jstsp19_amd as J
    from jstsp19_amd import _lib
    env = dict(env)
    for k, v in env.items():
        os.environ[k] = v
    try:
        cd, rd = (np.complex128, np.float64) if c64 else (np.complex64, np.float32)
        f = lambda x, dt: np.ascontiguousarray(np.swapaxes(x.cpu().numpy(), -1, -2)).astype(dt)      # column-major bytes, trial slowest
        Bt = inp["B"] if B is None else B
        sy, om, a, b = f(inp["subY"], cd), f(inp["Omega"], rd), f(inp["A"], cd), f(Bt, cd)
        batch, N, M = inp["subY"].shape
        Gr, G2 = inp["A"].shape[-1], Bt.shape[1]
        S, Y = np.empty(batch * Gr * G2, cd), np.empty(batch * N * M, cd)
        ce = np.empty(batch * 3 * Imax, np.float64)
        p = lambda z: z.ctypes.data_as(C.c_void_p)
        dp = lambda z: np.ascontiguousarray(z.numpy(), np.float64)
        ty, ts, rh = dp(inp["tau_Y"]), dp(inp["tau_Z"]), dp(inp["rho"])
        ctx = J.default_context(0)
        fn = getattr(ctx._lib, "jstsp_proposed_algorithm_" + ("c64" if c64 else "c32"))
        d = lambda z: z.ctypes.data_as(C.POINTER(C.c_double))
        _lib.check(fn(ctx.handle, N, M, Gr, G2, batch, p(sy), p(om), p(a), 0, p(b), G2 * M, Imax, d(ty), d(ts), d(rh), 0, None,
                      p(S), p(Y), p(ce), 0), "proposed")
        gt = ctx.last_dictionary_block()
    finally:
        for k in env:
            os.environ.pop(k, None)
    return [S, Y, ce], gt


@pytest.mark.parametrize("c64", [False, True], ids=["c32", "c64"])
@pytest.mark.parametrize("Nt,L,T,batch", [(64, 8, 4, 5), (16, 8, 32, 6), (32, 12, 24, 3)])
def test_host_side_compaction_is_bit_identical_and_falls_back_on_the_first_mismatch(Nt, L, T, batch, c64):
    """JSTSP_HOST: the dictionary is tested on the host while it is staged and uploaded as first block + leading columns
    (csrc/hostpack.hip; JSTSP_HOST_COMPACT=2 forces the route at these small sizes).  Same results, bit for bit, as the plain
    upload (=0) - with free leading columns, and for a dictionary with one entry off by one ulp (the plain upload is taken)."""
    import torch
    from jstsp19_amd.system_model import build_trials
    p = _params(Nt, L, T)
    inp = build_trials(p, 0, batch, seed=31)
    B = inp["B"].clone()
    g = torch.Generator(device=B.device); g.manual_seed(2)
    for ld in range(1, L):                                       # columns m < ld of block ld: anything goes
        blk = B[:, ld * Nt:(ld + 1) * Nt, :ld]
        B[:, ld * Nt:(ld + 1) * Nt, :ld] = 0.3 * torch.complex(torch.randn(blk.shape, generator=g, device=B.device),
                                                               torch.randn(blk.shape, generator=g, device=B.device))
    for Bt in (None, B):
        r0, gt0 = _host_solve(inp, 8, {"JSTSP_HOST_COMPACT": "0"}, B=Bt, c64=c64)
        r1, gt1 = _host_solve(inp, 8, {"JSTSP_HOST_COMPACT": "2"}, B=Bt, c64=c64)
        assert gt0 == Nt and gt1 == Nt
        _same(r1, r0)
        # JSTSP_HOST_THREADS: the host-side test / compaction on one thread and on three (a count that does not divide the trials)
        for nthr in ("1", "3"):
            r1t, gt1t = _host_solve(inp, 8, {"JSTSP_HOST_COMPACT": "2", "JSTSP_HOST_THREADS": nthr}, B=Bt, c64=c64)
            assert gt1t == Nt
            _same(r1t, r0)
    B2 = inp["B"].clone()
    v = torch.view_as_real(B2)
    v[batch - 1, 3 * Nt + 1, 40, 1] = torch.nextafter(v[batch - 1, 3 * Nt + 1, 40, 1], torch.tensor(10.0, device=B.device))
    r2, gt2 = _host_solve(inp, 8, {"JSTSP_HOST_COMPACT": "2"}, B=B2, c64=c64)
    r3, gt3 = _host_solve(inp, 8, {"JSTSP_HOST_COMPACT": "0"}, B=B2, c64=c64)
    assert gt2 == 0 and gt3 == 0
    _same(r2, r3)
