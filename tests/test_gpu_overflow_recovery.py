"""The fused pass (csrc/fused.hip) predicts the f16 scale of `k` (proposed_algorithm.m:43) from the previous iteration's
maximum.  A trial whose k outgrows the prediction must not poison anything: it is flagged, solved again by the
three-kernel iteration inside the same call, and the call returns what that iteration returns - for that trial only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(inp, Imax, env=None, want_ce=True, Omega=None):
    import torch
    import jstsp19_amd as J
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    try:
        r = J.proposed_algorithm(inp["subY"], inp["Omega"] if Omega is None else Omega, inp["A"], inp["B"], Imax,
                                 inp["tau_Y"].numpy(), inp["tau_Z"].numpy(), inp["rho"].numpy(), "approximate",
                                 want_ce=want_ce)
        torch.cuda.synchronize()
        n = J.default_context(0).last_fused_fallbacks()
    finally:
        for k in env:
            os.environ.pop(k, None)
    return [None if x is None else x.cpu().numpy() for x in r], n


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def _small():
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=16, Nr=64, L=8, T=32, Mr=8, snr_db=5.0)            # N=64, M=512, Gr=64, G2=128: fused-pass shape
    assert p.solver_shape == (64, 512, 64, 128)
    return build_trials(p, 0, 6, seed=41)


def test_forced_overflow_of_every_trial_is_recovered():
    """JSTSP_FUSED_KBACK=-20 puts the k scale 2^20 above what its previous maximum allows: every pass overflows for
    every trial.  The call must still return the three-kernel iteration's results, and say that it re-solved them."""
    inp = _small()
    (S0, Y0, c0), n0 = _solve(inp, 15, {"JSTSP_FUSED": "0"})
    (S1, Y1, c1), n1 = _solve(inp, 15)
    assert n0 == 0 and n1 == 0                                               # nothing to recover on healthy inputs
    (S2, Y2, c2), n2 = _solve(inp, 15, {"JSTSP_FUSED_KBACK": "-20"})
    assert n2 == 6
    assert np.all(np.isfinite(S2)) and np.all(np.isfinite(Y2))
    assert _rel(S2, S0) < 1e-6 and _rel(Y2, Y0) < 1e-6                       # the same kernels as JSTSP_FUSED=0
    fin = np.isfinite(c0)
    assert np.array_equal(np.isfinite(c2), fin) and np.max(np.abs(c2[fin] - c0[fin]) / np.abs(c0[fin])) < 1e-4
    assert _rel(S1, S0) < 1e-5                                               # and the healthy fused solve agrees anyway


def test_forced_overflow_in_the_window_kernel_is_recovered():
    """The same with a block-Toeplitz dictionary of block height 64, i.e. through fused_pass64_kernel (csrc/fused.hip)."""
    import jstsp19_amd as J
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=64, Nr=64, L=4, T=8, Mr=8, snr_db=5.0)
    assert p.solver_shape == (64, 512, 64, 256)
    inp = build_trials(p, 0, 5, seed=43)
    (S0, Y0, c0), _ = _solve(inp, 12, {"JSTSP_FUSED": "0"})
    (S1, Y1, c1), n1 = _solve(inp, 12)
    assert n1 == 0 and J.default_context(0).last_dictionary_block() == 64
    assert _rel(S1, S0) < 1e-5 and _rel(Y1, Y0) < 1e-5
    (S2, Y2, c2), n2 = _solve(inp, 12, {"JSTSP_FUSED_KBACK": "-20"})
    assert n2 == 5
    assert _rel(S2, S0) < 1e-6 and _rel(Y2, Y0) < 1e-6                       # the same kernels as JSTSP_FUSED=0
    fin = np.isfinite(c0)
    assert np.array_equal(np.isfinite(c2), fin) and np.max(np.abs(c2[fin] - c0[fin]) / np.abs(c0[fin])) < 1e-4


def test_one_trial_whose_k_jumps_is_re_solved_alone_and_matches_the_oracle():
    """A weight matrix Omega with entries just above -2*rho where nothing was sampled makes 1/(Omega + 2 rho) = 1e3 there
    (iK1 of proposed_algorithm.m:14-20 is defined for any Omega): X is zero at those entries after iteration 1 and
    1e3 x (rho Xs + ...) after iteration 2 - k grows by more than two orders of magnitude between two passes, in ONE trial."""
    import torch
    from oracle import solvers as O
    inp = _small()
    bad = 3
    inp["rho"] = torch.from_numpy(inp["rho"].numpy().astype(np.float32).astype(np.float64))   # (Omega + 2 rho exact in fp32)
    rho = float(inp["rho"][bad])
    Om = inp["Omega"].clone()
    zero = (Om[bad] == 0).nonzero()
    pick = zero[torch.linspace(0, len(zero) - 1, 7).long()]
    for r, c in pick.tolist():
        Om[bad, r, c] = -2.0 * rho + 1e-3
    # (negative weights make the iteration itself unstable: X grows 170-fold per iteration in the float64 oracle too, so
    #  the comparison stops at 6 iterations, where everything - including the squared norms of convergence_error - is
    #  still far inside the fp32 range)
    IM = 6
    (S1, Y1, c1), n1 = _solve(inp, IM, Omega=Om)
    assert n1 == 1                                                           # that trial, and only that trial
    assert np.all(np.isfinite(S1)) and np.all(np.isfinite(Y1))
    (S0, Y0, c0), _ = _solve(inp, IM, {"JSTSP_FUSED": "0"}, Omega=Om)
    assert np.array_equal(np.isfinite(c1), np.isfinite(c0)), np.argwhere(np.isfinite(c1) != np.isfinite(c0))[:8].tolist()
    assert np.all(np.isfinite(c0[:, 1:, :2])), np.argwhere(~np.isfinite(c0))[:8].tolist()
    assert _rel(S1[bad], S0[bad]) < 1e-6 and _rel(Y1[bad], Y0[bad]) < 1e-6
    for t in range(6):
        assert _rel(S1[t], S0[t]) < 1e-5
    # against the float64 oracle: the flagged trial and a healthy neighbour
    A_h = inp["A"].cpu().numpy().astype(np.complex128)
    for t in (bad, bad + 1):
        So, Yo, _ = O.proposed_algorithm(inp["subY"][t].cpu().numpy().astype(np.complex128),
                                         Om[t].cpu().numpy().astype(np.float64), A_h,
                                         inp["B"][t].cpu().numpy().astype(np.complex128), IM, float(inp["tau_Y"][t]),
                                         float(inp["tau_Z"][t]), float(inp["rho"][t]), "approximate", want_ce=False)
        assert _rel(S1[t], So) < 5e-4, t
    # two outputs only (no three-Gram pass): same recovery
    (S3, _, _), n3 = _solve(inp, IM, Omega=Om, want_ce=False)
    assert n3 == 1 and _rel(S3, S1) < 1e-5


def test_pipelined_host_call_equals_the_staged_one_and_recovers_per_half():
    """A JSTSP_HOST call of >= 128 problems runs as two halves on two internal contexts (proposed.hip): bit for bit the outputs of
    the single staged call, also when trials in BOTH halves overflow their predicted scale and are solved again."""
    import os
    import torch
    import jstsp19_amd as J
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
    inp = build_trials(p, 0, 128, seed=11)
    h = {k: inp[k].cpu().numpy() for k in ("subY", "Omega", "B")}
    A = inp["A"].cpu().numpy()
    hyp = [inp[k].numpy() for k in ("tau_Y", "tau_Z", "rho")]

    def run(env, Imax=12):
        os.environ.update(env)
        try:
            r = J.proposed_algorithm(h["subY"], h["Omega"], A, h["B"], Imax, *hyp, "approximate")
            n = J.default_context(0).last_fused_fallbacks()
        finally:
            for k in env:
                os.environ.pop(k, None)
        return [np.asarray(x) for x in r], n

    one, n1 = run({"JSTSP_HOST_PIPELINE": "0"})
    two, n2 = run({})
    assert n1 == 0 and n2 == 0
    for a, b in zip(one, two):
        assert a.tobytes() == b.tobytes()
    # every trial forced through the recovery: both halves re-solve theirs with the three-kernel iteration
    ref, _ = run({"JSTSP_FUSED": "0", "JSTSP_HOST_PIPELINE": "0"})
    rec, nrec = run({"JSTSP_FUSED_KBACK": "-20"})
    assert nrec == 128
    for a, b in zip(ref, rec):
        assert a.tobytes() == b.tobytes()


def test_pipelined_complex_double_host_call_equals_the_staged_one_and_recovers_per_half():
    """The MEX gateway's route (interleaved complex DOUBLES in host memory, jstsp_proposed_algorithm_c64) is pipelined the same
    way (c64.hip: each half uploaded and narrowed on its own context): bit for bit the staged call, with and without recovery."""
    import ctypes as C
    import os
    import jstsp19_amd as J
    from jstsp19_amd import _lib
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
    batch, Imax = 128, 12
    inp = build_trials(p, 0, batch, seed=11)
    lib, ctx = _lib.load(), J.default_context(0)
    colm = lambda a, dt: np.ascontiguousarray(np.swapaxes(a.cpu().numpy(), -1, -2).astype(dt))
    sy, om, b, a64 = colm(inp["subY"], np.complex128), colm(inp["Omega"], np.float64), colm(inp["B"], np.complex128), colm(inp["A"], np.complex128)
    N, M = inp["subY"].shape[1:]
    Gr, G2 = inp["A"].shape[1], inp["B"].shape[1]
    ty, ts, rh = (np.ascontiguousarray(inp[k].numpy(), dtype=np.float64) for k in ("tau_Y", "tau_Z", "rho"))
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))

    def run(env):
        S = np.empty(batch * Gr * G2, np.complex128); Y = np.empty(batch * N * M, np.complex128); ce = np.empty(batch * 3 * Imax, np.float64)
        os.environ.update(env)
        try:
            _lib.check(lib.jstsp_proposed_algorithm_c64(ctx.handle, N, M, Gr, G2, batch, vp(sy), vp(om), vp(a64), 0, vp(b), G2 * M, Imax, dp(ty),
                                                        dp(ts), dp(rh), 0, None, vp(S), vp(Y), vp(ce), 0), "proposed_algorithm_c64")
            n = ctx.last_fused_fallbacks()
        finally:
            for k in env:
                os.environ.pop(k, None)
        return (S, Y, ce), n

    one, n1 = run({"JSTSP_HOST_PIPELINE": "0"})
    two, n2 = run({})
    assert n1 == 0 and n2 == 0
    for x, y in zip(one, two):
        assert x.tobytes() == y.tobytes()
    assert np.isfinite(one[0]).all() and np.count_nonzero(one[0]) > 0
    ref, _ = run({"JSTSP_FUSED": "0", "JSTSP_HOST_PIPELINE": "0"})
    rec, nrec = run({"JSTSP_FUSED_KBACK": "-20"})
    assert nrec == batch
    for x, y in zip(ref, rec):
        assert x.tobytes() == y.tobytes()
