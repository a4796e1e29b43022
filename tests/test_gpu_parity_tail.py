"""The accuracy statement of include/jstsp.h - NMSE of the HIP path within 1e-6 of the float64 solve - over THOUSANDS of
full-size trials, not a handful.

tests/golden/fullsize_port.npz holds the float64 side (oracle/cpu_port.cpp, 5.8 core-seconds per trial: computed once on a
GPU box's host by tools/parity_tail.py, converted by tests/golden/make_fullsize_port_fixture.py): per trial the NMSE
(plot_errorVSsnr.m:138-141) and convergence_error of proposed_algorithm.m:35-69 at BASELINE configs[1]'s shape
(N=64, M=4096, Gr=64, G2=512, Imax=100) for
  * the 256 trials of the bench workload (5 dB),
  * 10 SNR points (-15:3:12 dB) x 256 trials of the configs[3] sweep (plot_errorVSsnr.m:48-51),
  * proposed_algorithm_angles on 64 trials at -15, 0 and 12 dB,
keyed by the library's counter-based generator (seed, sweep index, trial index) and pinned by a fingerprint of the inputs the
generator must reproduce.  The HIP side is recomputed here (a millisecond per trial) and compared trial by trial."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6                      # |NMSE_hip - NMSE_float64| per trial (BASELINE.json north_star; include/jstsp.h)
IMAX = 100


from oracle.fullsize_fixture import fixture, solve_group  # noqa: E402
from conftest import check_below, TOL_CE  # noqa: E402


def check_ce(fx, group, rows, ces):
    """convergence_error against the float64 solve for the rows the fixture keeps it for (conftest.TOL_CE relative, finite pattern equal)."""
    ce_rows = fx[group + "/ce_rows"]
    pos = {int(r): k for k, r in enumerate(ce_rows)}
    n = 0
    for k, r in enumerate(rows):
        if int(r) not in pos:
            continue
        ref = fx[group + "/ce_port"][pos[int(r)]].astype(np.float64)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(ces[k]), fin)
        check_below("fullsize_fixture.ce." + group, np.max(np.abs(ces[k][fin] - ref[fin]) / np.abs(ref[fin])), TOL_CE)
        n += 1
    return n


@pytest.mark.parametrize("want_ce", [True, False], ids=["three_outputs", "two_outputs"])
def test_64_trials_at_three_snr_points_both_solvers(want_ce):
    """>= 64 full-size trials at -15, 0 and 12 dB, proposed_algorithm and proposed_algorithm_angles, with and without
    convergence_error (the two-output call takes another code path: Z stored by the pass, no norm Grams)."""
    fx = fixture()
    for group, angles in (("sweep_proposed", False), ("sweep_angles", True)):
        snr, trial = fx[group + "/snr_db"], fx[group + "/trial"]
        for db in (-15.0, 0.0, 12.0):
            rows = np.nonzero((snr == db) & (trial < 64))[0]
            assert len(rows) == 64
            nmse, ces = solve_group(fx, group, rows, want_ce=want_ce, angles=angles, chunk=64)
            d = np.abs(nmse - fx[group + "/nmse_port"][rows])
            assert d.max() < TOL, (group, db, float(d.max()), int(rows[np.argmax(d)]))
            if want_ce:
                assert check_ce(fx, group, rows, ces) >= 16


def test_all_2816_trials_of_the_bench_batch_and_the_snr_sweep():
    """Every trial the fixture holds for proposed_algorithm: the bench's 256 and the sweep's 10 x 256 (three-output call).
    Also the distribution: rms below a third of the tolerance."""
    fx = fixture()
    worst = []
    for group in ("bench_proposed", "sweep_proposed"):
        n = len(fx[group + "/nmse_port"])
        nmse, ces = solve_group(fx, group, np.arange(n), want_ce=True, angles=False)
        d = nmse - fx[group + "/nmse_port"]
        worst.append((group, float(np.abs(d).max()), float(np.sqrt(np.mean(d ** 2)))))
        assert np.abs(d).max() < TOL, worst
        assert np.sqrt(np.mean(d ** 2)) < TOL / 3, worst
        assert check_ce(fx, group, np.arange(n), ces) >= 160


def test_sweep_runner_at_the_configs3_shape_against_the_fixture():
    """BASELINE configs[3] at its own shape through the sweep runner (plot_errorVSsnr.m:48-51,170: N=64, M=4096, two SNR points x
    16 realisations, both solvers, two-output calls, NMSE by the library's spectral-norm kernel): per-trial values against the
    float64 fixture, and the merged call (trials of both points in one solver call) against one call per point."""
    import torch
    from jstsp19_amd.montecarlo import run_sweep
    from jstsp19_amd.system_model import SweepParams
    fx = fixture()
    base = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8)
    snrs, nt = [-15, -12], 16                                # sweep indices 0 and 1 of the fixture
    s1, s2 = [], []
    mean = run_sweep(base, snrs, nt, Imax=IMAX, batch=nt, samples=s1)
    per_point = run_sweep(base, snrs, nt, Imax=IMAX, batch=nt, samples=s2, merge=False)
    torch.cuda.synchronize()
    assert mean.shape == (2, 2) and torch.equal(mean, per_point)          # what else is in the batch does not change a trial
    for pt in range(2):
        assert torch.equal(s1[pt], s2[pt])
        m = (fx["sweep_proposed/sweep_idx"] == pt) & (fx["sweep_proposed/trial"] < nt)
        ref = fx["sweep_proposed/nmse_port"][m]
        assert np.array_equal(fx["sweep_proposed/trial"][m], np.arange(nt)) and fx["sweep_proposed/snr_db"][m][0] == snrs[pt]
        assert np.abs(s1[pt][:, 0].numpy() - ref).max() < TOL, (pt, float(np.abs(s1[pt][:, 0].numpy() - ref).max()))
        assert abs(float(mean[pt, 0]) - ref.mean()) < TOL
    ma = (fx["sweep_angles/sweep_idx"] == 0) & (fx["sweep_angles/trial"] < nt)
    assert np.abs(s1[0][:, 1].numpy() - fx["sweep_angles/nmse_port"][ma]).max() < TOL


# ---- the HELD-OUT fixture (round 5): another generator seed (20260105), 10 SNR points x 256 proposed_algorithm trials and
#      10 x 128 proposed_algorithm_angles trials.  tests/golden/fullsize_port.npz above is the set the round-4 defaults (how often
#      R v is recomputed, which products run in float64) were CHOSEN on.  Round 5 evaluated BOTH the round-4 and the round-5
#      defaults on this set (profiles/r05_parity_heldout_r04default.json), so it has informed a decision: since round 6 it is the
#      second VALIDATION set, and the held-out set proper is HELDOUT2 below (generated once, after the numerics were frozen).
HELDOUT = "fullsize_port_heldout"


def _heldout():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", HELDOUT + ".npz")
    if not os.path.exists(path):
        pytest.skip("held-out fixture not generated")
    fx = fixture(HELDOUT)
    assert int(fx["sweep_proposed/seed"][0]) != 20190913
    return fx


def test_heldout_three_output_calls_128_trials_per_point_both_solvers():
    """The round-5 held-out set, three-output call: the first 128 trials of each of the 10 SNR points of proposed_algorithm (1280 of its
    2560; the whole set is the committed measurement profiles/r05_parity_heldout_and_setA.json, and the round-6 set below is solved in
    full) + all 1280 proposed_algorithm_angles solves against float64.  The accuracy statement (max < 1e-6) and the distribution (rms
    below a third of it); convergence_error on the rows kept."""
    fx = _heldout()
    worst = []
    for group, angles, nmin in (("sweep_proposed", False, 1280), ("sweep_angles", True, 1280)):
        rows = np.nonzero(fx[group + "/trial"] < 128)[0]
        n = len(rows)
        assert n >= nmin
        nmse, ces = solve_group(fx, group, rows, want_ce=True, angles=angles)
        d = nmse - fx[group + "/nmse_port"][rows]
        worst.append((group, float(np.abs(d).max()), float(np.sqrt(np.mean(d ** 2))), int(rows[np.argmax(np.abs(d))])))
        assert np.abs(d).max() < TOL, worst
        assert np.sqrt(np.mean(d ** 2)) < TOL / 3, worst
        assert check_ce(fx, group, rows, ces) >= 300
    import jstsp19_amd as J
    assert J.default_context(0).last_lanczos_mismatches() == 0


def test_heldout_two_output_calls_64_trials_per_point_both_solvers():
    """The two-output call (no convergence_error: Z stored by the pass, no norm Grams - another code path) on 64 held-out trials
    at each of the 10 SNR points, both solvers."""
    fx = _heldout()
    for group, angles in (("sweep_proposed", False), ("sweep_angles", True)):
        rows = np.nonzero(fx[group + "/trial"] < 64)[0]
        assert len(rows) == 640
        nmse, _ = solve_group(fx, group, rows, want_ce=False, angles=angles, chunk=64)
        d = np.abs(nmse - fx[group + "/nmse_port"][rows])
        assert d.max() < TOL, (group, float(d.max()), int(rows[np.argmax(d)]))


# ---- the SECOND held-out fixture (round 6): generator seed 20261003, 10 x 256 proposed_algorithm + 10 x 128 proposed_algorithm_angles
#      trials, float64 side by oracle/cpu_port.cpp on a GPU box's host (tools/parity_tail.py, recipe in
#      tests/golden/make_fullsize_port_fixture.py).  Generated ONCE, after the last change of the round that touches S (the round's
#      later changes - norm Grams on the high f16 plane - touch convergence_error(:,1:2) only); no default was chosen or re-chosen
#      after looking at it.  The contract is stated as a maximum AND as distribution figures (rms, 99th percentile).
HELDOUT2 = "fullsize_port_heldout2"


def _heldout2():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", HELDOUT2 + ".npz")
    if not os.path.exists(path):
        pytest.skip("second held-out fixture not generated")
    fx = fixture(HELDOUT2)
    assert int(fx["sweep_proposed/seed"][0]) not in (20190913, 20260105)
    return fx


@pytest.mark.parametrize("want_ce", [True, False], ids=["three_outputs", "two_outputs"])
def test_heldout2_all_trials_both_call_forms_both_solvers(want_ce):
    """Every trial of the second held-out set, both call forms (the drivers make the two-output one, plot_errorVSsnr.m:137): 2560
    proposed_algorithm + 1280 proposed_algorithm_angles solves against float64.  max < 1e-6 (the statement), rms < 2e-7 and the 99th
    percentile < 6e-7 (the distribution: measured 1.4e-7 / 4.5e-7 on the earlier sets)."""
    from conftest import check_below
    fx = _heldout2()
    for group, angles, nmin in (("sweep_proposed", False, 2560), ("sweep_angles", True, 1280)):
        n = len(fx[group + "/nmse_port"])
        assert n >= nmin
        nmse, ces = solve_group(fx, group, np.arange(n), want_ce=want_ce, angles=angles)
        d = nmse - fx[group + "/nmse_port"]
        tag = "heldout2.%s.%s" % (group, "ce" if want_ce else "noce")
        check_below(tag + ".max", np.abs(d).max(), TOL)
        check_below(tag + ".rms", np.sqrt(np.mean(d ** 2)), 2e-7)
        check_below(tag + ".p99", np.quantile(np.abs(d), 0.99), 6e-7)
        if want_ce:
            assert check_ce(fx, group, np.arange(n), ces) >= 300
    import jstsp19_amd as J
    assert J.default_context(0).last_lanczos_mismatches() == 0
