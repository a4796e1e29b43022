"""The only recorded outputs of the COMMITTED reference code: results/errorVSsnr_angles.fig, 44 numbers = 4 algorithms x
11 SNR points, ONE unseeded realisation each (tests/golden/errorVSsnr_angles_published.json; extraction script next to
it).  A single draw cannot be reproduced, but it must look like a draw from OUR output distribution at the same
parameters: for every published point the HIP sweep's per-trial NMSE values (512 realisations) must bracket it within
their [0.5 %, 99.5 %] quantiles, for at least 40 of the 44 points (with 44 independent draws and a 1 % two-sided tail,
more than 4 misses has probability 1e-4 if the distributions agree).

What this pins: that the system model + solvers + NMSE definition produce the published values' distribution (location and
spread per SNR point and algorithm, including where the capped value 1 is reached).  What it does not pin: any individual
output - parity against the reference stays "unpinned" (DESIGN.md section 6)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
COLUMN = {"Proposed": 0, "Proposed with angle information": 1, "VAMP": 3, "MMV-OMP": 4}   # run_points columns


def test_published_single_trial_curves_are_draws_from_our_distribution():
    import torch
    from jstsp19_amd.montecarlo import driver, run_points
    with open(os.path.join(HERE, "golden", "errorVSsnr_angles_published.json")) as f:
        pub = json.load(f)
    d = driver("errorVSsnr")                                    # plot_errorVSsnr.m:8-25 at its own parameters
    assert d["values"] == [int(v) for v in pub["snr_db"]]
    samples = []
    mean = run_points(d["points"], 512, Imax=d["Imax"], numOfnz=d["numOfnz"], baselines=True, batch=512, samples=samples)
    torch.cuda.synchronize()
    assert len(samples) == 11 and all(s.shape == (512, 5) for s in samples)
    inside, report = 0, []
    for name, col in COLUMN.items():
        for i, y in enumerate(pub["series"][name]):
            v = samples[i][:, col].numpy()
            assert np.all(np.isfinite(v)), (name, i)
            lo, hi = np.quantile(v, 0.005), np.quantile(v, 0.995)
            ok = lo - 1e-12 <= y <= hi + 1e-12
            inside += ok
            report.append((name, pub["snr_db"][i], y, float(lo), float(np.median(v)), float(hi), bool(ok)))
    misses = [r for r in report if not r[-1]]
    assert inside >= 40, "published points outside our [0.5 %%, 99.5 %%] range: %s" % (misses,)
    # the figure's own structural facts: MMV-OMP sits at the cap 1 at low SNR and drops below it at high SNR; the genie
    # support ('angles') is better than the plain solver on average at every SNR
    omp = np.array([float(samples[i][:, 4].mean()) for i in range(11)])
    assert np.all(omp[:3] > 0.98) and omp[-1] < 0.6
    assert np.all(mean[:, 1].numpy() < mean[:, 0].numpy())


# ---- the second figure that holds outputs of the hot loop: results/errorVSadmmiters.fig (convergence_error curves) ----------
def admmiters_published():
    with open(os.path.join(HERE, "golden", "errorVSadmmiters_published.json")) as f:
        return json.load(f)


def admmiters_panel_points(pub):
    """(N_T, L_R, SNR) from the stored panel titles -> sweep points.  The figure comes from an OLDER revision of
    plot_errorVSadmmiters.m than the committed one (70 iterations, two curves; tests/golden/make_published_admmiters.py), so
    what is not in the titles follows the committed script: Nr = 32, L = 4, 'ps' combiner, frame T = 10 N_T used as it is
    (:21, :49), 20 realisations (:20)."""
    import re
    from jstsp19_amd.system_model import SweepParams
    pts = []
    for p in pub["panels"]:
        nt, lr, db = (int(v) for v in re.match(r"N_T=(\d+), L_R=(\d+), SNR=(\d+)db", p["title"]).groups())
        pts.append(SweepParams(Nt=nt, Nr=32, L=4, T=10 * nt, Mr=lr, snr_db=float(db), beamformer="ps", T_prop=10 * nt))
    return pts


def check_admmiters_shape(cur, pub, panels):
    """cur: (len(panels), 2, 70, 3) mean convergence_error of run_convergence_curves.  epsilon_1 = column 1
    (norm(V1)^2/norm(X)^2, proposed_algorithm.m:67), epsilon_2 = column 2 (:69).  The revision mismatch makes this a check of
    the decay SHAPE: every published point of the first 30 iterations within one decade of ours, at least six decades of
    decay of epsilon_1 over the 70 iterations (published: 6.3 to 10.2), epsilon_2 peaking within the first four iterations
    and ending on a plateau within a decade of the published one."""
    for k in panels:
        p = pub["panels"][k]
        e1, e2 = cur[k, 0, :, 0], cur[k, 0, :, 1]
        p1, p2 = np.array(p["epsilon_1"]), np.array(p["epsilon_2"])
        assert np.all(np.abs(np.log10(e1[:30] / p1[:30])) < 1.0), (p["title"], np.abs(np.log10(e1[:30] / p1[:30])).max())
        assert np.all(np.abs(np.log10(e2 / p2)) < 1.0), (p["title"], np.abs(np.log10(e2 / p2)).max())
        assert e1[-1] < 1e-6 * e1[0] and p1[-1] < 1e-6 * p1[0]
        assert np.all(np.diff(np.log10(e1[:40])) < 0)                        # monotone while far above the float floor
        assert np.argmax(e2) < 4 and np.argmax(p2) < 4
        assert abs(np.log10(e2[-1] / e2[-10])) < 0.05                        # plateau


def test_published_convergence_curves_have_our_decay_shape():
    """results/errorVSadmmiters.fig: the only other reference-held data about the hot loop - its third output.  A band, not
    a pin (unseeded, older revision): parity against the reference stays "unpinned"."""
    import torch
    from jstsp19_amd.montecarlo import run_convergence_curves
    pub = admmiters_published()
    pts = admmiters_panel_points(pub)
    cur = run_convergence_curves(pts, 20, Imax=pub["iterations"], batch=20).numpy()
    torch.cuda.synchronize()
    assert cur.shape == (4, 2, 70, 3)
    check_admmiters_shape(cur, pub, range(4))
    # the genie-support variant (the dashed curves of the committed script) also decays, though not as deep (float64
    # oracle: 2e-4 to 4e-4 of its first value after 70 iterations)
    assert np.all(cur[:, 1, -1, 0] < 1e-2 * cur[:, 1, 0, 0])
