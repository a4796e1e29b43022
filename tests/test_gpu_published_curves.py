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
