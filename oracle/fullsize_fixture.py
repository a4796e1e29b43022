"""The float64 side of the full-size parity checks (TEST INFRASTRUCTURE: imported by tests/, by bench.py's parity leg and by
tools/ only - never by the product).

tests/golden/fullsize_port.npz holds, for 3008 trials at BASELINE configs[1]'s shape (N=64, M=4096, Gr=64, G2=512,
Imax=100), what oracle/cpu_port.cpp - the float64 C++ restatement of basic_system_functions/proposed_algorithm.m:1-73 and
proposed_algorithm_angles.m:1-85 - returns on the inputs the library's counter-based generator builds for the key (seed,
sweep index, trial index): per-trial NMSE (plot_errorVSsnr.m:138-141), convergence_error for a subset, the
hyper-parameters it was given and a fingerprint of the arrays.  5.8 core-seconds per trial: computed once on a GPU box's host
(tools/parity_tail.py), converted by tests/golden/make_fullsize_port_fixture.py.  `solve_group` recomputes the HIP side."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IMAX = 100


def fixture(name="fullsize_port"):
    """``fullsize_port``: the 3008 trials of round 4 (generator seed 20190913 - the set the round-4 defaults were chosen on);
    ``fullsize_port_heldout``: 2560 + 1280 trials of another seed (round 5), never used to choose a switch."""
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return {k: z[k] for k in z.files}


def solve_group(fx, group, rows, *, want_ce, angles, chunk=256):
    """HIP results for fixture rows `rows` of `group`: (nmse (n,), ce (n, Imax, 3) or None).  Rows are processed in runs of
    consecutive trials of one sweep point; every rebuilt trial must reproduce the fixture's fingerprint."""
    import torch
    import jstsp19_amd as J
    from jstsp19_amd.system_model import SweepParams, build_trials
    from oracle import solvers as O
    sidx, trial, snr, fp = (fx[group + "/" + k] for k in ("sweep_idx", "trial", "snr_db", "fingerprint"))
    rows = np.asarray(rows)
    nmse = np.empty(len(rows))
    ces = np.empty((len(rows), IMAX, 3)) if want_ce else None
    i = 0
    while i < len(rows):
        j = i + 1
        while (j < len(rows) and j - i < chunk and sidx[rows[j]] == sidx[rows[i]] and snr[rows[j]] == snr[rows[i]]
               and trial[rows[j]] == trial[rows[j - 1]] + 1):
            j += 1
        r = rows[i:j]
        p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=float(snr[r[0]]))
        seed = int(fx[group + "/seed"][r[0]]) if (group + "/seed") in fx else 20190913
        inp = build_trials(p, int(trial[r[0]]), len(r), seed=seed, sweep_idx=int(sidx[r[0]]))
        f = torch.stack([inp["subY"].abs().double().sum((1, 2)), inp["B"].abs().double().sum((1, 2)),
                         inp["Omega"].double().sum((1, 2))], 1).cpu().numpy()
        np.testing.assert_allclose(f, fp[r][:, :3], rtol=1e-9, err_msg="the generator no longer reproduces the fixture's inputs")
        # The hyper-parameters are INPUTS of the solver (plot_errorVSsnr.m:127-130): the float64 side was solved with the values
        # the fixture records, and so is the HIP side.  (The builder's own tau_Y, tau_Z, rho agree with them to fp32 rounding
        # only - rho = sigma_6 / ||Y||_F comes from a Gram whose split-K count depends on how many trials are built per call.)
        np.testing.assert_allclose(np.stack([inp[k].numpy() for k in ("tau_Y", "tau_Z", "rho")], 1), fp[r][:, 3:], rtol=2e-6)
        hyp = [np.ascontiguousarray(fp[r][:, 3 + k]) for k in range(3)]
        S, _, ce = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], IMAX, *hyp, "approximate",
                                        indx_S=inp["indx_S"] if angles else None, want_ce=want_ce)
        torch.cuda.synchronize()
        assert J.default_context(0).last_fused_fallbacks() == 0
        Sh = S.cpu().numpy().astype(np.complex128)
        zb = inp["Zbar"].cpu().numpy().astype(np.complex128)
        nmse[i:j] = [O.nmse_capped(Sh[t], zb[t]) for t in range(len(r))]
        if want_ce:
            ces[i:j] = ce.cpu().numpy()
        i = j
    return nmse, ces


