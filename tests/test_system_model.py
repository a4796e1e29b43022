"""The torch tensor-op input builder (tests/torch_builder.py: test infrastructure since round 4 - the product builds its
trials with the library's own kernels) vs the oracle's per-trial literal restatement of plot_errorVSsnr.m:57-136, on the
same random draws; and the sweep runners of jstsp19_amd.montecarlo on the CPU with that builder and the float64 oracle as
hooks.  Runs on CPU."""
import os
import sys

import numpy as np
import pytest
import torch

from jstsp19_amd.system_model import SweepParams, TrainingParams

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # (also when imported as tests.test_system_model by a spawned worker)
from torch_builder import build_inputs, build_inputs_training, builder as torch_builder, draw_trials, draw_trials_training
from oracle import system_model as osm


def _to_np(draws, t):
    return dict(gains=draws["gains"][t].numpy(), u_r=draws["u_r"][t].numpy(), u_t=draws["u_t"][t].numpy(),
                noise=draws["noise"][t].numpy(), qam_idx=draws["qam_idx"][t].numpy(),
                omega_rows=draws["omega_rows"][t].numpy())


def test_build_inputs_matches_oracle_per_trial():
    p = SweepParams(Nt=4, Nr=16, L=3, T=6, Mr=4, snr_db=5.0)
    draws = draw_trials(p, [0, 1, 7], seed=123, device="cpu")
    out = build_inputs(p, draws, out_dtype=torch.complex128)
    op = dict(Nt=p.Nt, Nr=p.Nr, Mr_e=p.Mr_e, Gr=p.Gr, Gt=p.Gt, clusters=p.clusters, rays=p.rays, L=p.L,
              Mr=p.Mr, T=p.T, noise_var=p.noise_var)
    for t in range(3):
        ref = osm.training_inputs_errorVSsnr(op, _to_np(draws, t))
        np.testing.assert_allclose(out["H"][t].numpy(), ref["H"], atol=1e-12)
        np.testing.assert_allclose(out["Zbar"][t].numpy(), ref["Zbar"], atol=1e-12)
        np.testing.assert_allclose(out["subY"][t].numpy(), ref["subY"], atol=1e-11)
        np.testing.assert_array_equal(out["Omega"][t].numpy(), ref["Omega"])
        np.testing.assert_allclose(out["A"].numpy(), ref["A"], atol=1e-12)
        np.testing.assert_allclose(out["B"][t].numpy(), ref["B"], atol=1e-12)
        np.testing.assert_allclose(float(out["tau_Y"][t]), ref["tau_Y"], rtol=1e-12)
        np.testing.assert_allclose(float(out["tau_Z"][t]), ref["tau_Z"], rtol=1e-12)
        np.testing.assert_allclose(float(out["rho"][t]), ref["rho"], rtol=1e-10)
        np.testing.assert_array_equal(out["indx_S"][t].numpy(), ref["indx_S"])
        assert int(out["Omega"][t].sum(dim=0).min()) == p.Mr == int(out["Omega"][t].sum(dim=0).max())


def test_draws_do_not_depend_on_batch_composition():
    """A trial's numbers are keyed by (seed, sweep, trial id): sharding trials over GPUs cannot
    change them."""
    p = SweepParams(Nt=2, Nr=8, L=2, T=4, Mr=2)
    a = draw_trials(p, [0, 1, 2, 3], seed=5, device="cpu")
    b = draw_trials(p, [2, 3], seed=5, device="cpu")
    for k in a:
        assert torch.equal(a[k][2:], b[k])
    c = draw_trials(p, [2, 3], seed=5, sweep_idx=1, device="cpu")
    assert not torch.equal(c["gains"], b["gains"])


def test_colmajor_layout_of_outputs():
    p = SweepParams(Nt=2, Nr=8, L=2, T=4, Mr=2)
    out = build_inputs(p, draw_trials(p, [0, 1], device="cpu"))
    N, M, Gr, G2 = p.solver_shape
    assert out["subY"].shape == (2, N, M) and out["subY"].stride() == (N * M, 1, N)
    assert out["B"].shape == (2, G2, M) and out["B"].stride() == (G2 * M, 1, G2)
    assert out["A"].shape == (N, Gr) and out["A"].stride() == (1, N)


def test_conventional_hbf_inputs_match_oracle():
    """hbf.m + plot_errorVSsnr.m:73-80 (the LS / VAMP baselines' measurement)."""
    p = SweepParams(Nt=4, Nr=16, L=3, T=12, Mr=4, snr_db=0.0)
    assert p.T_hbf == 3 * 4                                    # round(12/(16/4)) * Nt
    assert SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4).T_hbf == 16   # round(4.375) = 4 (plot_errorVSsnr.m:22)
    assert SweepParams(Nt=2, Nr=8, L=2, T=10, Mr=2).T_hbf == 3 * 2    # round(2.5) = 3: half AWAY from zero
    draws = draw_trials(p, [3, 4], seed=9, device="cpu")
    out = build_inputs(p, draws, out_dtype=torch.complex128, with_hbf=True)
    for t in range(2):
        d = _to_np(draws, t)
        H = out["H"][t].numpy()
        Psi_rows = np.stack([osm.toeplitz_rows(osm.qam4_alphabet()[d["qam_idx"][k]], p.L) for k in range(p.Nt)], axis=2)
        Nn = np.sqrt(p.noise_var / 2) * d["noise"]
        Th = p.T_hbf
        Yc, Wc, Psi_bar, _ = osm.hbf(H, Nn[:, :Th], Psi_rows[:, :Th, :], Th, p.Nr, osm.create_beamformer(p.Nr, "ZC"))
        Dr = np.exp(-2j * np.pi * np.outer(np.arange(p.Nr), np.arange(p.Gr)) / p.Gr) / np.sqrt(p.Nr)
        Dt = np.exp(-2j * np.pi * np.outer(np.arange(p.Nt), np.arange(p.Gt)) / p.Gt) / np.sqrt(p.Nt)
        np.testing.assert_allclose(out["Y_hbf"][t].numpy(), Yc, atol=1e-11)
        np.testing.assert_allclose(out["A_hbf"].numpy(), Wc.conj().T @ Dr, atol=1e-12)
        Bc = np.concatenate([Dt.conj().T @ Psi_bar[:, :, l] for l in range(p.L)])
        np.testing.assert_allclose(out["B_hbf"][t].numpy(), Bc, atol=1e-12)


def test_sweep_points_of_the_sibling_drivers():
    from jstsp19_amd.montecarlo import sweep_points
    base = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4)
    pts = sweep_points(base, "L", [1, 3, 5])
    assert [q.L for q in pts] == [1, 3, 5] and all(q.Nt == 4 and q.solver_shape[3] == q.L * 4 for q in pts)
    pts = sweep_points(base, "Nt", [2, 8])
    assert [q.Gt for q in pts] == [2, 8]
    assert [q.snr_db for q in sweep_points(base, "snr_db", [-15, 0, 15])] == [-15.0, 0.0, 15.0]


def _oracle_alg12(inp, Imax):
    """plot_errorVSsnr_approx.m:60-72 with the oracle as solver (the runner's hook)."""
    from oracle import solvers as O
    A = inp["A"].cpu().numpy().astype(complex)
    PA = np.linalg.pinv(A)
    out = ([], [])
    for t in range(inp["subY"].shape[0]):
        B = inp["B"][t].cpu().numpy().astype(complex)
        PB = np.linalg.pinv(B)
        zb = inp["Zbar"][t].cpu().numpy()
        for col, kind in enumerate(("std", "approximate")):
            _, Y, _ = O.proposed_algorithm(inp["subY"][t].cpu().numpy().astype(complex),
                                           inp["Omega"][t].cpu().numpy().astype(float), A, B, Imax, float(inp["tau_X"][t]),
                                           float(inp["tau_S"][t]), float(inp["rho"][t]), kind, want_ce=False)
            out[col].append(O.nmse_capped(PA @ Y @ PB, zb))
    return torch.tensor(out[0]), torch.tensor(out[1])


def test_training_builder_matches_oracle_per_trial():
    """wideband_hybBF_comm_system_training.m + plot_errorVSsnr_approx.m:45-58: batched torch builder vs the
    oracle's per-trial restatement on the same draws, at the driver's own sizes."""
    p = TrainingParams(Nt=4, Nr=32, L=4, T=70, ratio=0.75, snr_db=-5.0)
    assert p.Lr == 24 and p.solver_shape == (32, 70, 32, 16)
    draws = draw_trials_training(p, [0, 3], seed=5, device="cpu")
    out = build_inputs_training(p, draws, out_dtype=torch.complex128)
    op = dict(Nt=p.Nt, Nr=p.Nr, L=p.L, T=p.T, clusters=p.clusters, rays=p.rays, ratio=p.ratio, noise_var=p.noise_var)
    for t in range(2):
        ref = osm.training_inputs_errorVSsnr_approx(op, {k: v[t].numpy() for k, v in draws.items()})
        assert ref["Lr"] == 24
        np.testing.assert_allclose(out["H"][t].numpy(), ref["H"], atol=1e-12)
        np.testing.assert_allclose(out["Zbar"][t].numpy(), ref["Zbar"], atol=1e-11)
        np.testing.assert_allclose(out["subY"][t].numpy(), ref["subY"], atol=1e-11)
        np.testing.assert_array_equal(out["Omega"][t].numpy(), ref["Omega"])
        np.testing.assert_allclose(out["A"].numpy(), ref["A"], atol=1e-12)
        np.testing.assert_allclose(out["B"][t].numpy(), ref["B"], atol=1e-12)
        np.testing.assert_allclose(float(out["tau_X"][t]), ref["tau_X"], rtol=1e-12)
        np.testing.assert_allclose(float(out["tau_S"][t]), ref["tau_X"] / 2, rtol=1e-12)
        np.testing.assert_allclose(float(out["rho"][t]), ref["rho"], rtol=1e-10)
        assert np.all(ref["Omega"].sum(axis=0) == 24)
    # the combiner is unitary and A = W' Dr with Dr the same DFT: A is the identity (Gr = Nr)
    np.testing.assert_allclose(out["A"].numpy(), np.eye(32), atol=1e-12)
    assert TrainingParams(Nr=10, ratio=0.25).Lr == 3          # MATLAB round(2.5) = 3, not numpy's 2


def test_training_builder_is_the_reference_alternative_formulation():
    """The commented check of wideband_hybBF_comm_system_training.m:35-44: R - N equals
    sum_k [H(:,k,1) ... H(:,k,L)] * Psi_i(1:L,:,k)."""
    rng = np.random.default_rng(3)
    p = dict(Nt=3, Nr=8, L=3, T=12, clusters=2, rays=2, ratio=0.5, noise_var=0.0)
    d = osm.draw_trial_approx(rng, p)
    inp = osm.training_inputs_errorVSsnr_approx(p, d)
    H = inp["H"]
    Y = np.zeros((8, 12), complex)
    for k in range(3):
        Y += H[:, k, :] @ osm.toeplitz_rows(d["pilots"][k] / np.sqrt(2), 3)
    n = np.arange(8)
    W = np.exp(-2j * np.pi * np.outer(n, n) / 8) / np.sqrt(8)
    np.testing.assert_allclose(inp["subY"], inp["Omega"] * (W.conj().T @ Y), atol=1e-12)


def test_alg1_vs_alg2_sweep_runner_with_oracle_solver():
    """run_approx_sweep (plot_errorVSsnr_approx.m:34-85) on CPU with the oracle as the solver hook: shape, caps,
    the NMSE falls with the SNR, and both variants end close to each other (the figure's point)."""
    from jstsp19_amd.montecarlo import run_approx_sweep
    base = TrainingParams(Nt=2, Nr=8, L=2, T=20, ratio=0.75)
    out = run_approx_sweep(base, [-10.0, 10.0], [5, 20], 3, batch=2, device=torch.device("cpu"),
                           solve_fn=_oracle_alg12, builder=torch_builder).numpy()
    assert out.shape == (2, 2, 2) and np.all(out > 0) and np.all(out <= 1)
    assert np.all(out[:, 1, :] < out[:, 0, :])
    assert np.all(np.abs(out[1, :, 0] - out[1, :, 1]) < 0.5 * out[1, :, 0] + 1e-3)


def _oracle_params(p):
    return dict(Nt=p.Nt, Nr=p.Nr, Mr_e=p.Mr_e, Gr=p.Gr, Gt=p.Gt, clusters=p.clusters, rays=p.rays, L=p.L, Mr=p.Mr,
                T=p.T, noise_var=p.noise_var, beamformer=p.beamformer, rho_rule=p.rho_rule, rho_scale=p.rho_scale,
                T_prop=p.T_prop)


@pytest.mark.parametrize("kw", [dict(beamformer="fft"),                                  # plot_errorVSframelength.m:123
                                dict(rho_rule="max"),                                    # plot_errorVSdelays.m:128
                                dict(beamformer="ps", rho_scale=0.5, T_prop=12)])        # plot_errorVSzy.m:53,65,30
def test_sibling_driver_construction_variants_match_oracle(kw):
    p = SweepParams(Nt=4, Nr=16, L=3, T=6, Mr=4, snr_db=5.0, **kw)
    assert p.T_prop == (12 if "T_prop" in kw else 24)
    draws = draw_trials(p, [2], seed=11, device="cpu")
    out = build_inputs(p, draws, out_dtype=torch.complex128, with_hbf=True)
    ref = osm.training_inputs_errorVSsnr(_oracle_params(p), _to_np(draws, 0))
    np.testing.assert_allclose(out["subY"][0].numpy(), ref["subY"], atol=1e-11)
    np.testing.assert_allclose(out["A"].numpy(), ref["A"], atol=1e-12)
    np.testing.assert_allclose(out["B"][0].numpy(), ref["B"], atol=1e-12)
    np.testing.assert_allclose(float(out["rho"][0]), ref["rho"], rtol=1e-10)
    if kw.get("beamformer") in ("fft", "ps"):                  # unitary DFT combiner x DFT dictionary
        np.testing.assert_allclose(out["A"].numpy(), np.eye(16), atol=1e-12)
        np.testing.assert_allclose(osm.create_beamformer(16, "fft"), osm.create_beamformer(16, "ps"), atol=1e-14)
    if kw.get("rho_rule") == "max":
        s = np.linalg.svd(ref["subY"], compute_uv=False)
        np.testing.assert_allclose(ref["rho"], s[0] / np.linalg.norm(ref["subY"], "fro"), rtol=1e-12)


def test_driver_presets_follow_the_reference_scripts():
    from jstsp19_amd.montecarlo import admmiters_points, driver, zy_points
    d = driver("errorVSdelays")                                 # plot_errorVSdelays.m:16,43-46: (L, T) move together
    assert [(p.L, p.T, p.T_prop, p.T_hbf) for p in d["points"]] == [(2, 5, 20, 4), (4, 10, 40, 4), (6, 15, 60, 8),
                                                                    (8, 20, 80, 12), (10, 25, 100, 12)]
    assert all(p.rho_rule == "max" and p.beamformer == "ZC" and p.snr_db == 5.0 for p in d["points"])
    d = driver("errorVSnt")                                     # plot_errorVSnt.m:7,22,44-48
    assert [(p.Nt, p.Gt, p.T, p.T_prop) for p in d["points"]] == [(4, 4, 35, 140), (6, 6, 35, 210), (8, 8, 35, 280),
                                                                  (12, 12, 35, 420), (16, 16, 25, 400)]
    assert all(p.beamformer == "fft" and p.rho_rule == "max" for p in d["points"])
    d = driver("errorVSnrf")                                    # :20,45 — round(5/(32/16)) = round(2.5) = 3 in MATLAB
    assert [(p.Mr, p.T_hbf) for p in d["points"]] == [(4, 4), (8, 4), (12, 8), (16, 12)]
    assert driver("rateVSframelength")["metric"] == "rate" and driver("errorVSsnr")["values"] == list(range(-15, 16, 3))
    assert [p.solver_shape for p in admmiters_points()] == [(32, 40, 32, 16), (32, 160, 32, 64), (32, 160, 32, 64),
                                                            (32, 480, 32, 64)]
    z = zy_points()[0]
    assert z.solver_shape == (32, 80, 32, 64) and z.rho_scale == 0.5 and z.rays == 6
    with pytest.raises(ValueError):
        driver("nope")


def _oracle_curves(inp, Imax):
    from oracle import solvers as O
    A = inp["A"].cpu().numpy().astype(complex)
    ce, cea = [], []
    for t in range(inp["subY"].shape[0]):
        args = (inp["subY"][t].cpu().numpy().astype(complex), inp["Omega"][t].cpu().numpy().astype(float), A,
                inp["B"][t].cpu().numpy().astype(complex), Imax, float(inp["tau_Y"][t]), float(inp["tau_Z"][t]),
                float(inp["rho"][t]), "approximate")
        ce.append(O.proposed_algorithm(*args)[2])
        cea.append(O.proposed_algorithm(*args, indx_S=inp["indx_S"][t].cpu().numpy())[2])
    return torch.tensor(np.stack(ce)), torch.tensor(np.stack(cea))


def _oracle_zy(inp, Imax):
    from oracle import solvers as O
    A = inp["A"].cpu().numpy().astype(complex)
    ez, ey = [], []
    for t in range(inp["subY"].shape[0]):
        B = inp["B"][t].cpu().numpy().astype(complex)
        S, Y, _ = O.proposed_algorithm(inp["subY"][t].cpu().numpy().astype(complex),
                                       inp["Omega"][t].cpu().numpy().astype(float), A, B, Imax, float(inp["tau_Y"][t]),
                                       float(inp["tau_Z"][t]), float(inp["rho"][t]), "approximate", want_ce=False)
        zb = inp["Zbar"][t].cpu().numpy()
        ez.append(O.nmse_capped(S, zb))
        ey.append(O.nmse_capped(A.conj().T @ Y @ np.linalg.pinv(B), zb))
    return torch.tensor(ez), torch.tensor(ey)


def test_convergence_curve_and_zy_runners_with_oracle_solver():
    """plot_errorVSadmmiters.m:32-71 and plot_errorVSzy.m:28-84 on CPU with the oracle as the solver hook."""
    from jstsp19_amd.montecarlo import run_convergence_curves, run_zy
    pts = [SweepParams(Nt=2, Nr=8, L=2, T=12, Mr=4, snr_db=15.0, beamformer="ps", T_prop=12)]
    cur = run_convergence_curves(pts, 3, Imax=12, batch=2, device=torch.device("cpu"), solve_fn=_oracle_curves, builder=torch_builder).numpy()
    assert cur.shape == (1, 2, 12, 3) and np.all(cur >= 0)
    assert np.all(np.isinf(cur[:, :, 0, 2])) and np.all(np.isfinite(cur[:, :, 1:, :]))   # C = 0 before iteration 1: x/0
    assert cur[0, 0, -1, 0] < cur[0, 0, 0, 0]                    # epsilon_1 falls over the iterations
    one = run_convergence_curves(pts, 1, Imax=12, batch=1, device=torch.device("cpu"), solve_fn=_oracle_curves, builder=torch_builder).numpy()
    inp = build_inputs(pts[0], draw_trials(pts[0], [0], device="cpu"))
    np.testing.assert_allclose(one[0, 0], _oracle_curves(inp, 12)[0][0].numpy(), rtol=1e-12)
    zy = run_zy([pts[0].replace(rho_scale=0.5)], 3, Imax=10, batch=2, device=torch.device("cpu"), solve_fn=_oracle_zy, builder=torch_builder).numpy()
    assert zy.shape == (1, 2) and np.all(zy > 0) and np.all(zy <= 1)


def test_published_admmiters_panel_has_the_oracles_decay_shape():
    """results/errorVSadmmiters.fig, first panel ('N_T=4, L_R=24, SNR=5db'), against the float64 oracle as the solver hook
    (CPU tier of tests/test_gpu_published_curves.py::test_published_convergence_curves_have_our_decay_shape; 8 realisations)."""
    from test_gpu_published_curves import admmiters_panel_points, admmiters_published, check_admmiters_shape
    from jstsp19_amd.montecarlo import run_convergence_curves
    pub = admmiters_published()
    pts = admmiters_panel_points(pub)
    assert (pts[1].Nt, pts[1].Mr, pts[2].snr_db, pts[3].Mr) == (8, 24, 15.0, 19)
    cur = run_convergence_curves(pts[:1], 8, Imax=70, batch=8, device=torch.device("cpu"), solve_fn=_oracle_curves, builder=torch_builder).numpy()
    check_admmiters_shape(cur, pub, [0])
