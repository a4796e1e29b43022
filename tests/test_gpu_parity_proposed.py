"""HIP path vs the float64 oracle / golden fixtures — proposed_algorithm(_angles).

Tolerances (fp32 device arithmetic vs float64 reference restatement):
  S, Y   : max|d| / max|ref| <= 1e-5   (conftest.TOL_S: 5x the largest error measured, round 6)
  NMSE   : |d| <= 1e-6 (BASELINE.json north_star), every fixture
  ce     : relative 5e-4 on every finite entry of convergence_error (conftest.TOL_CE)
"""
import numpy as np
import pytest

from conftest import load_golden, rel_err, check_below, ce_rel, TOL_S, TOL_CE, TOL_NMSE

pytestmark = pytest.mark.gpu



def _check(out, g, key, nmse_tol=1e-6, check_ce=True):
    import jstsp19_amd as J
    from oracle import solvers as O
    S, Y, ce = out
    check_below("golden.S", rel_err(S, g["S_" + key]), TOL_S)
    check_below("golden.Y", rel_err(Y, g["Y_" + key]), TOL_S)
    nm = O.nmse_capped(np.asarray(S, dtype=np.complex128), g["Zbar"])
    check_below("golden.nmse", abs(nm - float(g["nmse_" + key])), nmse_tol)
    if check_ce:
        ref = g["ce_" + key]
        assert ce.shape == ref.shape
        assert np.isinf(ce[0, 2]) and np.isinf(ref[0, 2])          # 0-divide at i = 1 (proposed_algorithm.m:51)
        check_below("golden.ce3", ce_rel(ce[1:, 2], ref[1:, 2]), TOL_CE)
        check_below("golden.ce12", ce_rel(ce[:, :2], ref[:, :2]), TOL_CE)


@pytest.mark.parametrize("name", ["proposed_small", "proposed_small_lowsnr", "proposed_refnative"])
def test_proposed_host_path_matches_golden(name):
    import jstsp19_amd as J
    g = load_golden(name)
    out = J.proposed_algorithm(g["subY"], g["Omega"], g["A"], g["B"], int(g["Imax"]), float(g["tau_Y"]),
                               float(g["tau_Z"]), float(g["rho"]), "approximate")
    _check(out, g, "approximate", nmse_tol=TOL_NMSE)


@pytest.mark.parametrize("name", ["proposed_small", "proposed_refnative"])
def test_proposed_angles_matches_golden(name):
    import jstsp19_amd as J
    g = load_golden(name)
    out = J.proposed_algorithm_angles(g["subY"], g["Omega"], g["indx_S"], g["A"], g["B"], int(g["Imax"]),
                                      float(g["tau_Y"]), float(g["tau_Z"]), float(g["rho"]), "approximate", 100)
    _check(out, g, "angles", nmse_tol=TOL_NMSE)


def test_first_iteration_invariants():
    """Y_1 = 0 (svt of the zero matrix, svt.m:8-12) and ce(1,3) = Inf (proposed_algorithm.m:51)."""
    import jstsp19_amd as J
    g = load_golden("proposed_small")
    S, Y, ce = J.proposed_algorithm(g["subY"], g["Omega"], g["A"], g["B"], 1, float(g["tau_Y"]),
                                    float(g["tau_Z"]), float(g["rho"]), "approximate")
    assert np.all(Y == 0)
    assert np.isinf(ce[0, 2])
    assert np.all(np.isfinite(S))


def test_batched_device_path_per_trial_and_shared_dictionaries():
    """batch > 1 on device-resident tensors: per-trial B, and one A shared by the batch.
    Each problem must equal its own un-batched solve (same kernels => tight tolerance) and
    the oracle."""
    import torch
    import jstsp19_amd as J
    from oracle import solvers as O
    from oracle import system_model as sm
    from oracle.make_golden import PARAMS_REF
    dev = torch.device("cuda:0")
    trials = []
    for t in range(5):
        p = dict(PARAMS_REF)
        p["noise_var"] = 10 ** (-(3.0 * t - 5) / 10)
        rng = np.random.default_rng(100 + t)
        trials.append(sm.training_inputs_errorVSsnr(p, sm.draw_trial(rng, p)))
    stack = lambda k, dt: np.stack([tr[k] for tr in trials]).astype(dt)
    subY, Om, B = stack("subY", np.complex64), stack("Omega", np.float32), stack("B", np.complex64)
    A = trials[0]["A"].astype(np.complex64)                 # ZC combiner x DFT: identical for every trial
    tY = np.array([tr["tau_Y"] for tr in trials]); tZ = np.array([tr["tau_Z"] for tr in trials])
    rho = np.array([tr["rho"] for tr in trials])
    cm = lambda a: J.colmajor(torch.from_numpy(a).to(dev))
    S, Y, ce = J.proposed_algorithm(cm(subY), cm(Om), cm(A), cm(B), 100, tY, tZ, rho, "approximate")
    torch.cuda.synchronize()
    S = S.cpu().numpy(); Y = Y.cpu().numpy(); ce = ce.cpu().numpy()
    assert S.shape == (5, 32, 16) and Y.shape == (5, 32, 140) and ce.shape == (5, 100, 3)
    for t, tr in enumerate(trials):
        So, Yo, ceo = O.proposed_algorithm(tr["subY"], tr["Omega"], tr["A"], tr["B"], 100, tr["tau_Y"],
                                           tr["tau_Z"], tr["rho"], "approximate")
        check_below("batched.S", rel_err(S[t], So), TOL_S)
        check_below("batched.Y", rel_err(Y[t], Yo), TOL_S)
        check_below("batched.nmse", abs(O.nmse_capped(S[t].astype(complex), tr["Zbar"]) - O.nmse_capped(So, tr["Zbar"])), TOL_NMSE)
        check_below("batched.ce", ce_rel(ce[t], ceo), TOL_CE)
        S1, Y1, _ = J.proposed_algorithm(tr["subY"], tr["Omega"], tr["A"], tr["B"], 100, tr["tau_Y"],
                                         tr["tau_Z"], tr["rho"], "approximate", want_ce=False)
        assert rel_err(S[t], S1) < 1e-5


def test_ragged_shapes_not_multiples_of_the_tile():
    """N, M, Gr, G2 that are not multiples of 32/64/16 (edge tiles, odd N*M => scalar path)."""
    import jstsp19_amd as J
    from oracle import solvers as O
    rng = np.random.default_rng(7)
    N, M, Gr, G2 = 7, 13, 5, 9
    A = (rng.standard_normal((N, Gr)) + 1j * rng.standard_normal((N, Gr))) / np.sqrt(N)
    B = (rng.standard_normal((G2, M)) + 1j * rng.standard_normal((G2, M))) / np.sqrt(G2)
    S0 = np.zeros((Gr, G2), complex); S0[1, 2] = 3 + 1j; S0[4, 7] = -2j
    Om = (rng.random((N, M)) < 0.5).astype(float)
    subY = Om * (A @ S0 @ B + 0.05 * (rng.standard_normal((N, M)) + 1j * rng.standard_normal((N, M))))
    args = (subY, Om, A, B, 25, 0.01, 0.02, 0.3, "approximate")
    So, Yo, ceo = O.proposed_algorithm(*args)
    S, Y, ce = J.proposed_algorithm(*args)
    check_below("ragged.S", rel_err(S, So), TOL_S); check_below("ragged.Y", rel_err(Y, Yo), TOL_S)
    check_below("ragged.ce", ce_rel(ce, ceo), TOL_CE)


def test_bad_arguments_are_rejected_not_crashed():
    import jstsp19_amd as J
    g = load_golden("proposed_small")
    with pytest.raises(ValueError):
        J.proposed_algorithm(g["subY"], g["Omega"][:, :-1], g["A"], g["B"], 5, 1.0, 1.0, 1.0)
    with pytest.raises(ValueError):
        J.proposed_algorithm(g["subY"], g["Omega"], g["A"][:-1], g["B"], 5, 1.0, 1.0, 1.0)


def test_proposed_std_type_matches_golden_and_oracle():
    """type ~= 'approximate' ('std'): v = U\\(L\\k), the LU least squares of proposed_algorithm.m:29,53."""
    import jstsp19_amd as J
    from oracle import solvers as O
    g = load_golden("proposed_small")
    S, Y, ce = J.proposed_algorithm(g["subY"], g["Omega"], g["A"], g["B"], int(g["Imax"]), float(g["tau_Y"]),
                                    float(g["tau_Z"]), float(g["rho"]), "std")
    assert rel_err(S, g["S_std"]) < 5e-4 and rel_err(Y, g["Y_std"]) < 5e-4
    assert np.all(ce[:, 2] == 0)                         # column 3 is only written by 'approximate' (:51)
    np.testing.assert_allclose(ce[:, :2], g["ce_std"][:, :2], rtol=5e-3)
    # a larger, well-conditioned over-determined system incl. the Newton-Schulz inverse (G2 > 128)
    rng = np.random.default_rng(77)
    N, M, Gr, G2 = 24, 200, 16, 140
    A = (rng.standard_normal((N, Gr)) + 1j * rng.standard_normal((N, Gr))) / np.sqrt(N)
    B = (rng.standard_normal((G2, M)) + 1j * rng.standard_normal((G2, M))) / np.sqrt(M)
    S0 = np.zeros((Gr, G2), complex)
    S0[rng.integers(0, Gr, 10), rng.integers(0, G2, 10)] = rng.standard_normal(10) + 1j * rng.standard_normal(10)
    Om = (rng.random((N, M)) < 0.6).astype(float)
    subY = Om * (A @ S0 @ B + 0.02 * (rng.standard_normal((N, M)) + 1j * rng.standard_normal((N, M))))
    args = (subY, Om, A, B, 20, 0.01, 0.02, 0.3, "std")
    So, Yo, _ = O.proposed_algorithm(*args)
    Sg, Yg, _ = J.proposed_algorithm(*args)
    assert rel_err(Sg, So) < 1e-3 and rel_err(Yg, Yo) < 1e-3
    with pytest.raises(J.JstspError):                    # under-determined K2 is refused, not approximated
        J.proposed_algorithm(subY[:, :100], Om[:, :100], A, B[:, :100], 5, 0.01, 0.02, 0.3, "std")


def test_sweep_runner_on_hip_matches_oracle_per_point():
    """montecarlo.run_sweep (the plot_errorVSsnr.m:48-180 counterpart) with the HIP solvers vs the same
    sweep with the oracle as solver: identical trials (RNG keyed by global ids) => identical means to 1e-6."""
    import torch
    from jstsp19_amd.montecarlo import run_sweep
    from jstsp19_amd.system_model import SweepParams
    from oracle import solvers as O

    def oracle_solve(inp, Imax):
        e, ea = [], []
        A = inp["A"].cpu().numpy().astype(complex)
        for t in range(inp["subY"].shape[0]):
            args = (inp["subY"][t].cpu().numpy().astype(complex), inp["Omega"][t].cpu().numpy().astype(float), A,
                    inp["B"][t].cpu().numpy().astype(complex), Imax, float(inp["tau_Y"][t]), float(inp["tau_Z"][t]),
                    float(inp["rho"][t]), "approximate")
            S, _, _ = O.proposed_algorithm(*args, want_ce=False)
            Sa, _, _ = O.proposed_algorithm(*args, indx_S=inp["indx_S"][t].cpu().numpy(), want_ce=False)
            zb = inp["Zbar"][t].cpu().numpy()
            e.append(O.nmse_capped(S, zb)); ea.append(O.nmse_capped(Sa, zb))
        return torch.tensor(e), torch.tensor(ea)

    base = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4)
    dev = torch.device("cuda:0")
    hip = run_sweep(base, [-9, 3, 15], 6, Imax=100, batch=4, device=dev).numpy()
    ref = run_sweep(base, [-9, 3, 15], 6, Imax=100, batch=4, device=dev, solve_fn=oracle_solve, builder="hip").numpy()
    np.testing.assert_allclose(hip, ref, atol=1e-6)
    # (the HIP sweep above solved the trials of all three points in one call; point by point, batch by batch is the same)
    np.testing.assert_allclose(run_sweep(base, [-9, 3, 15], 6, Imax=100, batch=4, device=dev, merge=False).numpy(), hip, atol=1e-6)
    assert hip.shape == (3, 2) and np.all(hip[:, 1] <= hip[:, 0] + 1e-3)      # angle information helps
    # with the conventional-HBF baselines (LS, VAMP, MMV-OMP) as extra columns, and the TSSR recipe
    full = run_sweep(base, [3], 4, Imax=100, batch=4, device=dev, baselines=True, tssr=(30, 0.1)).numpy()
    assert full.shape == (1, 7) and np.all(np.isfinite(full)) and np.all(full > 0) and np.all(full <= 1)
    np.testing.assert_allclose(full[0, :2], run_sweep(base, [3], 4, Imax=100, batch=4, device=dev).numpy()[0], atol=1e-7)
    # numOfnz = 100 >= the 32 atoms of the square A: joint OMP ends at the LS estimate (the two curves of
    # results/errorVSsnr_angles.fig coincide)
    assert abs(full[0, 4] - full[0, 2]) < 1e-3
    # the rate metric of plot_rateVSframelength.m:81 on the same trials: positive, angle information does not hurt
    rt = run_sweep(base, [3], 4, Imax=100, batch=4, device=dev, metric="rate").numpy()
    assert rt.shape == (1, 2) and np.all(rt > 0) and rt[0, 1] >= rt[0, 0] - 1e-3


def test_alg1_vs_alg2_sweep_on_hip_matches_oracle_per_point():
    """montecarlo.run_approx_sweep (plot_errorVSsnr_approx.m:34-85: wideband_hybBF_comm_system_training inputs,
    'std' and 'approximate', S = pinv(A)*Y*pinv(B)) with the HIP solvers vs the oracle on identical trials,
    at the driver's own sizes (Nt 4, Nr 32, L 4, T 70, ratio 0.75)."""
    import torch
    from jstsp19_amd.montecarlo import run_approx_sweep
    from jstsp19_amd.system_model import TrainingParams
    from tests.test_system_model import _oracle_alg12
    base = TrainingParams()
    dev = torch.device("cuda:0")
    from torch_builder import builder as torch_builder
    hip = run_approx_sweep(base, [-15, 0, 15], [10, 50], 4, batch=4, device=dev, builder=torch_builder).numpy()
    ref = run_approx_sweep(base, [-15, 0, 15], [10, 50], 4, batch=4, device=dev, solve_fn=_oracle_alg12, builder=torch_builder).numpy()
    assert hip.shape == (2, 3, 2) and np.all(hip > 0) and np.all(hip <= 1)
    np.testing.assert_allclose(hip, ref, rtol=2e-4, atol=1e-6)
    assert np.all(hip[:, 2, :] < hip[:, 0, :])                              # the NMSE falls with the SNR


@pytest.mark.parametrize("name", ["errorVSdelays", "errorVSnt", "errorVSnrf", "rateVSframelength"])
def test_sibling_driver_presets_on_hip_match_oracle(name):
    """montecarlo.run_driver: the sibling drivers at their own parameters (beamformer kind, min/max eigenvalue in rho,
    joint (L,T) / (Nt,T) axes) — proposed / angles columns against the oracle on the same HIP-built trials."""
    import torch
    from jstsp19_amd.montecarlo import driver, run_points
    from oracle import solvers as O
    d = driver(name)
    pts = d["points"][:2] + d["points"][-1:]
    rate = d["metric"] == "rate"

    def oracle_solve(inp, Imax):
        e, ea = [], []
        A = inp["A"].cpu().numpy().astype(complex)
        for t in range(inp["subY"].shape[0]):
            args = (inp["subY"][t].cpu().numpy().astype(complex), inp["Omega"][t].cpu().numpy().astype(float), A,
                    inp["B"][t].cpu().numpy().astype(complex), Imax, float(inp["tau_Y"][t]), float(inp["tau_Z"][t]),
                    float(inp["rho"][t]), "approximate")
            S, _, _ = O.proposed_algorithm(*args, want_ce=False)
            Sa, _, _ = O.proposed_algorithm(*args, indx_S=inp["indx_S"][t].cpu().numpy(), want_ce=False)
            zb = inp["Zbar"][t].cpu().numpy()
            e.append(O.nmse_capped(S, zb)); ea.append(O.nmse_capped(Sa, zb))
        return torch.tensor(e), torch.tensor(ea)

    dev = torch.device("cuda:0")
    hip = run_points(pts, 2, Imax=d["Imax"], numOfnz=d["numOfnz"], metric=d["metric"], baselines=True, batch=2,
                     device=dev).numpy()
    assert hip.shape == (3, 5) and np.all(np.isfinite(hip)) and np.all(hip > 0)
    if not rate:
        ref = run_points(pts, 2, Imax=d["Imax"], batch=2, device=dev, solve_fn=oracle_solve, builder="hip").numpy()
        np.testing.assert_allclose(hip[:, :2], ref, rtol=1e-3, atol=2e-6)
        assert np.all(hip <= 1)


def test_convergence_curves_and_zy_on_hip_match_oracle():
    """run_convergence_curves (plot_errorVSadmmiters.m:32-71, the first panel's parameters) and run_zy
    (plot_errorVSzy.m:28-84 at its own size) against the oracle on the same HIP-built trials."""
    import torch
    from jstsp19_amd.montecarlo import admmiters_points, run_convergence_curves, run_zy, zy_points
    from tests.test_system_model import _oracle_curves, _oracle_zy
    dev = torch.device("cuda:0")
    pts = admmiters_points()[:1]
    hip = run_convergence_curves(pts, 3, Imax=40, batch=3, device=dev).numpy()
    ref = run_convergence_curves(pts, 3, Imax=40, batch=3, device=dev, solve_fn=_oracle_curves, builder="hip").numpy()
    assert hip.shape == (1, 2, 40, 3)
    assert np.all(np.isinf(hip[:, :, 0, 2])) and np.all(np.isinf(ref[:, :, 0, 2]))      # C = 0 before iteration 1
    np.testing.assert_allclose(hip[:, :, 1:, :], ref[:, :, 1:, :], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(hip[:, :, 0, :2], ref[:, :, 0, :2], rtol=2e-3, atol=1e-7)
    big = run_convergence_curves(admmiters_points()[3:], 2, Imax=100, batch=2, device=dev).numpy()   # 32 x 480, G2 = 64
    assert big.shape == (1, 2, 100, 3) and np.all(np.isfinite(big[:, :, 1:, :])) and big[0, 0, -1, 0] < big[0, 0, 0, 0]
    zy = run_zy(zy_points(), 2, batch=2, device=dev).numpy()
    zr = run_zy(zy_points(), 2, batch=2, device=dev, solve_fn=_oracle_zy, builder="hip").numpy()
    assert zy.shape == (1, 2)
    np.testing.assert_allclose(zy, zr, rtol=1e-3, atol=2e-6)


def test_lanczos_lambda_max_agrees_with_householder_sturm():
    """convergence_error(:,1:2) takes lambda_max from the one-wave Lanczos kernel (JSTSP_LANCZOS=0 switches back to the
    Householder + Sturm kernel): both must give the same ratios far inside the 2e-3 parity tolerance — Gram orders
    32 (reference-native), 12, 64 and 100 (two rows per lane)."""
    import os
    import jstsp19_amd as J
    rng = np.random.default_rng(41)
    r = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    g = load_golden("proposed_refnative")
    cases = [(g["subY"], g["Omega"], g["A"], g["B"], 30, float(g["tau_Y"]), float(g["tau_Z"]), float(g["rho"]), "approximate")]
    for (N, M, Gr, G2) in [(12, 40, 6, 10), (64, 300, 32, 40), (100, 160, 40, 30)]:
        A, B = r(N, Gr) / np.sqrt(N), r(G2, M) / np.sqrt(G2)
        Om = (rng.random((3, N, M)) < 0.5).astype(float)
        cases.append((Om * r(3, N, M), Om, A, B, 15, 0.01, 0.02, 0.3, "approximate"))
    old = os.environ.get("JSTSP_LANCZOS")
    try:
        for args in cases:
            os.environ["JSTSP_LANCZOS"] = "0"
            S0, Y0, ce0 = J.proposed_algorithm(*args)
            os.environ["JSTSP_LANCZOS"] = "1"
            S1, Y1, ce1 = J.proposed_algorithm(*args)
            assert np.array_equal(S0, S1) and np.array_equal(Y0, Y1)          # the norms do not feed back
            np.testing.assert_allclose(ce1[..., :2], ce0[..., :2], rtol=2e-5)
    finally:
        if old is None:
            os.environ.pop("JSTSP_LANCZOS", None)
        else:
            os.environ["JSTSP_LANCZOS"] = old
