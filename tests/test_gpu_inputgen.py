"""HIP input construction (jstsp_build_trials_c32, csrc/inputgen.hip) vs the oracle's per-trial
restatement of plot_errorVSsnr.m:57-136 on the SAME random draws (the library hands its raw Philox
draws back; the oracle rebuilds every array from them in float64), plus the statistical and
sharding properties of the generator."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _oracle_inputs(p, out, t):
    from oracle import system_model as osm
    Np = p.clusters * p.rays
    u_r = np.zeros((p.L, Np)); u_r[0] = out["u_r"][t].cpu().numpy()
    u_t = np.zeros((p.L, Np)); u_t[0] = out["u_t"][t].cpu().numpy()
    om = out["Omega"][t].cpu().numpy()
    rows = np.stack([np.flatnonzero(om[:, j]) for j in range(om.shape[1])])
    d = dict(gains=out["gains"][t].cpu().numpy().astype(complex), u_r=u_r, u_t=u_t,
             noise=out["noise"][t].cpu().numpy().astype(complex), qam_idx=out["qam_idx"][t].cpu().numpy().astype(int),
             omega_rows=rows)
    op = dict(Nt=p.Nt, Nr=p.Nr, Mr_e=p.Mr_e, Gr=p.Gr, Gt=p.Gt, clusters=p.clusters, rays=p.rays, L=p.L, Mr=p.Mr,
              T=p.T, noise_var=p.noise_var, beamformer=p.beamformer, rho_rule=p.rho_rule, rho_scale=p.rho_scale,
              T_prop=p.T_prop)
    return osm.training_inputs_errorVSsnr(op, d), d


@pytest.mark.parametrize("kw", [dict(Nt=4, Nr=16, L=3, T=6, Mr=4, snr_db=5.0),
                                dict(Nt=4, Nr=32, L=4, T=20, Mr=4, snr_db=-5.0),          # plot_errorVSsnr.m shape
                                dict(Nt=2, Nr=12, L=2, T=7, Mr=3, Mr_e=9, Gr=16, Gt=4, clusters=3, rays=2, snr_db=10.0),
                                # what the sibling drivers change: 'fft' combiner + max(eigs) (plot_errorVSnt.m:123,129),
                                # 'ps' combiner, the frame itself as T_prop, rho halved (plot_errorVSzy.m:30,53,65)
                                dict(Nt=6, Nr=32, L=4, T=10, Mr=4, snr_db=15.0, beamformer="fft", rho_rule="max"),
                                dict(Nt=16, Nr=32, L=4, T=80, Mr=16, rays=6, snr_db=15.0, beamformer="ps", rho_scale=0.5,
                                     T_prop=80)])
def test_build_trials_matches_oracle_on_its_own_draws(kw):
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(**kw)
    out = build_trials(p, 5, 3, seed=77, sweep_idx=2, want_draws=True, want_H=True)
    torch.cuda.synchronize()
    N, M, Gr, G2 = p.solver_shape
    assert out["subY"].stride() == (N * M, 1, N) and out["B"].stride() == (G2 * M, 1, G2)
    for t in range(3):
        ref, _ = _oracle_inputs(p, out, t)
        Hm = out["H"][t].cpu().numpy().reshape(p.Nr, p.L, p.Nt).transpose(0, 2, 1)     # columns s + Nt*l
        assert rel_err(Hm, ref["H"]) < 2e-6
        assert rel_err(out["Zbar"][t].cpu().numpy(), ref["Zbar"]) < 5e-6
        np.testing.assert_array_equal(out["Omega"][t].cpu().numpy(), ref["Omega"])
        assert rel_err(out["subY"][t].cpu().numpy(), ref["subY"]) < 5e-6
        assert rel_err(out["A"].cpu().numpy(), ref["A"]) < 2e-6
        assert rel_err(out["B"][t].cpu().numpy(), ref["B"]) < 2e-6
        np.testing.assert_allclose(float(out["tau_Y"][t]), ref["tau_Y"], rtol=2e-6)
        np.testing.assert_allclose(float(out["tau_Z"][t]), ref["tau_Z"], rtol=2e-6)
        np.testing.assert_allclose(float(out["rho"][t]), ref["rho"], rtol=5e-5)
        # indx_S: a permutation that orders |vec(Zbar)| descending (fp32 magnitudes: near-ties may swap)
        ix = out["indx_S"][t].cpu().numpy().astype(np.int64) - 1
        assert np.array_equal(np.sort(ix), np.arange(Gr * G2))
        mag = np.abs(ref["Zbar"].reshape(-1, order="F"))[ix]
        assert np.all(np.diff(mag) <= 2e-6 * mag[0])
        lead = 10
        np.testing.assert_array_equal(ix[:lead] + 1, ref["indx_S"][:lead])


def test_conventional_hbf_outputs_match_oracle():
    from jstsp19_amd.system_model import SweepParams, build_trials
    from oracle import system_model as osm
    p = SweepParams(Nt=4, Nr=16, L=3, T=12, Mr=4, snr_db=0.0)
    out = build_trials(p, 0, 2, seed=9, with_hbf=True, want_draws=True, want_H=True)
    Th = p.T_hbf
    for t in range(2):
        ref, d = _oracle_inputs(p, out, t)
        Psi_rows = np.stack([osm.toeplitz_rows(osm.qam4_alphabet()[d["qam_idx"][k]], p.L) for k in range(p.Nt)], axis=2)
        Nn = np.sqrt(p.noise_var / 2) * d["noise"]
        Yc, Wc, Psi_bar, _ = osm.hbf(ref["H"], Nn[:, :Th], Psi_rows[:, :Th, :], Th, p.Nr, osm.create_beamformer(p.Nr, "ZC"))
        Dr = np.exp(-2j * np.pi * np.outer(np.arange(p.Nr), np.arange(p.Gr)) / p.Gr) / np.sqrt(p.Nr)
        Dt = np.exp(-2j * np.pi * np.outer(np.arange(p.Nt), np.arange(p.Gt)) / p.Gt) / np.sqrt(p.Nt)
        assert rel_err(out["Y_hbf"][t].cpu().numpy(), Yc) < 5e-6
        assert rel_err(out["A_hbf"].cpu().numpy(), Wc.conj().T @ Dr) < 2e-6
        Bc = np.concatenate([Dt.conj().T @ Psi_bar[:, :, l] for l in range(p.L)])
        assert rel_err(out["B_hbf"][t].cpu().numpy(), Bc) < 2e-6


def test_draws_are_keyed_by_trial_not_by_batch():
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=2, Nr=8, L=2, T=4, Mr=2)
    a = {k: v.clone() if torch.is_tensor(v) else v for k, v in build_trials(p, 0, 4, seed=5, want_draws=True).items()}
    b = build_trials(p, 2, 2, seed=5, want_draws=True)
    for k in ("subY", "Omega", "B", "Zbar", "indx_S", "gains", "noise", "qam_idx", "tau_Y", "rho"):
        assert torch.equal(a[k][2:], b[k]), k
    c = build_trials(p, 2, 2, seed=5, sweep_idx=1, want_draws=True)
    assert not torch.equal(c["gains"], a["gains"][2:])
    d = build_trials(p, 2, 2, seed=6, want_draws=True)
    assert not torch.equal(d["noise"], a["noise"][2:])


def test_generator_statistics():
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=4, Nr=32, L=4, T=20, Mr=4, Mr_e=32)
    out = build_trials(p, 0, 64, seed=1, want_draws=True)
    nz = torch.view_as_real(out["noise"]).double()
    n = nz.numel()
    assert abs(float(nz.mean())) < 5 / np.sqrt(n)
    assert abs(float(nz.var()) - 1.0) < 5 * np.sqrt(2.0 / n)
    assert abs(float((nz ** 4).mean()) - 3.0) < 0.05                       # Gaussian kurtosis
    assert abs(float((nz[..., 0] * nz[..., 1]).mean())) < 5 / np.sqrt(n / 2)    # real/imag uncorrelated
    g = torch.view_as_real(out["gains"]).double()
    assert abs(float(g.var()) - 0.5) < 0.05                                # 1/sqrt(2) (randn + j randn)
    for k in ("u_r", "u_t"):
        u = out[k].double()
        assert float(u.min()) > 0.0 and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 0.05
    q = out["qam_idx"].flatten().long()
    frac = torch.bincount(q, minlength=4).double() / q.numel()
    assert int(q.max()) == 3 and float((frac - 0.25).abs().max()) < 5 * np.sqrt(0.1875 / q.numel())
    om = out["Omega"]
    assert torch.equal(om.sum(dim=1), torch.full_like(om.sum(dim=1), float(p.Mr)))    # exactly Mr rows per column
    row_freq = om.double().mean(dim=(0, 2))                                 # each row sampled Mr/Mr_e of the time
    assert float((row_freq - p.Mr / p.Mr_e).abs().max()) < 0.02


def test_solver_runs_on_hip_built_inputs_and_recovers_the_channel():
    """End to end on library-built inputs: the genie-aided solve lands well below the LS error."""
    import jstsp19_amd as J
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=4, Nr=32, L=4, T=20, Mr=4, snr_db=5.0)
    inp = build_trials(p, 0, 16, seed=3)
    Sa, _, _ = J.proposed_algorithm_angles(inp["subY"], inp["Omega"], inp["indx_S"], inp["A"], inp["B"], 100,
                                           inp["tau_Y"].numpy(), inp["tau_Z"].numpy(), inp["rho"].numpy(),
                                           "approximate", None, want_ce=False)
    e = J.nmse_spectral(Sa, inp["Zbar"])
    assert float(e.mean()) < 0.2


@pytest.mark.parametrize("kw", [dict(Nt=4, Nr=32, L=4, T=70, ratio=0.75, snr_db=5.0),        # plot_errorVSsnr_approx.m:8-20
                                dict(Nt=3, Nr=16, L=2, T=24, ratio=0.5, snr_db=-5.0, clusters=3, rays=2)])
def test_training_builder_mode_matches_oracle_on_its_own_draws(kw):
    """wideband_hybBF_comm_system_training.m:1-58 + plot_errorVSsnr_approx.m:45-58 as a mode of jstsp_build_trials_c32
    (Gaussian Hermitian-Toeplitz pilots, unitary DFT combiner, round(ratio*Nr) rows per column, rho of :51-53) against the
    oracle's restatement on the library's own draws."""
    from jstsp19_amd.system_model import TrainingParams, build_trials_training
    from oracle import system_model as osm
    p = TrainingParams(**kw)
    out = build_trials_training(p, 3, 3, seed=11, sweep_idx=1, want_draws=True, want_H=True)
    torch.cuda.synchronize()
    Np = p.clusters * p.rays
    params = dict(Nt=p.Nt, Nr=p.Nr, L=p.L, T=p.T, clusters=p.clusters, rays=p.rays, ratio=p.ratio, noise_var=p.noise_var)
    for t in range(3):
        u_r = np.zeros((p.L, Np)); u_r[0] = out["u_r"][t].cpu().numpy()
        u_t = np.zeros((p.L, Np)); u_t[0] = out["u_t"][t].cpu().numpy()
        om = out["Omega"][t].cpu().numpy()
        assert np.all(om.sum(axis=0) == p.Lr)
        rows = np.stack([np.flatnonzero(om[:, j]) for j in range(om.shape[1])])
        d = dict(gains=out["gains"][t].cpu().numpy().astype(complex), u_r=u_r, u_t=u_t,
                 noise=out["noise"][t].cpu().numpy().astype(complex),
                 pilots=out["pilot_sym"][t].cpu().numpy().astype(complex), omega_rows=rows)
        ref = osm.training_inputs_errorVSsnr_approx(params, d)
        np.testing.assert_array_equal(om, ref["Omega"])
        assert rel_err(out["subY"][t].cpu().numpy(), ref["subY"]) < 5e-6
        assert rel_err(out["A"].cpu().numpy(), ref["A"]) < 2e-6
        assert rel_err(out["B"][t].cpu().numpy(), ref["B"]) < 5e-6
        assert rel_err(out["Zbar"][t].cpu().numpy(), ref["Zbar"]) < 5e-6
        np.testing.assert_allclose(float(out["tau_X"][t]), ref["tau_X"], rtol=2e-6)
        np.testing.assert_allclose(float(out["tau_S"][t]), ref["tau_S"], rtol=2e-6)
        np.testing.assert_allclose(float(out["rho"][t]), ref["rho"], rtol=5e-5)
    # Gaussian pilots: unit-variance parts, and the 4-QAM path still reports its alphabet values
    ps = torch.view_as_real(out["pilot_sym"]).double()
    assert abs(float(ps.var()) - 1.0) < 0.2


def test_alg1_vs_alg2_sweep_on_library_built_inputs():
    """run_approx_sweep with its default builder (the library's) gives the curves of the tensor-op builder statistically:
    different generators, same distribution."""
    from jstsp19_amd.montecarlo import run_approx_sweep
    from jstsp19_amd.system_model import TrainingParams
    base = TrainingParams(Nt=4, Nr=32, L=4, T=70, ratio=0.75)
    a = run_approx_sweep(base, [0.0, 10.0], [20], 48, batch=48)
    from torch_builder import builder as torch_builder
    b = run_approx_sweep(base, [0.0, 10.0], [20], 48, batch=48, builder=torch_builder)
    assert a.shape == (1, 2, 2) and torch.isfinite(a).all()
    assert float((a - b).abs().max()) < 0.35 * float(b.max())
    assert float(a[0, 1, 0]) < float(a[0, 0, 0])                # higher SNR, lower error (Alg. 1)
