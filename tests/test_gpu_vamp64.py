"""Float64 VAMP on the device (csrc/vamp64.hip; jstsp_vamp_c64 / jstsp_vamp_kron_c64) against the float64 oracle AT THE REFERENCE'S
OPERATING POINT: nitMax = 100, sigma = 1, no stopping rule (benchmark_algorithms/vamp.m:9,38,45; VampGlmEst.m:505-511).

The fp32-storage path (vamp.hip) can follow the float64 recurrences for about 12 iterations only - the iteration amplifies a rounding
difference ~1e9-fold over its 100 iterations (tests/test_oracle.py) - and is compared statistically at 100 (tests/test_gpu_baselines.py).
The float64 path leaves 1e-16 x 1e9: here the reference's actual output is compared PER TRIAL: x to 1e-5 of max|x|, the capped spectral
NMSE (plot_errorVSsnr.m:103-106) to 1e-6."""
import numpy as np
import pytest

from conftest import check_below, rel_err

pytestmark = pytest.mark.gpu

TOL_X = 1e-5
TOL_NMSE = 1e-6


def _hbf_trials(db, nt, seed):
    """the conventional-HBF system of plot_errorVSsnr.m:73-101 at the reference-native parameters, as complex128 host arrays"""
    import jstsp19_amd as J
    from jstsp19_amd.system_model import SweepParams, build_trials
    p = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4, snr_db=db)
    inp = build_trials(p, 0, nt, seed=seed, with_hbf=True)
    Bh = inp["B_hbf"].cpu().numpy().astype(np.complex128)
    Yh = inp["Y_hbf"].cpu().numpy().astype(np.complex128)
    A = inp["A_hbf"].cpu().numpy().astype(np.complex128)
    Gb = Bh @ Bh.conj().transpose(0, 2, 1)                    # (B*B'), plot_errorVSsnr.m:79
    Ym = Yh @ Bh.conj().transpose(0, 2, 1)                    # Y*B', :80
    return A, Gb, Ym, inp["Zbar"].cpu().numpy().astype(np.complex128)


def test_vamp_kron_float64_at_100_iterations_per_trial_three_snr_points():
    """16 trials at each of -6, 3 and 12 dB through jstsp_vamp_kron_c64 (host arrays, one batched call per point) against
    oracle.vamp.vamp_kron on the same inputs, nit = 100, numOfnz = 100 (plot_errorVSsnr.m:26,100)."""
    import jstsp19_amd as J
    from oracle import solvers as O
    from oracle import vamp as V
    nt = 16
    for db in (-6.0, 3.0, 12.0):
        A, Gb, Ym, Zb = _hbf_trials(db, nt, seed=616)
        X = np.asarray(J.vamp_kron(Ym, A, Gb, 1.0, 100))
        assert X.dtype == np.complex128 and X.shape == (nt,) + Zb.shape[1:]
        for t in range(nt):
            ref = V.vamp_kron(Ym[t], A, Gb[t], 1.0, 100)
            check_below("vamp64.kron.x", rel_err(X[t], ref), TOL_X)
            check_below("vamp64.kron.nmse", abs(O.nmse_capped(X[t], Zb[t]) - O.nmse_capped(ref, Zb[t])), TOL_NMSE)


def test_vamp_dense_float64_is_the_reference_call_at_the_drivers_size():
    """The drivers' own call: Phi = kron((B*B').', A) (512 x 512), y = vec(Y*B'), x = vamp(y, Phi, 1, numOfnz) - through
    jstsp_vamp_c64 (float64 Jacobi of the order-512 Gram) against the LITERAL restatement (dense real-stacked matrix, full SVD,
    vamp.m line by line) and against the factored float64 call, at nit = 100; device-resident complex128 tensors as well."""
    import torch
    import jstsp19_amd as J
    from oracle import solvers as O
    from oracle import vamp as V
    nt = 3
    A, Gb, Ym, Zb = _hbf_trials(6.0, nt, seed=99)
    Phi = np.stack([np.kron(Gb[t].T, A) for t in range(nt)])
    y = np.stack([Ym[t].reshape(-1, order="F") for t in range(nt)])
    x = np.asarray(J.vamp(y, Phi, 1.0, 100))
    xk = np.asarray(J.vamp_kron(Ym, A, Gb, 1.0, 100))
    for t in range(nt):
        ref = V.vamp_literal(y[t], Phi[t], 1.0, 100)
        check_below("vamp64.dense.x", rel_err(x[t], ref), TOL_X)
        check_below("vamp64.dense_vs_kron.x", rel_err(x[t], xk[t].reshape(-1, order="F")), TOL_X)
        X = x[t].reshape(Zb[t].shape, order="F")
        check_below("vamp64.dense.nmse", abs(O.nmse_capped(X, Zb[t]) - O.nmse_capped(ref.reshape(Zb[t].shape, order="F"), Zb[t])), TOL_NMSE)
    # device arrays (JSTSP_DEVICE): the same bits as the host call
    dev = torch.device("cuda:0")
    cm = lambda a: J.colmajor(torch.from_numpy(np.ascontiguousarray(a)).to(dev))
    xd = J.vamp_kron(cm(Ym), cm(A), cm(Gb), 1.0, 100)
    torch.cuda.synchronize()
    assert xd.dtype == torch.complex128 and np.array_equal(xd.cpu().numpy(), xk)


def test_vamp_float64_tall_system_and_few_iterations_match_tightly():
    """M > N (VampGlmEst.m:407-411) and the first iterations, where no amplification has happened yet: 1e-12."""
    import jstsp19_amd as J
    from oracle import vamp as V
    from conftest import load_golden
    g = load_golden("vamp_tall")
    for k, nit in enumerate(g["nits"]):
        x = np.asarray(J.vamp(g["y"].astype(np.complex128), g["A"].astype(np.complex128), float(g["sigma"]), float(g["L"]), nit=int(nit)))
        assert rel_err(x, g["x_dense"][k]) < 1e-10
        xk = np.asarray(J.vamp_kron(g["Y"].astype(np.complex128), g["Af"].astype(np.complex128), g["Gb"].astype(np.complex128), float(g["sigma"]),
                                    float(g["Lk"]), nit=int(nit)))
        assert rel_err(xk, g["x_kron"][k]) < 1e-10
    g = load_golden("vamp")
    for nit in (1, 5, 12):
        ref = V.vamp_kron(g["Y"], g["A"], g["Gb"], float(g["sigma"]), float(g["L"]), nit=nit)
        out = np.asarray(J.vamp_kron(g["Y"].astype(np.complex128), g["A"].astype(np.complex128), g["Gb"].astype(np.complex128), float(g["sigma"]),
                                     float(g["L"]), nit=nit))
        assert rel_err(out, ref) < 1e-10, (nit, rel_err(out, ref))
