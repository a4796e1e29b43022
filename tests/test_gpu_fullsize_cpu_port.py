"""Full-size float64 evidence at BASELINE configs[1] (N=64, M=4096, Gr=64, G2=512, Imax=100): 8 trials of
proposed_algorithm AND proposed_algorithm_angles against oracle/cpu_port.cpp - the float64 C++ restatement that
tests/test_cpu_port.py ties to the numpy oracle and its golden fixture (1e-8), run here with one trial per host thread.
The numpy oracle alone needs 25 s per trial at this size; the port does the 16 solves in about 10 s on 16 cores."""
import numpy as np
import pytest

from conftest import check_below, ce_rel, TOL_S, TOL_CE, TOL_NMSE  # noqa: E402

pytestmark = pytest.mark.gpu


def test_eight_trials_proposed_and_angles_against_the_float64_port():
    import torch
    import jstsp19_amd as J
    from jstsp19_amd.system_model import SweepParams, build_trials
    from oracle import build_cpu_port as bp
    from oracle import solvers as O
    lib = bp.load()
    p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=0.0)
    nt, Imax = 8, 100
    inp = build_trials(p, 4000, nt, seed=303)
    ty, tz, rho = inp["tau_Y"].numpy(), inp["tau_Z"].numpy(), inp["rho"].numpy()
    S, Y, ce = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], Imax, ty, tz, rho, "approximate")
    Sa, Ya, cea = J.proposed_algorithm_angles(inp["subY"], inp["Omega"], inp["indx_S"], inp["A"], inp["B"], Imax, ty, tz, rho,
                                              "approximate", None)
    torch.cuda.synchronize()
    assert J.default_context(0).last_fused_fallbacks() == 0
    h = {k: inp[k].cpu().numpy() for k in ("subY", "Omega", "B", "Zbar", "indx_S")}
    A_h = inp["A"].cpu().numpy()
    for (Sg, Yg, cg), idx in (((S, Y, ce), None), ((Sa, Ya, cea), h["indx_S"])):
        So, Yo, co, used = bp.proposed_algorithm(lib, h["subY"], h["Omega"], A_h, h["B"], Imax, ty, tz, rho, indx_S=idx,
                                                 want_ce=True, threads=nt)
        Sg, Yg, cg = (x.cpu().numpy() for x in (Sg, Yg, cg))
        for t in range(nt):
            zb = h["Zbar"][t]
            assert abs(O.nmse_capped(Sg[t].astype(np.complex128), zb) - O.nmse_capped(So[t], zb)) < 1e-6, (idx is not None, t)
            check_below("fullsize_port.S", np.max(np.abs(Sg[t] - So[t])) / np.max(np.abs(So[t])), TOL_S)
            check_below("fullsize_port.Y", np.max(np.abs(Yg[t] - Yo[t])) / np.max(np.abs(Yo[t])), TOL_S)
            fin = np.isfinite(co[t])
            assert np.array_equal(np.isfinite(cg[t]), fin)
            check_below("fullsize_port.ce", np.max(np.abs(cg[t][fin] - co[t][fin]) / np.abs(co[t][fin])), TOL_CE)
