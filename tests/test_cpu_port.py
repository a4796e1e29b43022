"""oracle/cpu_port.cpp (the timed CPU baseline of bench.py: float64 C++, OpenMP over trials) against the numpy oracle
and its golden fixture - the baseline has to compute the same thing as what it is the baseline of."""
import numpy as np

from conftest import load_golden


def _lib():
    from oracle import build_cpu_port as bp
    return bp, bp.load()


def test_cpu_port_matches_the_reference_native_golden():
    bp, lib = _lib()
    g = load_golden("proposed_refnative")
    S, Y, ce, used = bp.proposed_algorithm(lib, g["subY"][None], g["Omega"][None], g["A"], g["B"][None], int(g["Imax"]),
                                           float(g["tau_Y"]), float(g["tau_Z"]), float(g["rho"]))
    from oracle import solvers as O
    So, Yo, ceo = O.proposed_algorithm(g["subY"], g["Omega"], g["A"], g["B"], int(g["Imax"]), float(g["tau_Y"]),
                                       float(g["tau_Z"]), float(g["rho"]), "approximate")
    assert np.max(np.abs(S[0] - So)) / np.max(np.abs(So)) < 1e-8
    assert np.max(np.abs(Y[0] - Yo)) / np.max(np.abs(Yo)) < 1e-8
    assert np.isinf(ce[0, 0, 2]) and np.isinf(ceo[0, 2])
    fin = np.isfinite(ceo)
    assert np.max(np.abs(ce[0][fin] - ceo[fin]) / np.abs(ceo[fin])) < 1e-7
    assert abs(O.nmse_capped(S[0], g["Zbar"]) - float(g["nmse_approximate"])) < 1e-9


def test_cpu_port_batched_ragged_shapes_and_angles():
    """Shapes that are no multiple of the 16 x 4 register tile, a shared and a per-trial dictionary, the _angles mask,
    several threads."""
    bp, lib = _lib()
    from oracle import solvers as O
    rng = np.random.default_rng(7)
    c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    for (N, M, Gr, G2, batch, shared) in [(9, 21, 7, 10, 3, False), (32, 70, 18, 33, 2, True), (16, 24, 16, 8, 4, False)]:
        A = c(N, Gr) / np.sqrt(N)
        B = c(G2, M) / np.sqrt(G2) if shared else c(batch, G2, M) / np.sqrt(G2)
        S0 = np.zeros((batch, Gr, G2), complex)
        for t in range(batch):
            S0[t].flat[rng.choice(Gr * G2, 5, replace=False)] = 3 * c(5)
        Bt = lambda t: B if shared else B[t]
        Om = (rng.random((batch, N, M)) < 0.4).astype(float)
        subY = Om * (np.stack([A @ S0[t] @ Bt(t) for t in range(batch)]) + 0.05 * c(batch, N, M))
        tY = 1.0 / np.sum(np.abs(subY) ** 2, axis=(1, 2)); tS = np.full(batch, 0.02); rho = 0.15 + 0.05 * rng.random(batch)
        idx = np.stack([np.argsort(-np.abs(S0[t]).reshape(-1, order="F"), kind="stable") + 1 for t in range(batch)])
        for indx in (None, idx):
            S, Y, ce, used = bp.proposed_algorithm(lib, subY, Om, A, B, 14, tY, tS, rho, indx_S=indx, threads=3)
            assert 1 <= used <= 3
            for t in range(batch):
                So, Yo, ceo = O.proposed_algorithm(subY[t], Om[t], A, Bt(t), 14, tY[t], tS[t], rho[t], "approximate",
                                                   indx_S=None if indx is None else indx[t])
                assert np.max(np.abs(S[t] - So)) / np.max(np.abs(So)) < 1e-7, (N, M, t)
                assert np.max(np.abs(Y[t] - Yo)) / max(np.max(np.abs(Yo)), 1e-300) < 1e-7
                fin = np.isfinite(ceo)
                assert np.array_equal(np.isfinite(ce[t]), fin)
                assert np.max(np.abs(ce[t][fin] - ceo[fin]) / np.abs(ceo[fin])) < 1e-6
