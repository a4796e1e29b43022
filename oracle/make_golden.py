"""Generate the committed golden fixtures under tests/golden/ (TEST INFRASTRUCTURE).

The reference is MATLAB and cannot run here (no matlab/octave in the image),
and it ships no test vectors of its own, so these fixtures are produced by the oracle's
LITERAL restatement (dense kron / lu / full svd in the reference's operation order) on
seeded numpy inputs.  They pin the structured oracle, the HIP path and any later refactor
to the same numbers.  Run:  python -m oracle.make_golden
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

from . import solvers as S
from . import system_model as sm

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

PARAMS_SMALL = dict(Nt=2, Nr=8, Mr_e=8, Gr=8, Gt=2, clusters=2, rays=3, L=3, Mr=3, T=5)
# reference-native parameters, plot_errorVSsnr.m:8-25
PARAMS_REF = dict(Nt=4, Nr=32, Mr_e=32, Gr=32, Gt=4, clusters=2, rays=3, L=4, Mr=4, T=35)


def trial(params, snr_db, seed):
    p = dict(params)
    p["noise_var"] = 10 ** (-snr_db / 10)          # plot_errorVSsnr.m:49
    rng = np.random.default_rng(seed)
    d = sm.draw_trial(rng, p)
    return p, d, sm.training_inputs_errorVSsnr(p, d)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %s (%.1f KiB)" % (path, os.path.getsize(path) / 1024))


def gen_proposed(name, params, snr_db, seed, Imax, literal=True, types=("approximate",)):
    p, d, inp = trial(params, snr_db, seed)
    out = dict(subY=inp["subY"], Omega=inp["Omega"], A=inp["A"], B=inp["B"], Zbar=inp["Zbar"],
               indx_S=inp["indx_S"], tau_Y=inp["tau_Y"], tau_Z=inp["tau_Z"], rho=inp["rho"], Imax=Imax,
               snr_db=snr_db, seed=seed, literal=literal)
    fn = S.proposed_algorithm_literal if literal else S.proposed_algorithm
    for ty in types:
        t0 = time.time()
        Sx, Y, ce = fn(inp["subY"], inp["Omega"], inp["A"], inp["B"], Imax, inp["tau_Y"], inp["tau_Z"],
                       inp["rho"], ty)
        out["S_" + ty], out["Y_" + ty], out["ce_" + ty] = Sx, Y, ce
        out["nmse_" + ty] = S.nmse_capped(Sx, inp["Zbar"])
        print("  %s %s: nmse %.6f (%.1fs)" % (name, ty, out["nmse_" + ty], time.time() - t0))
    t0 = time.time()
    Sa, Ya, cea = fn(inp["subY"], inp["Omega"], inp["A"], inp["B"], Imax, inp["tau_Y"], inp["tau_Z"],
                     inp["rho"], "approximate", indx_S=inp["indx_S"])
    out["S_angles"], out["Y_angles"], out["ce_angles"] = Sa, Ya, cea
    out["nmse_angles"] = S.nmse_capped(Sa, inp["Zbar"])
    print("  %s angles: nmse %.6f (%.1fs)" % (name, out["nmse_angles"], time.time() - t0))
    save(name, **out)


def gen_svt():
    rng = np.random.default_rng(11)
    out = {}
    for k, (r, c) in enumerate([(6, 9), (9, 6), (32, 140), (16, 16)]):
        lowrank = (rng.standard_normal((r, 3)) + 1j * rng.standard_normal((r, 3))) @ \
                  (rng.standard_normal((3, c)) + 1j * rng.standard_normal((3, c)))
        Y = lowrank + 0.1 * (rng.standard_normal((r, c)) + 1j * rng.standard_normal((r, c)))
        tau = float(0.3 * np.linalg.svd(Y, compute_uv=False)[1])
        out["Y%d" % k], out["tau%d" % k], out["X%d" % k] = Y, tau, S.svt(Y, tau)
    out["n"] = 4
    save("svt", **out)


def gen_omp():
    rng = np.random.default_rng(12)
    out = {}
    # (a) closed-form KAT: unitary DFT dictionary, noiseless k-sparse vector => exact recovery
    n, k = 32, 5
    F = np.exp(-2j * np.pi * np.outer(np.arange(n), np.arange(n)) / n) / np.sqrt(n)
    x = np.zeros(n, complex)
    sup = np.sort(rng.choice(n, k, replace=False))
    x[sup] = (rng.standard_normal(k) + 1j * rng.standard_normal(k)) + 2 * np.sign(rng.standard_normal(k))
    v = F @ x
    xh, idx, _, T = S.omp_literal(F, v, k)
    out.update(A0=F, v0=v, m0=k, x0=xh, idx0=idx, T0=T, xtrue0=x)
    # (b) random over-complete dictionary with noise, m larger than the sparsity
    meas, size_d, m = 24, 40, 8
    A = (rng.standard_normal((meas, size_d)) + 1j * rng.standard_normal((meas, size_d))) / np.sqrt(2 * meas)
    x = np.zeros(size_d, complex)
    x[rng.choice(size_d, 4, replace=False)] = rng.standard_normal(4) + 1j * rng.standard_normal(4)
    v = A @ x + 0.01 * (rng.standard_normal(meas) + 1j * rng.standard_normal(meas))
    xh, idx, _, T = S.omp_literal(A, v, m)
    out.update(A1=A, v1=v, m1=m, x1=xh, idx1=idx, T1=T)
    # (c) Kronecker dictionary (conventional-HBF baseline, plot_errorVSdelays.m:77 style)
    p, d, inp = trial(PARAMS_SMALL, 5.0, 5)
    Af, Bf = inp["A"], inp["B"]
    y = S.vec(Af @ inp["Zbar"] @ Bf) + 0.01 * (rng.standard_normal(Af.shape[0] * Bf.shape[1])
                                              + 1j * rng.standard_normal(Af.shape[0] * Bf.shape[1]))
    Phi = np.kron(Bf.T, Af)
    xh, idx, _, T = S.omp_literal(Phi, y, 6)
    out.update(Af2=Af, Bf2=Bf, y2=y, m2=6, x2=xh, idx2=idx)
    save("omp", **out)


def gen_sparse_admm():
    rng = np.random.default_rng(13)
    Mr, Mt = 8, 6
    Dr = np.exp(-2j * np.pi * np.outer(np.arange(Mr), np.arange(Mr)) / Mr) / np.sqrt(Mr)
    Dt = np.exp(-2j * np.pi * np.outer(np.arange(Mt), np.arange(Mt)) / Mt) / np.sqrt(Mt)
    Sp = np.zeros((Mr, Mt), complex)
    Sp[rng.integers(0, Mr, 4), rng.integers(0, Mt, 4)] = rng.standard_normal(4) + 1j * rng.standard_normal(4)
    H = Dr @ Sp @ Dt.conj().T
    OH = H + 0.05 * (rng.standard_normal((Mr, Mt)) + 1j * rng.standard_normal((Mr, Mt)))
    Sx, ce = S.sparse_admm_literal(H, OH, Dr, Dt, 40)
    # non-unitary dictionaries (still Gr*Gt == Mr*Mt): exercises the eigen-solve
    Dr2 = Dr + 0.2 * (rng.standard_normal((Mr, Mr)) + 1j * rng.standard_normal((Mr, Mr))) / np.sqrt(Mr)
    Dt2 = Dt + 0.2 * (rng.standard_normal((Mt, Mt)) + 1j * rng.standard_normal((Mt, Mt))) / np.sqrt(Mt)
    Sx2, ce2 = S.sparse_admm_literal(H, OH, Dr2, Dt2, 40)
    save("sparse_admm", Htrue=H, OH=OH, Dr=Dr, Dt=Dt, Imax=40, S=Sx, ce=ce, Dr2=Dr2, Dt2=Dt2, S2=Sx2, ce2=ce2)


def gen_mc():
    rng = np.random.default_rng(14)
    Mr, Mt = 10, 14
    H = (rng.standard_normal((Mr, 2)) + 1j * rng.standard_normal((Mr, 2))) @ \
        (rng.standard_normal((2, Mt)) + 1j * rng.standard_normal((2, Mt)))
    Omega = (rng.random((Mr, Mt)) < 0.6).astype(float)
    OH = Omega * (H + 0.01 * (rng.standard_normal((Mr, Mt)) + 1j * rng.standard_normal((Mr, Mt))))
    tau, rho, Imax = 0.5, 0.1, 60
    X1 = S.mc_svt(OH, Omega, Imax, tau, rho)
    X2, ce2 = S.mc_admm_literal(H, OH, Omega, Imax, tau, rho)
    save("mc", Htrue=H, OH=OH, Omega=Omega, tau=tau, rho=rho, Imax=Imax, X_svt=X1, X_admm=X2, ce_admm=ce2)


def gen_vamp():
    """vamp.m on the conventional-HBF system of plot_errorVSsnr.m:73-101 (small parameters):
    Phi = kron((B*B').', A), y = vec(Y_hbf*B'), sigma = 1, L = numOfnz."""
    from . import vamp as V
    rng = np.random.default_rng(15)
    p = dict(PARAMS_SMALL)
    p["noise_var"] = 10 ** (-1.0)
    d = sm.draw_trial(rng, p)
    H, Zbar, _, _, Dr, Dt = sm.wideband_mmwave_channel(p["L"], p["Nr"], p["Nt"], 2, 3, p["Gr"], p["Gt"], d["gains"],
                                                       d["u_r"], d["u_t"])
    T_hbf = 8
    Psi_rows = np.stack([sm.toeplitz_rows(sm.qam4_alphabet()[d["qam_idx"][k]], p["L"]) for k in range(p["Nt"])], axis=2)
    Nn = np.sqrt(p["noise_var"] / 2) * d["noise"]
    Yc, Wc, Psi_bar, _ = sm.hbf(H, Nn[:, :T_hbf], Psi_rows[:, :T_hbf, :], T_hbf, p["Nr"], sm.create_beamformer(p["Nr"], "ZC"))
    A = Wc.conj().T @ Dr                                                       # :74
    B = np.concatenate([Dt.conj().T @ Psi_bar[:, :, l] for l in range(p["L"])])   # :75-78
    Gb = B @ B.conj().T
    Phi = np.kron(Gb.T, A)                                                     # :79
    Ym = Yc @ B.conj().T                                                       # :80 (before vec)
    y = S.vec(Ym)
    t0 = time.time()
    x_lit = V.vamp_literal(y, Phi, 1, 12)                                      # :100, numOfnz scaled down
    x_kron = V.vamp_kron(Ym, A, Gb, 1, 12)
    print("  vamp literal vs kron: %.2e (%.1fs)" % (np.abs(x_lit - S.vec(x_kron)).max(), time.time() - t0))
    save("vamp", A=A, B=B, Gb=Gb, Y=Ym, Phi=Phi, y=y, sigma=1.0, L=12, x=x_lit, Zbar=Zbar)


def gen_vamp_tall():
    """The M > N branch of VampGlmEst.m:407-411 (V, d from eig(A'A), :196-218): a dense 30 x 12 dictionary and a Kronecker one
    with Na = 10 > Gr = 4.  Outputs of the LITERAL restatement (real-stacked matrix) per iteration count."""
    from . import vamp as V
    rng = np.random.default_rng(17)
    r = lambda *sh: rng.standard_normal(sh) + 1j * rng.standard_normal(sh)
    M, N = 30, 12
    A = r(M, N) / np.sqrt(M)
    x0 = np.zeros(N, complex); x0[[2, 7, 9]] = [2, -1j, 1 + 1j]
    y = A @ x0 + 0.05 * r(M)
    Na, Gr, G2 = 10, 4, 5
    Af = r(Na, Gr) / 3
    Bh = r(G2, 9) / 3
    Gb = Bh @ Bh.conj().T
    X0 = np.zeros((Gr, G2), complex); X0[1, 2] = 2; X0[3, 0] = -1.5j
    Y = Af @ X0 @ Gb + 0.02 * r(Na, G2)
    Phi = np.kron(Gb.T, Af)
    nits = (2, 5, 12)
    xd = np.stack([V.vamp_literal(y, A, 1.0, 3, nit=n) for n in nits])
    xk = np.stack([V.vamp_literal(S.vec(Y), Phi, 1.0, 4, nit=n).reshape(Gr, G2, order="F") for n in nits])
    save("vamp_tall", A=A, y=y, L=3, Af=Af, Gb=Gb, Y=Y, Lk=4, sigma=1.0, nits=np.array(nits), x_dense=xd, x_kron=xk)


def gen_baselines2():
    """Joint OMP (published simultaneous OMP: sparse-plex is un-vendored, parity unpinned), pinv / LS with an
    ill-conditioned square pilot factor (plot_errorVSsnr.m:83), the TSSR / SVT-based recipes (:151-162) and the rate of
    plot_rateVSframelength.m:81 on seeded inputs."""
    rng = np.random.default_rng(16)
    r = lambda *sh: rng.standard_normal(sh) + 1j * rng.standard_normal(sh)
    N, Gr, Sx = 16, 16, 9
    A = r(N, Gr) / np.sqrt(N)
    Z0 = np.zeros((Gr, Sx), complex)
    Z0[[2, 7, 11]] = 3 * r(3, Sx)
    Y = A @ Z0 + 0.05 * r(N, Sx)
    Zl2, sl2 = S.mmv_omp(A, Y, 5, "l2")
    Zl1, sl1 = S.mmv_omp(A, Y, 5, "l1")
    # LS with cond(B) = 1e3 (square B as the drivers' T_hbf == G2)
    G2 = 8
    U, _ = np.linalg.qr(r(G2, G2)); V, _ = np.linalg.qr(r(G2, G2))
    B = (U * np.logspace(0, -3, G2)) @ V.conj().T
    Yl = A @ r(Gr, G2) @ B
    S_ls = np.linalg.pinv(A) @ Yl @ np.linalg.pinv(B)
    # TSSR on a small completion problem
    M = 20
    Bw = r(G2, M) / np.sqrt(G2)
    Om = (rng.random((N, M)) < 0.6).astype(float)
    Zt = np.zeros((Gr, G2), complex); Zt[[1, 9]] = 2 * r(2, G2)
    Yp = Om * (A @ Zt @ Bw + 0.02 * r(N, M))
    S_tssr, Y_svt, S_svt = S.tssr(Yp, Om, A, Bw, 25, 0.05, 0.1, 4)
    Zb = r(Gr, G2)
    Sx2 = Zb + 0.2 * r(Gr, G2)
    save("baselines2", A=A, Y=Y, K=5, Z_l2=Zl2, sup_l2=sl2, Z_l1=Zl1, sup_l1=sl1, B_ls=B, Y_ls=Yl, S_ls=S_ls,
         B_t=Bw, Omega_t=Om, Y_t=Yp, tau_t=0.05, rho_t=0.1, Imax_t=25, K_t=4, S_tssr=S_tssr, Y_svt=Y_svt, S_svt=S_svt,
         Zbar_r=Zb, S_r=Sx2, noise_var=0.3, rate=S.rate(Sx2, Zb, 0.3))


def main():
    t0 = time.time()
    gen_proposed("proposed_small", PARAMS_SMALL, 5.0, 1, 30, literal=True, types=("approximate", "std"))
    gen_proposed("proposed_small_lowsnr", PARAMS_SMALL, -10.0, 2, 30, literal=True, types=("approximate",))
    gen_svt()
    gen_omp()
    gen_sparse_admm()
    gen_mc()
    gen_vamp()
    gen_vamp_tall()
    gen_baselines2()
    if "--fast" not in sys.argv:
        # reference-native shape (N=32, M=140, Gr=32, G2=16): dense K1 is 4480^2, K2 4480x512
        gen_proposed("proposed_refnative", PARAMS_REF, 5.0, 3, 100, literal=True, types=("approximate",))
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
