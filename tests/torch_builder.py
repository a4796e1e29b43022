"""Torch tensor-op construction of the solver inputs (TEST INFRASTRUCTURE since round 4).

The product builds its Monte-Carlo trials with the library's own kernels (jstsp_build_trials_c32 via
jstsp19_amd.system_model.build_trials).  This is the independent second implementation of
plot_errorVSsnr.m:57-136 and wideband_hybBF_comm_system_training.m:1-58 from rounds 1-3, written with torch
tensor ops so that it also runs on the CPU: the CPU tier uses it to exercise the sweep runner (sharding, merging,
the single all-reduce) with the float64 oracle as the solver, and tests/test_system_model.py checks it against
oracle/system_model.py.  It has its own random streams (torch generators keyed by (seed, sweep, trial)): its curves agree
with the library builder's statistically, not sample by sample.

``builder(p, trial_ids, seed, sweep_idx, device, with_hbf)`` below is the hook jstsp19_amd.montecarlo accepts.
"""
import math

import torch

from jstsp19_amd.system_model import SweepParams, TrainingParams


def _trial_seed(seed, sweep_idx, trial_idx):
    # splitmix-style mixing of (seed, sweep, trial) into one 63-bit generator seed
    x = (seed * 0x9E3779B97F4A7C15 + sweep_idx * 0xBF58476D1CE4E5B9 + trial_idx * 0x94D049BB133111EB) & (2 ** 64 - 1)
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & (2 ** 64 - 1)
    x ^= x >> 31
    return x & (2 ** 63 - 1)


def draw_trials(p: SweepParams, trial_ids, seed=20190913, sweep_idx=0, device="cuda"):
    """Draw the random numbers of the given global trial indices (one generator per trial).

    Returns a dict of batched tensors: gains (T,L,Np) complex128, u_r/u_t (T,L,Np) float64,
    noise (T,Nr,T_prop) complex128 (unit-variance real and imaginary parts),
    qam_idx (T,Nt,T_prop) int64 in 0..3, omega_rows (T,T_prop,Mr) int64.
    """
    Np = p.clusters * p.rays
    out = {k: [] for k in ("gains", "u_r", "u_t", "noise", "qam_idx", "omega_rows")}
    g = torch.Generator(device=device)
    for tid in trial_ids:
        g.manual_seed(_trial_seed(seed, sweep_idx, int(tid)))
        gr = torch.randn((2, p.L, Np), generator=g, device=device, dtype=torch.float64)
        out["gains"].append(torch.complex(gr[0], gr[1]) / math.sqrt(2.0))        # wideband_mmwave_channel.m:19
        out["u_r"].append(torch.rand((p.L, Np), generator=g, device=device, dtype=torch.float64))   # :20
        out["u_t"].append(torch.rand((p.L, Np), generator=g, device=device, dtype=torch.float64))   # :22
        nz = torch.randn((2, p.Nr, p.T_prop), generator=g, device=device, dtype=torch.float64)
        out["noise"].append(torch.complex(nz[0], nz[1]))                         # plot_errorVSsnr.m:60
        out["qam_idx"].append(torch.randint(0, 4, (p.Nt, p.T_prop), generator=g, device=device))    # qam4mod.m:8
        # proposed_hbf.m:37-40: randperm(Lr_e)(1:Lr) per column = the Lr smallest of Lr_e uniforms
        keys = torch.rand((p.T_prop, p.Mr_e), generator=g, device=device)
        out["omega_rows"].append(keys.argsort(dim=1)[:, :p.Mr])
    return {k: torch.stack(v) for k, v in out.items()}


def dft_dictionary(Mn, G, device, dtype=torch.complex128):
    """wideband_mmwave_channel.m:9-10 — 1/sqrt(M) exp(-j (0:M-1)' 2 pi (0:G-1)/G)."""
    n = torch.arange(Mn, device=device, dtype=torch.float64)[:, None]
    gg = torch.arange(G, device=device, dtype=torch.float64)[None, :]
    ph = -2.0 * math.pi * n * gg / G
    return (torch.complex(torch.cos(ph), torch.sin(ph)) / math.sqrt(Mn)).to(dtype)


def zc_beamformer(N, device, dtype=torch.complex128):
    """createBeamformer.m:15-16 ('ZC') — 1/sqrt(N) exp(-j 11 (0:N-1)' pi (1:N)/N)."""
    n = torch.arange(N, device=device, dtype=torch.float64)[:, None]
    m = torch.arange(1, N + 1, device=device, dtype=torch.float64)[None, :]
    ph = -11.0 * n * math.pi * m / N
    return (torch.complex(torch.cos(ph), torch.sin(ph)) / math.sqrt(N)).to(dtype)


def _beamformer(p, device):
    """createBeamformer(Nr, kind): 'ZC' (:15-16), 'fft' (:5) and 'ps' (:12-13) — the last two are the same matrix."""
    return zc_beamformer(p.Nr, device) if p.beamformer == "ZC" else dft_dictionary(p.Nr, p.Nr, device)


def _steer(phi, Mn):
    """wideband_mmwave_channel.m:42-52 — exp(-j pi sin(0 - phi) (0:M-1)'), un-normalised."""
    n = torch.arange(Mn, device=phi.device, dtype=torch.float64)
    ph = -math.pi * torch.sin(-phi)[..., None] * n           # (..., Mn)
    return torch.complex(torch.cos(ph), torch.sin(ph))


def _laplacian(u):
    """wideband_mmwave_channel.m:56-62."""
    beta = 1.0 / (1.0 - math.exp(-math.sqrt(2.0) * math.pi / 50.0))
    return beta * (math.exp(-math.sqrt(2.0) / 50.0 * math.pi) - torch.cosh(u))


def _channel(p, draws):
    """wideband_mmwave_channel.m:1-40 for a batch: H (T,Nr,Nt,L), Zbar (T,Gr,L*Gt), Dr, Dt (complex128)."""
    dev = draws["gains"].device
    T = draws["gains"].shape[0]
    Np = p.clusters * p.rays
    c128 = torch.complex128
    Dr = dft_dictionary(p.Nr, p.Gr, dev)
    Dt = dft_dictionary(p.Nt, p.Gt, dev)
    # taps reuse tap 1's steering vectors (:24), cluster c's rays weighted (C - c) (:29)
    Ar1 = _steer(_laplacian(draws["u_r"][:, 0, :]), p.Nr).transpose(1, 2)      # (T, Nr, Np)
    At1 = _steer(_laplacian(draws["u_t"][:, 0, :]), p.Nt).transpose(1, 2)      # (T, Nt, Np)
    w = (p.clusters - torch.arange(Np, device=dev) // p.rays).to(torch.float64)   # (Np,)
    coef = draws["gains"] * w / math.sqrt(Np)                                   # (T, L, Np)   :33
    # H[t,:,:,l] = Ar1 diag(coef[t,l]) At1^H
    H = torch.einsum("trp,tlp,tsp->trsl", Ar1, coef.to(c128), At1.conj())       # (T, Nr, Nt, L)
    Z = torch.einsum("rg,trsl,sh->tghl", Dr.conj(), H, Dt)                      # Dr' H_l Dt   :35
    Zbar = Z.permute(0, 1, 3, 2).reshape(T, p.Gr, p.L * p.Gt)                   # [Z_1 ... Z_L] :38  (col = l*Gt + h)
    return H, Zbar, Dr, Dt


def _toeplitz_rows(sym, L):
    """Rows 1..L of the Hermitian ``toeplitz(s_k)`` of every pilot sequence: (T,Nt,Tp) -> (T,Nt,Tp,L)
    with ``[..., k, :, l] = Psi_bar(k,:,l)`` (proposed_hbf.m:15-18)."""
    Tp = sym.shape[-1]
    j = torch.arange(Tp, device=sym.device)
    rows = []
    for l in range(L):
        d = j - l
        r = sym[:, :, d.abs()]
        rows.append(torch.where((d >= 0)[None, None, :], r, r.conj()))
    return torch.stack(rows, dim=-1)


def build_inputs(p: SweepParams, draws, out_dtype=torch.complex64, with_hbf=False):
    """plot_errorVSsnr.m:57-136 for a batch of trials.

    Returns a dict of device tensors, matrices column-major per problem as the C ABI wants:
      subY (T,N,M), Omega (T,N,M) float32, A (N,Gr) [shared: ZC x DFT is trial-independent],
      B (T,G2,M), Zbar (T,Gr,G2) complex128, H (T,Nr,Nt,L) complex128,
      tau_Y, tau_Z, rho (T,) float64 on the host side of the C ABI (returned as CPU tensors),
      indx_S (T, Gr*G2) int32 1-based (plot_errorVSsnr.m:143).
    """
    from jstsp19_amd.solvers import colmajor
    dev = draws["gains"].device
    T = draws["gains"].shape[0]
    c128 = torch.complex128
    H, Zbar, Dr, Dt = _channel(p, draws)
    # --- pilots: Psi_bar(k,:,l) = row l of toeplitz(s_k) (proposed_hbf.m:17), Hermitian Toeplitz
    s = 1.0 / math.sqrt(2.0)
    alphabet = torch.tensor([complex(s, s), complex(-s, s), complex(s, -s), complex(-s, -s)], device=dev, dtype=c128)
    sym = alphabet[draws["qam_idx"]]                                            # (T, Nt, T_prop)
    Tp = p.T_prop
    Psi_bar = _toeplitz_rows(sym, p.L)                                          # (T, Nt, T_prop, L)
    # --- received signal, sampling mask, measurement (proposed_hbf.m:13-42)
    Y = torch.einsum("trsl,tsjl->trj", H, Psi_bar)                              # sum_l H_l Psi_bar_l   :19
    R = Y + math.sqrt(p.noise_var / 2.0) * draws["noise"]                       # :22, plot_errorVSsnr.m:60
    Wfull = _beamformer(p, dev)
    W_e = Wfull[:, :p.Mr_e]                                                     # :11, plot_errorVSsnr.m:124
    Omega = torch.zeros((T, p.Mr_e, Tp), device=dev, dtype=torch.float64)
    Omega.scatter_(1, draws["omega_rows"].transpose(1, 2), 1.0)                 # :36-41
    subY = Omega * torch.einsum("re,trj->tej", W_e.conj(), R)                   # :42
    # --- hyper-parameters (plot_errorVSsnr.m:127-130): eigs() returns the 6 largest => sigma_6^2
    fro2 = (subY.abs() ** 2).sum(dim=(1, 2))
    tau_Y = 1.0 / fro2
    tau_Z = 0.5 / (Zbar.abs() ** 2).sum(dim=(1, 2))
    sv = torch.linalg.svdvals(subY)
    rho = p.rho_scale * torch.sqrt(sv[:, 0 if p.rho_rule == "max" else 5] ** 2 / fro2)
    # --- dictionary factors (:132-136)
    A = W_e.conj().transpose(0, 1) @ Dr                                         # Mr_e x Gr
    B = torch.einsum("sh,tsjl->tlhj", Dt.conj(), Psi_bar).reshape(T, p.L * p.Gt, Tp)   # rows l*Gt + h
    absz = Zbar.transpose(1, 2).reshape(T, -1).abs()                            # vec order (column-major)
    indx_S = (torch.argsort(absz, dim=1, descending=True, stable=True) + 1).to(torch.int32)
    extra = {}
    if with_hbf:
        # conventional HBF with all Nr RF chains over a shorter frame (plot_errorVSsnr.m:73-80, hbf.m:1-26)
        Th = p.T_hbf
        Wc = Wfull                                                              # Mr_hbf = Nr columns (:11,:73)
        Psi_c = Psi_bar[:, :, :Th, :]                                           # Psi_i(1:T_hbf,1:T_hbf,:) rows 1..L
        Rc = torch.einsum("trsl,tsjl->trj", H, Psi_c) + math.sqrt(p.noise_var / 2.0) * draws["noise"][:, :, :Th]
        Y_hbf = torch.einsum("re,trj->tej", Wc.conj(), Rc)                      # hbf.m:24
        A_hbf = Wc.conj().transpose(0, 1) @ Dr                                  # :74
        B_hbf = torch.einsum("sh,tsjl->tlhj", Dt.conj(), Psi_c).reshape(T, p.L * p.Gt, Th)   # :75-78
        extra = dict(Y_hbf=colmajor(Y_hbf.to(out_dtype)), A_hbf=colmajor(A_hbf.to(out_dtype)),
                     B_hbf=colmajor(B_hbf.to(out_dtype)))
    return dict(**extra, subY=colmajor(subY.to(out_dtype)), Omega=colmajor(Omega.to(torch.float32)),
                A=colmajor(A.to(out_dtype)), B=colmajor(B.to(out_dtype)), Zbar=Zbar, H=H,
                tau_Y=tau_Y.cpu(), tau_Z=tau_Z.cpu(), rho=rho.cpu(), indx_S=indx_S)


def draw_trials_training(p: TrainingParams, trial_ids, seed=20190913, sweep_idx=0, device="cuda"):
    """Random numbers of plot_errorVSsnr_approx.m:45-46 for the given global trial indices:
    gains, u_r, u_t as ``draw_trials``; noise (T,Nr,T) and pilots (T,Nt,T) complex128 with unit-variance
    parts; omega_rows (T,T,Lr) int64."""
    Np = p.clusters * p.rays
    out = {k: [] for k in ("gains", "u_r", "u_t", "noise", "pilots", "omega_rows")}
    g = torch.Generator(device=device)
    for tid in trial_ids:
        g.manual_seed(_trial_seed(seed, sweep_idx, int(tid)))
        gr = torch.randn((2, p.L, Np), generator=g, device=device, dtype=torch.float64)
        out["gains"].append(torch.complex(gr[0], gr[1]) / math.sqrt(2.0))        # wideband_mmwave_channel.m:19
        out["u_r"].append(torch.rand((p.L, Np), generator=g, device=device, dtype=torch.float64))
        out["u_t"].append(torch.rand((p.L, Np), generator=g, device=device, dtype=torch.float64))
        nz = torch.randn((2, p.Nr, p.T), generator=g, device=device, dtype=torch.float64)
        out["noise"].append(torch.complex(nz[0], nz[1]))                         # ...training.m:16
        pl = torch.randn((2, p.Nt, p.T), generator=g, device=device, dtype=torch.float64)
        out["pilots"].append(torch.complex(pl[0], pl[1]))                        # :20
        keys = torch.rand((p.T, p.Nr), generator=g, device=device)               # :50-51 randperm(Nr)(1:Lr)
        out["omega_rows"].append(keys.argsort(dim=1)[:, :p.Lr])
    return {k: torch.stack(v) for k, v in out.items()}


def build_inputs_training(p: TrainingParams, draws, out_dtype=torch.complex64):
    """wideband_hybBF_comm_system_training.m:1-58 + plot_errorVSsnr_approx.m:45-58 for a batch of trials.

    Same layout as ``build_inputs``: subY (T,N,M), Omega float32, A (N,Gr) shared (unitary DFT combiner x DFT
    dictionary), B (T,L*Nt,M), Zbar/H complex128, tau_X, tau_S, rho (T,) float64 CPU tensors.
    """
    from jstsp19_amd.solvers import colmajor
    dev = draws["gains"].device
    T = draws["gains"].shape[0]
    H, Zbar, Dr, Dt = _channel(p, draws)                                        # plot_errorVSsnr_approx.m:45
    Psi_bar = _toeplitz_rows(draws["pilots"] / math.sqrt(2.0), p.L)             # ...training.m:19-22,:28
    n = torch.arange(p.Nr, device=dev, dtype=torch.float64)
    ph = -2.0 * math.pi * n[:, None] * n[None, :] / p.Nr
    W = torch.complex(torch.cos(ph), torch.sin(ph)) / math.sqrt(p.Nr)           # :10  fft(eye(Nr))/sqrt(Nr)
    R = torch.einsum("trsl,tsjl->trj", H, Psi_bar) + math.sqrt(p.noise_var / 2.0) * draws["noise"]   # :16,:30,:33
    Omega = torch.zeros((T, p.Nr, p.T), device=dev, dtype=torch.float64)
    Omega.scatter_(1, draws["omega_rows"].transpose(1, 2), 1.0)                 # :48-53
    subY = Omega * torch.einsum("re,trj->tej", W.conj(), R)                     # :54
    fro2 = (subY.abs() ** 2).sum(dim=(1, 2))
    tau_X = 1.0 / fro2                                                          # plot_errorVSsnr_approx.m:50
    tau_S = tau_X / 2.0                                                         # :51
    sv = torch.linalg.svdvals(subY)
    rho = torch.sqrt(sv[:, 5] ** 2 * (tau_X + tau_S) / 2.0)                     # :52-53  eigs() -> sigma_6^2
    A = W.conj().transpose(0, 1) @ Dr                                           # :54
    B = torch.einsum("sh,tsjl->tlhj", Dt.conj(), Psi_bar).reshape(T, p.L * p.Nt, p.T)   # :55-58
    return dict(subY=colmajor(subY.to(out_dtype)), Omega=colmajor(Omega.to(torch.float32)),
                A=colmajor(A.to(out_dtype)), B=colmajor(B.to(out_dtype)), Zbar=Zbar, H=H,
                tau_X=tau_X.cpu(), tau_S=tau_S.cpu(), rho=rho.cpu())



def builder(p, trial_ids, seed, sweep_idx, device, with_hbf=False):
    """The input-builder hook of jstsp19_amd.montecarlo (run_points, run_approx_sweep, ...)."""
    if isinstance(p, TrainingParams):
        return build_inputs_training(p, draw_trials_training(p, list(trial_ids), seed=seed, sweep_idx=sweep_idx, device=device))
    return build_inputs(p, draw_trials(p, list(trial_ids), seed=seed, sweep_idx=sweep_idx, device=device), with_hbf=with_hbf)
