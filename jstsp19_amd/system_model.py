"""Batched, device-resident construction of the solver inputs (the caller side of the hot path).

What plot_errorVSsnr.m:57-136 does per trial on the host — channel, pilots, noise, the
random-spatial-sampling measurement, the dictionary factors A, B and the hyper-parameters —
done here for a whole batch of Monte-Carlo trials with torch tensor ops on the GPU, so that
the solver's inputs are born in HBM.  Not the timed hot path; torch is
used as plumbing.  Every function cites the reference lines it follows (paths relative to
/root/reference) and reproduces their quirks (tap-1 steering reuse, cumulative cluster sum,
Hermitian Toeplitz pilots, sigma_6 in rho).

Random numbers come from a counter-style scheme keyed by (seed, sweep index, global trial
index), so a trial's inputs do not depend on how trials are sharded over GPUs.
"""
from __future__ import annotations

import math

import torch

__all__ = ["SweepParams", "draw_trials", "build_inputs", "build_trials", "zc_beamformer", "dft_dictionary",
           "TrainingParams", "draw_trials_training", "build_inputs_training", "build_trials_training"]


class SweepParams:
    """Parameters of one sweep point — names follow plot_errorVSsnr.m:8-25.

    What the sibling drivers change in the construction: ``beamformer`` ('ZC', or 'fft' / 'ps' — the same unitary
    DFT matrix, createBeamformer.m:5,12-13), ``rho_rule`` ('min': min(eigs(Y'Y)) = the 6th largest eigenvalue,
    'max': the largest), ``rho_scale`` (plot_errorVSzy.m:65 halves rho) and ``T_prop`` (plot_errorVSadmmiters.m:21 and
    plot_errorVSzy.m:30 use the frame length itself instead of T*Nt).
    """

    def __init__(self, Nt, Nr, L, T, Mr, Mr_e=None, Gr=None, Gt=None, clusters=2, rays=3, snr_db=5.0,
                 beamformer="ZC", rho_rule="min", rho_scale=1.0, T_prop=None):
        self.Nt, self.Nr, self.L, self.T, self.Mr = Nt, Nr, L, T, Mr
        self.Mr_e = Nr if Mr_e is None else Mr_e
        self.Gr = Nr if Gr is None else Gr
        self.Gt = Nt if Gt is None else Gt
        self.clusters, self.rays = clusters, rays
        self.snr_db = float(snr_db)
        if beamformer not in ("ZC", "fft", "ps") or rho_rule not in ("min", "max"):
            raise ValueError("beamformer must be 'ZC', 'fft' or 'ps'; rho_rule 'min' or 'max'")
        self.beamformer, self.rho_rule, self.rho_scale = beamformer, rho_rule, float(rho_scale)
        self._T_prop = T_prop

    def replace(self, **kw):
        """A copy with some parameters changed (``Gt`` follows ``Nt`` unless given, as the drivers keep Gt = Nt)."""
        cur = dict(Nt=self.Nt, Nr=self.Nr, L=self.L, T=self.T, Mr=self.Mr, Mr_e=self.Mr_e, Gr=self.Gr, Gt=self.Gt,
                   clusters=self.clusters, rays=self.rays, snr_db=self.snr_db, beamformer=self.beamformer,
                   rho_rule=self.rho_rule, rho_scale=self.rho_scale, T_prop=self._T_prop)
        if "Nt" in kw and "Gt" not in kw:
            cur["Gt"] = kw["Nt"]
        cur.update(kw)
        return SweepParams(**cur)

    @property
    def T_prop(self):                       # plot_errorVSsnr.m:23
        return self.T * self.Nt if self._T_prop is None else self._T_prop

    @property
    def noise_var(self):                    # :49
        return 10.0 ** (-self.snr_db / 10.0)

    @property
    def T_hbf(self):                        # plot_errorVSsnr.m:22 — MATLAB round() is half away from zero
        x = self.T / (self.Nr / self.Mr)
        return int(math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1)) * self.Nt

    @property
    def solver_shape(self):
        """(N, M, Gr, G2) of the proposed_algorithm call."""
        return self.Mr_e, self.T_prop, self.Gr, self.L * self.Gt


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
    from .solvers import colmajor
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


class TrainingParams:
    """Parameters of the Alg.1-vs-Alg.2 driver — names follow plot_errorVSsnr_approx.m:8-20
    (``Gr = Nr``, ``Gt = Nt``; the frame is T columns, not T*Nt)."""

    def __init__(self, Nt=4, Nr=32, L=4, T=70, ratio=0.75, clusters=2, rays=3, snr_db=5.0):
        self.Nt, self.Nr, self.L, self.T, self.ratio = Nt, Nr, L, T, float(ratio)
        self.Gr, self.Gt = Nr, Nt
        self.clusters, self.rays = clusters, rays
        self.snr_db = float(snr_db)

    @property
    def Lr(self):                           # wideband_hybBF_comm_system_training.m:5 (MATLAB round)
        return int(math.floor(self.ratio * self.Nr + 0.5))

    @property
    def noise_var(self):                    # plot_errorVSsnr_approx.m:35
        return 10.0 ** (-self.snr_db / 10.0)

    @property
    def solver_shape(self):
        return self.Nr, self.T, self.Gr, self.L * self.Gt


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
    from .solvers import colmajor
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


def build_trials_training(p: TrainingParams, trial0, batch, *, seed=20190913, sweep_idx=0, device=None, want_draws=False,
                          want_H=False, ctx=None):
    """wideband_hybBF_comm_system_training.m:1-58 + plot_errorVSsnr_approx.m:45-58 for trials [trial0, trial0 + batch) on
    the HIP path: the same library call as ``build_trials`` with the model fields that make it that builder — Gaussian
    Hermitian-Toeplitz pilots (:19-22), the unitary DFT combiner over all Nr outputs (:10), ``Lr = round(ratio*Nr)``
    of them kept per column (:5,:48-53), the frame T itself, ``rho = sqrt(lambda_6 (tau_X + tau_S)/2)`` (:51-53).
    Same dict as ``build_inputs_training`` (tau_X, tau_S, rho; Zbar complex64, column-major)."""
    q = SweepParams(p.Nt, p.Nr, p.L, p.T, p.Lr, Mr_e=p.Nr, Gr=p.Gr, Gt=p.Gt, clusters=p.clusters, rays=p.rays,
                    snr_db=p.snr_db, beamformer="fft", rho_rule="min", rho_scale=math.sqrt(0.75), T_prop=p.T)
    o = build_trials(q, trial0, batch, seed=seed, sweep_idx=sweep_idx, device=device, want_draws=want_draws, want_H=want_H,
                     ctx=ctx, pilots="gauss")
    o["tau_X"] = o.pop("tau_Y")                                                 # plot_errorVSsnr_approx.m:50
    o["tau_S"] = o["tau_X"] / 2.0                                               # :51
    del o["tau_Z"]
    return o


def build_trials(p: SweepParams, trial0, batch, *, seed=20190913, sweep_idx=0, device=None, with_hbf=False,
                 want_draws=False, want_H=False, shared_pilots=False, ctx=None, pilots="qam4"):
    """plot_errorVSsnr.m:57-136 for trials [trial0, trial0 + batch) on the HIP path
    (``jstsp_build_trials_c32``, csrc/inputgen.hip): draws, channel, pilots, measurement, A, B,
    hyper-parameters and indx_S are produced by the library's own kernels — nothing but the output
    allocation goes through torch.  Same dict as ``build_inputs`` (Zbar complex64, column-major);
    ``shared_pilots``: one pilot set for the whole sweep point (``B`` identical for every trial: pass ``B[0]``).
    ``want_draws`` adds the raw draws (gains, u_r, u_t, noise, qam_idx) for checking against a CPU
    restatement.  The Philox streams are keyed by (seed, sweep_idx, global trial index).
    """
    import ctypes as C
    import numpy as np
    from . import _lib
    from .solvers import empty_colmajor
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    c = ctx if ctx is not None else _lib.default_context(device.index or 0)
    c.use_torch_stream()
    N, M, Gr, G2 = p.solver_shape
    Np = p.clusters * p.rays
    Th = p.T_hbf if with_hbf else 0
    model = _lib.Model(p.Nt, p.Nr, p.L, p.T_prop, p.Mr, p.Mr_e, p.Gr, p.Gt, p.clusters, p.rays, Th,
                       1 if shared_pilots else 0, p.noise_var, _lib.BF_ZC if p.beamformer == "ZC" else _lib.BF_DFT,
                       _lib.RHO_MAX if p.rho_rule == "max" else _lib.RHO_MIN6, p.rho_scale,
                       _lib.PILOTS_GAUSS if pilots == "gauss" else _lib.PILOTS_QAM4)
    c64, f32 = torch.complex64, torch.float32
    out = dict(subY=empty_colmajor(batch, N, M, c64, device), Omega=empty_colmajor(batch, N, M, f32, device),
               A=empty_colmajor(1, N, Gr, c64, device)[0], B=empty_colmajor(batch, G2, M, c64, device),
               Zbar=empty_colmajor(batch, Gr, G2, c64, device),
               indx_S=torch.empty((batch, Gr * G2), dtype=torch.int32, device=device))
    if want_H:
        out["H"] = empty_colmajor(batch, p.Nr, p.Nt * p.L, c64, device)
    if with_hbf:
        out["Y_hbf"] = empty_colmajor(batch, p.Nr, Th, c64, device)
        out["A_hbf"] = empty_colmajor(1, p.Nr, Gr, c64, device)[0]
        out["B_hbf"] = empty_colmajor(batch, G2, Th, c64, device)
    if want_draws:
        out["gains"] = torch.empty((batch, p.L, Np), dtype=c64, device=device)
        out["u_r"] = torch.empty((batch, Np), dtype=f32, device=device)
        out["u_t"] = torch.empty((batch, Np), dtype=f32, device=device)
        out["noise"] = empty_colmajor(batch, p.Nr, p.T_prop, c64, device)
        out["qam_idx"] = torch.empty((batch, p.Nt, p.T_prop), dtype=torch.uint8, device=device)
        out["pilot_sym"] = torch.empty((batch, p.Nt, p.T_prop), dtype=c64, device=device)
    hyp = {k: np.empty(batch, dtype=np.float64) for k in ("tau_Y", "tau_Z", "rho")}
    tr = _lib.Trials()
    for k, v in out.items():
        setattr(tr, k, v.data_ptr())
    for k, v in hyp.items():
        setattr(tr, k, v.ctypes.data_as(C.POINTER(C.c_double)))
    rc = c._lib.jstsp_build_trials_c32(c.handle, C.byref(model), C.c_uint64(seed), int(sweep_idx), int(trial0),
                                       int(batch), C.byref(tr), _lib.DEVICE)
    _lib.check(rc, "jstsp_build_trials_c32")
    out.update({k: torch.from_numpy(v) for k, v in hyp.items()})
    return out
