"""Batched, device-resident construction of the solver inputs (the caller side of the hot path).

What plot_errorVSsnr.m:57-136 does per trial on the host — channel, pilots, noise, the
random-spatial-sampling measurement, the dictionary factors A, B and the hyper-parameters —
is done for a whole batch of Monte-Carlo trials by the library's own kernels
(``jstsp_build_trials_c32``, csrc/inputgen.hip), so that the solver's inputs are born in HBM;
this module holds the parameter classes (names as in the reference's scripts) and the two thin
wrappers around that call.  Not the timed hot path.  The quirks of the
reference (tap-1 steering reuse, cumulative cluster sum, Hermitian Toeplitz pilots, sigma_6 in rho)
are reproduced by the kernels and checked against oracle/system_model.py on the library's own draws.

Random numbers come from counter-based Philox streams keyed by (seed, sweep index, global trial
index), so a trial's inputs do not depend on how trials are sharded over GPUs.

(Rounds 1-3 also kept a torch tensor-op implementation of the same construction here; it now lives in
tests/torch_builder.py, where the CPU-side tests of the sweep runner use it.)
"""
from __future__ import annotations

import math

import torch

__all__ = ["SweepParams", "TrainingParams", "build_trials", "build_trials_training"]


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


def build_trials_training(p: TrainingParams, trial0, batch, *, seed=20190913, sweep_idx=0, device=None, want_draws=False,
                          want_H=False, ctx=None):
    """wideband_hybBF_comm_system_training.m:1-58 + plot_errorVSsnr_approx.m:45-58 for trials [trial0, trial0 + batch) on
    the HIP path: the same library call as ``build_trials`` with the model fields that make it that builder — Gaussian
    Hermitian-Toeplitz pilots (:19-22), the unitary DFT combiner over all Nr outputs (:10), ``Lr = round(ratio*Nr)``
    of them kept per column (:5,:48-53), the frame T itself, ``rho = sqrt(lambda_6 (tau_X + tau_S)/2)`` (:51-53).
    Keys: subY, Omega, A, B, Zbar (complex64, column-major), indx_S, tau_X, tau_S, rho - the dict ``build_inputs_training`` of
    tests/torch_builder.py returns."""
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
    allocation goes through torch.  Keys: subY, Omega, A, B, Zbar (complex64, column-major), indx_S, tau_Y, tau_Z, rho
    (the dict ``build_inputs`` of tests/torch_builder.py returns);
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
