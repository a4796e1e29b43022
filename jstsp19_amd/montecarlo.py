"""Monte-Carlo sweep runner — the counterpart of plot_errorVSsnr.m:48-180.

Loop structure of the reference: for each sweep point (SNR) and each realisation build the
system (:57-136), call ``proposed_algorithm`` (:137) and ``proposed_algorithm_angles`` (:144),
turn S into the capped spectral NMSE (:138-141,:145-148) and average over realisations (:170).

Here the (sweep point, trial) pairs are flattened, cut into contiguous blocks, one block per
rank (one process per GPU), solved ``batch`` trials at a time, and the per-point NMSE sums are
combined with ONE all-reduce at the end (RCCL over xGMI when the process group is "nccl";
a few hundred bytes, latency-bound).  Random numbers are keyed by (seed, sweep idx, trial idx),
so the result does not depend on the number of ranks.
"""
from __future__ import annotations

import torch

from .system_model import SweepParams, TrainingParams, build_trials, build_trials_training

__all__ = ["partition", "run_sweep", "run_points", "sweep_points", "run_approx_sweep", "driver", "run_driver",
           "admmiters_points", "run_convergence_curves", "zy_points", "run_zy"]


def partition(n_items, world, rank):
    """Contiguous block [lo, hi) of rank ``rank`` when n_items are split over ``world`` ranks
    (first n_items % world ranks get one extra item)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def sweep_points(base: SweepParams, name, values):
    """Sweep points of the sibling drivers: vary one parameter of ``base`` —
    ``snr_db`` (plot_errorVSsnr.m:24,48), ``L`` (plot_errorVSdelays.m:45-51), ``T``
    (plot_errorVSframelength.m:46-51), ``Mr`` (plot_errorVSnrf.m:46), ``Nt`` (plot_errorVSnt.m:46-52),
    ``rays`` (plot_errorVSpaths.m:47-51)."""
    return [base.replace(**{name: v}) for v in values]


def driver(name):
    """The sweep of one of the reference's drivers at ITS OWN parameters: ``dict(points, Imax, numOfnz, n_trials,
    metric, axis, values)`` ready for ``run_points(d["points"], d["n_trials"], Imax=d["Imax"], numOfnz=d["numOfnz"],
    metric=d["metric"], baselines=True)``.  Each driver's construction quirks are kept: beamformer kind, the
    min/max eigenvalue in rho, the (L, T) and (Nt, T) pairs that move together.

    ================== ======================================= =====================================================
    name               sweep axis                              cite
    ================== ======================================= =====================================================
    errorVSsnr         snr_db = -15:3:15                       plot_errorVSsnr.m:8-25,124-130
    errorVSdelays      L = 2,4,6,8,10 with T = 5,10,...,25     plot_errorVSdelays.m:7-21,43-46,122,128
    errorVSframelength T = 5,15,25,35 (Nt = 8, 'fft')          plot_errorVSframelength.m:7-22,44-46,123,129
    errorVSnrf         Mr = 4,8,12,16 (T = 5)                  plot_errorVSnrf.m:7-22,44-46,122,128
    errorVSnt          Nt = 4,6,8,12,16 with T = 35,..,35,25   plot_errorVSnt.m:7-22,44-48,123,129
    errorVSpaths       rays = 1,3,6,9,12                       plot_errorVSpaths.m:7-23,45,122,128
    rateVSframelength  T = 5,10,15 (Nt = 8, 'fft'), rate       plot_rateVSframelength.m:7-22,44-46,116,122
    ================== ======================================= =====================================================
    """
    snr = lambda db: dict(snr_db=float(db))
    if name == "errorVSsnr":
        base = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4)
        axis, values, pts = "snr_db", list(range(-15, 16, 3)), None
        cfg = dict(Imax=100, numOfnz=100, n_trials=1, metric="nmse")
    elif name == "errorVSdelays":
        base = SweepParams(Nt=4, Nr=32, L=2, T=5, Mr=4, rho_rule="max", **snr(5))
        axis, values = "L", [2, 4, 6, 8, 10]
        pts = [base.replace(L=L, T=5 * (i + 1)) for i, L in enumerate(values)]
        cfg = dict(Imax=100, numOfnz=50, n_trials=10, metric="nmse")
    elif name == "errorVSframelength":
        base = SweepParams(Nt=8, Nr=32, L=4, T=5, Mr=4, beamformer="fft", **snr(15))
        axis, values, pts = "T", [5, 15, 25, 35], None
        cfg = dict(Imax=100, numOfnz=50, n_trials=1, metric="nmse")
    elif name == "errorVSnrf":
        base = SweepParams(Nt=4, Nr=32, L=4, T=5, Mr=4, rho_rule="max", **snr(5))
        axis, values, pts = "Mr", [4, 8, 12, 16], None
        cfg = dict(Imax=100, numOfnz=100, n_trials=50, metric="nmse")
    elif name == "errorVSnt":
        base = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4, beamformer="fft", rho_rule="max", **snr(15))
        axis, values = "Nt", [4, 6, 8, 12, 16]
        pts = [base.replace(Nt=nt, T=t) for nt, t in zip(values, [35, 35, 35, 35, 25])]
        cfg = dict(Imax=100, numOfnz=50, n_trials=50, metric="nmse")
    elif name == "errorVSpaths":
        base = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4, rho_rule="max", **snr(5))
        axis, values, pts = "rays", [1, 3, 6, 9, 12], None
        cfg = dict(Imax=100, numOfnz=100, n_trials=1, metric="nmse")
    elif name == "rateVSframelength":
        base = SweepParams(Nt=8, Nr=32, L=4, T=5, Mr=4, beamformer="fft", **snr(15))
        axis, values, pts = "T", [5, 10, 15], None
        cfg = dict(Imax=100, numOfnz=50, n_trials=1, metric="rate")
    else:
        raise ValueError("unknown driver %r" % (name,))
    cfg.update(points=pts if pts is not None else sweep_points(base, axis, values), axis=axis, values=values)
    return cfg


def run_driver(name, n_trials=None, **kw):
    """``run_points`` on the preset of ``driver(name)`` (all seven columns' worth of baselines unless overridden);
    ``n_trials`` defaults to the driver's own maxMCRealizations."""
    d = driver(name)
    kw.setdefault("baselines", True)
    return run_points(d["points"], d["n_trials"] if n_trials is None else n_trials, Imax=d["Imax"], numOfnz=d["numOfnz"],
                      metric=d["metric"], **kw)


def _score(S, zb, metric, noise_var):
    """Per-trial figure of merit: capped spectral NMSE (plot_errorVSsnr.m:138-141) or the rate of
    plot_rateVSframelength.m:81,113,130,135."""
    from . import solvers as J
    return J.nmse_spectral(S, zb) if metric == "nmse" else J.rate(S, zb, noise_var)


def _eye(n, like):
    from .solvers import colmajor
    return colmajor(torch.eye(n, dtype=torch.complex64, device=like.device))


def _times_h(Y, B):
    """``Y*B'`` (plot_errorVSsnr.m:80; also ``B*B'`` of :79 with Y = B) on the library's correlation kernel:
    ``jstsp_correlate_c32`` computes ``A'*K*B'``, here with A = I."""
    from . import solvers as J
    return J.correlate(Y, _eye(Y.shape[-2], Y), B)


def _times(A, S, P):
    """``A*S*P`` (``Y*pinv(B)`` of plot_errorVSsnr.m:117 with A = I; ``A*S_ls``; ``A'*Y*pinv(B)`` of plot_errorVSzy.m:73) on the
    library's synthesis kernel (``jstsp_synthesize_c32``).  ``A`` None = identity."""
    from . import solvers as J
    return J.synthesize(S, _eye(S.shape[-2], S) if A is None else A, P)


def _hip_baselines(inp, numOfnz, metric="nmse", noise_var=1.0, tssr=None, vamp_max_order=128):
    """LS, VAMP and MMV-OMP baselines of plot_errorVSsnr.m:73-121 on the conventional-HBF measurement; with
    ``tssr = (Imax, rho)`` also the commented TSSR recipe (:151,158-162) on the proposed scheme's measurement.
    Every product goes through the library (correlate / synthesize entry points), nothing through torch matmuls.

    LS of a factor too large for the float64 pinv kernel takes the fp32 Gram-inverse route, whose accuracy is
    ``6e-8 * cond(B B')``; the library records the conditioning on the device (``jstsp_last_conditioning``).  Where that
    record says the digits are not there (``lambda_min/lambda_max < 1e-6`` or a Newton-Schulz residual >= 1e-2) the LS
    column - and the MMV-OMP column when it had to be built from that LS estimate - is NaN instead of a wrong number."""
    from . import _lib
    from . import solvers as J
    zb = J.colmajor(inp["Zbar"].to(torch.complex64))
    ctx = _lib.default_context(inp["Y_hbf"].device.index or 0)
    nan = lambda: torch.full((inp["Y_hbf"].shape[0],), float("nan"), dtype=torch.float64)
    Bh = inp["B_hbf"]
    G2 = Bh.shape[1]
    try:
        PB = J.pinv(Bh)                                                                  # pinv(B)  :83, :117
    except J.JstspError as e:
        if e.code != _lib.E_UNSUPPORTED:                                                 # (too large for the in-LDS kernel)
            raise
        PB = None
    S_ls = J.ls_estimate(inp["Y_hbf"], inp["A_hbf"], Bh)                                 # :83
    ls_ok = True
    if PB is None:                                                                       # the Gram-inverse route ran for B
        rcond, ns_res = ctx.last_conditioning()
        ls_ok = rcond * rcond >= 1e-6 and ns_res < 1e-2
    out = {"ls": _score(S_ls, zb, metric, noise_var) if ls_ok else nan()}
    if G2 <= vamp_max_order and inp["A_hbf"].shape[0] <= 128:       # (orders above 128: one block-Jacobi decomposition of that order per trial)
        Gb = _times_h(Bh, Bh)                                                            # (B*B')  :79
        Ym = _times_h(inp["Y_hbf"], Bh)                                                  # Y_hbf*B' :80
        out["vamp"] = _score(J.vamp_kron(Ym, inp["A_hbf"], Gb, 1.0, numOfnz), zb, metric, noise_var)   # :100
    Ypb = None
    if PB is not None:
        Ypb = _times(None, inp["Y_hbf"], PB)                                             # Y_hbf_nr*pinv(B)  :117
    elif inp["A_hbf"].shape[0] == inp["A_hbf"].shape[1] and ls_ok:
        # B too large for the pinv kernel, A square: Y*pinv(B) = A*(pinv(A)*Y*pinv(B)) = A*S_ls (A invertible: a unitary
        # dictionary times the beamformer), with the Gram-inverse route of jstsp_ls_c32 behind S_ls
        Ypb = _times(inp["A_hbf"], S_ls, _eye(G2, S_ls))
    if Ypb is not None:
        Z, _, _ = J.mmv_omp(inp["A_hbf"], Ypb, numOfnz)                                  # :116-117
        out["omp_mmv"] = _score(Z, zb, metric, noise_var)
    elif not ls_ok:
        out["omp_mmv"] = nan()
    if tssr is not None:
        try:
            St, _, Ssvt = J.tssr(inp["subY"], inp["Omega"], inp["A"], inp["B"], tssr[0], inp["tau_Y"].numpy(), tssr[1],
                                 2 * numOfnz)                                            # :151,:160-161
            out["tssr"] = _score(St, zb, metric, noise_var)
            out["svt"] = _score(Ssvt, zb, metric, noise_var)                             # :152-153
        except J.JstspError as e:
            if e.code != _lib.E_UNSUPPORTED:
                raise
    return out


def _hip_solvers(device, metric="nmse"):
    """Default solver pair: the HIP path.  Raises if the library / GPU is missing."""
    from . import solvers as J

    def solve(inp, Imax, noise_var=1.0):
        S, _, _ = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], Imax, inp["tau_Y"].numpy(),
                                       inp["tau_Z"].numpy(), inp["rho"].numpy(), "approximate", want_ce=False)
        Sa, _, _ = J.proposed_algorithm_angles(inp["subY"], inp["Omega"], inp["indx_S"], inp["A"], inp["B"], Imax,
                                               inp["tau_Y"].numpy(), inp["tau_Z"].numpy(), inp["rho"].numpy(),
                                               "approximate", None, want_ce=False)
        zb = J.colmajor(inp["Zbar"].to(torch.complex64))
        return _score(S, zb, metric, noise_var), _score(Sa, zb, metric, noise_var)

    return solve


def _resolve_builder(builder, device=None):
    """None / "hip": the library's own input kernels; otherwise a callable hook (see ``run_points``)."""
    if builder is None or builder == "hip":
        if device is not None and torch.device(device).type != "cuda":
            raise ValueError("builder=None / 'hip' builds the inputs with the library's kernels and needs a CUDA (ROCm) device; on "
                             "device=%r pass a callable builder (the CPU-side tests use tests/torch_builder.py: builder)" % (device,))
        return "hip"
    if not callable(builder):
        raise ValueError("builder must be None, 'hip' or a callable (p, trial_ids, seed, sweep_idx, device, with_hbf) -> inputs "
                         "(the torch tensor-op builder of rounds 1-3 is tests/torch_builder.py: builder)")
    return builder


def _merge_key(p):
    """Sweep points whose trials may share one solver call: same array shapes and the same trial-independent A (the
    SNR, the number of paths and the rho rule only enter the per-trial arrays, which are built per point)."""
    return (p.Nt, p.Nr, p.L, p.T, p.Mr, p.Mr_e, p.Gr, p.Gt, p.beamformer, p.T_prop)


def _merge_inputs(inps):
    """Concatenate the per-trial arrays of several ``build_trials`` results (column-major layout kept)."""
    if len(inps) == 1:
        return inps[0]
    out = {}
    for k, v0 in inps[0].items():
        if k in ("A", "A_hbf"):                          # trial-independent: beamformer x dictionary
            out[k] = v0
        elif v0.is_cuda and v0.ndim == 3:
            out[k] = torch.cat([x[k].transpose(1, 2) for x in inps], 0).transpose(1, 2)
        else:
            out[k] = torch.cat([x[k] for x in inps], 0)
    return out


def _merge_cap(p, batch, with_hbf):
    """Trials per merged solver call: small problems are latency-bound per launch (one workgroup per matrix in the
    eigen-decompositions), so trials of several sweep points go into one call - up to about 1.5 GB of inputs."""
    N, M, Gr, G2 = p.solver_shape
    per_trial = 8 * (G2 * M + 2 * N * M + 2 * Gr * G2 + ((G2 + p.Nr) * p.T_hbf if with_hbf else 0))
    return max(batch, min(4096, int(1.5e9 // max(per_trial, 1))))


def run_points(points, n_trials, *, Imax=100, batch=64, seed=20190913, device=None, solve_fn=None, dist=None,
               baselines=False, numOfnz=100, builder=None, metric="nmse", tssr=None, merge=True, vamp_max_order=128,
               samples=None):
    """Mean capped NMSE per sweep point; columns (proposed_algorithm, proposed_algorithm_angles[, LS, VAMP, MMV-OMP
    [, TSSR]]).  ``metric="rate"``: the rate of plot_rateVSframelength.m:81 instead of the NMSE (HIP solvers only).
    ``tssr=(Imax_svt, rho_svt)`` adds the commented recipes of plot_errorVSsnr.m:151-162 as columns six and seven: TSSR
    (``mc_svt`` then joint OMP) and "SVT-based" (``pinv(A)*Y_svt*pinv(B)``).

    ``solve_fn(inputs, Imax) -> (nmse, nmse_angles)`` (two tensors of per-trial NMSE) defaults to the HIP path.
    ``baselines=True`` adds the LS and VAMP columns of plot_errorVSsnr.m:83-105 (HIP path only; VAMP is NaN
    where the delay factor's order L*Gt exceeds ``vamp_max_order`` - 128 by default: above that every trial costs one
    block-Jacobi eigen-decomposition of that order, csrc/eig_large.hip).  ``dist``: ``torch.distributed`` (initialised) or None.
    ``builder``: None / "hip" — inputs from the library's own kernels (``jstsp_build_trials_c32``) — or a callable
    ``builder(p, trial_ids, seed, sweep_idx, device, with_hbf) -> inputs`` (the CPU-side tests pass the torch tensor-op
    builder of tests/torch_builder.py, which runs without a GPU; it has its own random streams, so its curves agree with
    the library builder's statistically, not sample by sample).
    ``samples``: a list that receives, per sweep point, a float64 tensor (trials of THIS rank, ncol) of the per-trial values
    before averaging (what plot_errorVSsnr.m:138-141 computes per realisation) - for distributional checks.
    Returns a float64 tensor (len(points), ncol) identical on every rank.
    """
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    builder = _resolve_builder(builder, device)
    if metric not in ("nmse", "rate"):
        raise ValueError("metric must be 'nmse' or 'rate'")
    custom = solve_fn is not None
    if custom and metric != "nmse":
        raise ValueError("metric='rate' needs the HIP solvers (solve_fn=None)")
    if solve_fn is None:
        solve_fn = _hip_solvers(device, metric)
    n_pts = len(points)
    ncol = (7 if tssr is not None else 5) if baselines else 2
    lo, hi = partition(n_pts * n_trials, world, rank)
    acc = torch.zeros((n_pts, ncol + 1), dtype=torch.float64)   # sums per column, then the trial count
    # HIP path, NMSE metric: chunks of consecutive sweep points with equal shapes are solved in ONE call (the per-trial
    # results do not depend on what else is in the batch; a rate sweep keeps one noise variance per call)
    merging = merge and not custom and builder == "hip" and metric == "nmse"
    item = lo
    while item < hi:
        chunks, inps, total = [], [], 0
        while item < hi:
            pt = item // n_trials
            t0 = item % n_trials
            t1 = min(n_trials, t0 + batch, t0 + (hi - item))
            p = points[pt]
            if chunks and (not merging or _merge_key(p) != _merge_key(points[chunks[0][0]])
                           or total + (t1 - t0) > _merge_cap(p, batch, baselines)):
                break
            if builder == "hip":
                inps.append(build_trials(p, t0, t1 - t0, seed=seed, sweep_idx=pt, device=device, with_hbf=baselines))
            else:
                inps.append(builder(p, range(t0, t1), seed, pt, device, baselines))
            chunks.append((pt, t1 - t0))
            total += t1 - t0
            item += t1 - t0
        inp = _merge_inputs(inps)
        del inps
        p = points[chunks[0][0]]
        e, ea = solve_fn(inp, Imax) if custom else solve_fn(inp, Imax, p.noise_var)
        cols = [torch.as_tensor(e).double().cpu(), torch.as_tensor(ea).double().cpu()]
        if baselines:
            b = _hip_baselines(inp, numOfnz, metric, p.noise_var, tssr, vamp_max_order)
            for key in ("ls", "vamp", "omp_mmv", "tssr", "svt")[:ncol - 2]:
                cols.append(b[key].double().cpu() if key in b else torch.full((total,), float("nan"), dtype=torch.float64))
        o = 0
        for pt, cnt in chunks:
            for col, v in enumerate(cols):
                acc[pt, col] += float(v[o:o + cnt].sum())
            acc[pt, ncol] += cnt
            if samples is not None:
                while len(samples) < n_pts:
                    samples.append(torch.zeros((0, ncol), dtype=torch.float64))
                samples[pt] = torch.cat([samples[pt], torch.stack([v[o:o + cnt] for v in cols], dim=1)], 0)
            o += cnt
    if dist is not None:
        buf = acc.to(device) if dist.get_backend() == "nccl" else acc
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)       # the single collective of the sweep
        acc = buf.cpu()
    return acc[:, :ncol] / acc[:, ncol:ncol + 1]         # plot_errorVSsnr.m:170-171


def run_sweep(base: SweepParams, snr_db_list, n_trials, **kw):
    """The SNR sweep of plot_errorVSsnr.m:48-180 (see ``run_points``)."""
    return run_points(sweep_points(base, "snr_db", [float(s) for s in snr_db_list]), n_trials, **kw)


def _hip_alg12(inp, Imax):
    """plot_errorVSsnr_approx.m:60-72 on the HIP path: Alg.1 ('std') and Alg.2 ('approximate'), each scored
    through its completed measurement, ``S = pinv(A)*Y*pinv(B)``."""
    from . import solvers as J
    zb = J.colmajor(inp["Zbar"].to(torch.complex64))
    out = []
    for kind in ("std", "approximate"):
        _, Y, _ = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], Imax, inp["tau_X"].numpy(),
                                       inp["tau_S"].numpy(), inp["rho"].numpy(), kind, want_ce=False)   # :60,:67
        out.append(J.nmse_spectral(J.ls_estimate(Y, inp["A"], inp["B"]), zb))                            # :61-65
    return out


def run_approx_sweep(base: TrainingParams, snr_db_list, Imax_list, n_trials, *, batch=64, seed=20190913, device=None,
                     solve_fn=None, dist=None, builder=None):
    """The Alg.1-vs-Alg.2 sweep of plot_errorVSsnr_approx.m:34-85: for each SNR and each Imax, ``n_trials`` fresh
    realisations of wideband_hybBF_comm_system_training, both solver variants, capped NMSE of
    ``pinv(A)*Y*pinv(B)``, mean then ``min(., 1)`` (:76-77).

    Returns a float64 tensor (len(Imax_list), len(snr_db_list), 2) — ``[..., 0]`` is mean_error_proposed,
    ``[..., 1]`` mean_error_proposed_approx — identical on every rank.  ``solve_fn(inputs, Imax) -> (e_std, e_approx)``
    defaults to the HIP path; the (sweep point, trial) pairs are sharded over ranks as in ``run_points``.
    ``builder``: None / "hip" - the inputs from the library's own kernels (``jstsp_build_trials_c32`` with the training model
    fields) - or a callable hook as in ``run_points``.
    """
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    builder = _resolve_builder(builder, device)
    if solve_fn is None:
        solve_fn = _hip_alg12
    pts = [(si, ii) for si in range(len(snr_db_list)) for ii in range(len(Imax_list))]    # loop order of :34-38
    lo, hi = partition(len(pts) * n_trials, world, rank)
    acc = torch.zeros((len(pts), 3), dtype=torch.float64)
    item = lo
    while item < hi:
        pt = item // n_trials
        t0 = item % n_trials
        t1 = min(n_trials, t0 + batch, t0 + (hi - item))
        si, ii = pts[pt]
        p = TrainingParams(base.Nt, base.Nr, base.L, base.T, base.ratio, base.clusters, base.rays, float(snr_db_list[si]))
        if builder == "hip":
            inp = build_trials_training(p, t0, t1 - t0, seed=seed, sweep_idx=pt, device=device)
        else:
            inp = builder(p, range(t0, t1), seed, pt, device, False)
        e1, e2 = solve_fn(inp, int(Imax_list[ii]))
        acc[pt, 0] += float(torch.as_tensor(e1).double().sum())
        acc[pt, 1] += float(torch.as_tensor(e2).double().sum())
        acc[pt, 2] += t1 - t0
        item += t1 - t0
    if dist is not None:
        buf = acc.to(device) if dist.get_backend() == "nccl" else acc
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        acc = buf.cpu()
    mean = torch.clamp(acc[:, :2] / acc[:, 2:3], max=1.0)                                  # :76-77
    return mean.reshape(len(snr_db_list), len(Imax_list), 2).transpose(0, 1).contiguous()


def _generic_sharded(points, n_trials, width, work, *, batch, seed, device, dist, builder):
    """Shared skeleton of the curve-type drivers: shard the (point, trial) pairs over ranks, build ``batch`` trials at
    a time, add ``work(inputs, point) -> (batch, width)`` per-trial rows, one all-reduce, mean per point."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    lo, hi = partition(len(points) * n_trials, world, rank)
    acc = torch.zeros((len(points), width + 1), dtype=torch.float64)
    item = lo
    while item < hi:
        pt = item // n_trials
        t0 = item % n_trials
        t1 = min(n_trials, t0 + batch, t0 + (hi - item))
        p = points[pt]
        if builder == "hip":
            inp = build_trials(p, t0, t1 - t0, seed=seed, sweep_idx=pt, device=device)
        else:
            inp = builder(p, range(t0, t1), seed, pt, device, False)
        rows = torch.as_tensor(work(inp, p)).double().reshape(t1 - t0, width).cpu()
        acc[pt, :width] += rows.sum(dim=0)
        acc[pt, width] += t1 - t0
        item += t1 - t0
    if dist is not None:
        buf = acc.to(device) if dist.get_backend() == "nccl" else acc
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        acc = buf.cpu()
    return acc[:, :width] / acc[:, width:width + 1]


def admmiters_points():
    """The four panels of plot_errorVSadmmiters.m (:10-24, :76-90, :141-154, :206-219): (Nt, frame, SNR) =
    (4, 40, 15 dB), (16, 160, 15 dB), (16, 160, 5 dB), (16, 480, 5 dB); Mr = 16, 'ps' combiner, the frame is T
    itself (:21), 20 realisations, Imax = 100."""
    mk = lambda Nt, T, db: SweepParams(Nt=Nt, Nr=32, L=4, T=T, Mr=16, snr_db=float(db), beamformer="ps", T_prop=T)
    return [mk(4, 40, 15), mk(16, 160, 15), mk(16, 160, 5), mk(16, 480, 5)]


def run_convergence_curves(points, n_trials=20, *, Imax=100, batch=20, seed=20190913, device=None, solve_fn=None,
                           dist=None, builder=None):
    """plot_errorVSadmmiters.m:32-71: the mean over realisations of ``convergence_error`` (Imax x 3) of
    ``proposed_algorithm`` (:61) and of ``proposed_algorithm_angles`` (:64) per panel.

    Returns float64 (len(points), 2, Imax, 3): ``[:, 0]`` = mean_error_k, ``[:, 1]`` = mean_error_angles_k.
    ``solve_fn(inputs, Imax) -> (ce, ce_angles)`` (each (batch, Imax, 3)) defaults to the HIP path.
    """
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    builder = _resolve_builder(builder, device)

    def hip(inp, Imax_):
        from . import solvers as J
        args = (Imax_, inp["tau_Y"].numpy(), inp["tau_Z"].numpy(), inp["rho"].numpy(), "approximate")
        _, _, ce = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], *args)
        _, _, cea = J.proposed_algorithm_angles(inp["subY"], inp["Omega"], inp["indx_S"], inp["A"], inp["B"], *args)
        return ce, cea

    fn = hip if solve_fn is None else solve_fn

    def work(inp, p):
        ce, cea = fn(inp, Imax)
        return torch.stack([torch.as_tensor(ce).double().cpu(), torch.as_tensor(cea).double().cpu()], dim=1)

    out = _generic_sharded(points, n_trials, 2 * Imax * 3, work, batch=batch, seed=seed, device=device, dist=dist,
                           builder=builder)
    return out.reshape(len(points), 2, Imax, 3)


def zy_points(F_range=(5,)):
    """plot_errorVSzy.m:7-22,30: Nt = 16, Nr = 32, Mr = 16, 6 rays, frame = 16*F columns, 15 dB, 'ps', rho halved."""
    return [SweepParams(Nt=16, Nr=32, L=4, T=16 * F, Mr=16, rays=6, snr_db=15.0, beamformer="ps", rho_scale=0.5,
                        T_prop=16 * F) for F in F_range]


def run_zy(points=None, n_trials=1, *, Imax=50, batch=32, seed=20190913, device=None, solve_fn=None, dist=None,
           builder=None):
    """plot_errorVSzy.m:28-84: per frame length the mean capped NMSE of the solver's ``S`` (the "Z" curve, :67-71)
    and of ``A'*Y*pinv(B)`` built from its completed measurement (the "Y" curve, :73-77).
    Returns float64 (len(points), 2).  ``solve_fn(inputs, Imax) -> (e_z, e_y)`` defaults to the HIP path."""
    if points is None:
        points = zy_points()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    builder = _resolve_builder(builder, device)

    def hip(inp, Imax_):
        from . import solvers as J
        S, Y, _ = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], Imax_, inp["tau_Y"].numpy(),
                                       inp["tau_Z"].numpy(), inp["rho"].numpy(), "approximate", want_ce=False)   # :66
        zb = J.colmajor(inp["Zbar"].to(torch.complex64))
        Sy = _times(J.colmajor(inp["A"].conj().transpose(-1, -2).contiguous()), Y, J.pinv(inp["B"]))             # :73
        return J.nmse_spectral(S, zb), J.nmse_spectral(Sy, zb)

    fn = hip if solve_fn is None else solve_fn

    def work(inp, p):
        ez, ey = fn(inp, Imax)
        return torch.stack([torch.as_tensor(ez).double().cpu(), torch.as_tensor(ey).double().cpu()], dim=1)

    return _generic_sharded(points, n_trials, 2, work, batch=batch, seed=seed, device=device, dist=dist, builder=builder)
