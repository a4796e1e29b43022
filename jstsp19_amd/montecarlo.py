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

from .system_model import SweepParams, build_inputs, draw_trials

__all__ = ["partition", "run_sweep"]


def partition(n_items, world, rank):
    """Contiguous block [lo, hi) of rank ``rank`` when n_items are split over ``world`` ranks
    (first n_items % world ranks get one extra item)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _hip_solvers(device):
    """Default solver pair: the HIP path.  Raises if the library / GPU is missing."""
    from . import solvers as J

    def solve(inp, Imax):
        S, _, _ = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], Imax, inp["tau_Y"].numpy(),
                                       inp["tau_Z"].numpy(), inp["rho"].numpy(), "approximate", want_ce=False)
        Sa, _, _ = J.proposed_algorithm_angles(inp["subY"], inp["Omega"], inp["indx_S"], inp["A"], inp["B"], Imax,
                                               inp["tau_Y"].numpy(), inp["tau_Z"].numpy(), inp["rho"].numpy(),
                                               "approximate", None, want_ce=False)
        zb = J.colmajor(inp["Zbar"].to(torch.complex64))
        return J.nmse_spectral(S, zb), J.nmse_spectral(Sa, zb)

    return solve


def run_sweep(base: SweepParams, snr_db_list, n_trials, *, Imax=100, batch=64, seed=20190913, device=None,
              solve_fn=None, dist=None):
    """Mean capped NMSE per sweep point for (proposed_algorithm, proposed_algorithm_angles).

    ``solve_fn(inputs, Imax) -> (nmse, nmse_angles)`` (two tensors of per-trial NMSE) defaults to
    the HIP path.  ``dist``: ``torch.distributed`` (initialised) or None for a single process.
    Returns a float64 tensor (len(snr_db_list), 2) identical on every rank.
    """
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if solve_fn is None:
        solve_fn = _hip_solvers(device)
    n_pts = len(snr_db_list)
    lo, hi = partition(n_pts * n_trials, world, rank)
    acc = torch.zeros((n_pts, 3), dtype=torch.float64)          # sum nmse, sum nmse_angles, count
    item = lo
    while item < hi:
        pt = item // n_trials
        t0 = item % n_trials
        t1 = min(n_trials, t0 + batch, t0 + (hi - item))
        p = SweepParams(base.Nt, base.Nr, base.L, base.T, base.Mr, base.Mr_e, base.Gr, base.Gt, base.clusters,
                        base.rays, snr_db=float(snr_db_list[pt]))
        draws = draw_trials(p, list(range(t0, t1)), seed=seed, sweep_idx=pt, device=device)
        inp = build_inputs(p, draws)
        e, ea = solve_fn(inp, Imax)
        acc[pt, 0] += float(torch.as_tensor(e).double().sum())
        acc[pt, 1] += float(torch.as_tensor(ea).double().sum())
        acc[pt, 2] += t1 - t0
        item += t1 - t0
    if dist is not None:
        buf = acc.to(device) if dist.get_backend() == "nccl" else acc
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)       # the single collective of the sweep
        acc = buf.cpu()
    return acc[:, :2] / acc[:, 2:3]                      # plot_errorVSsnr.m:170-171
