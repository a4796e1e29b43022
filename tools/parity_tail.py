#!/usr/bin/env python3
"""Tail of |dNMSE| (HIP path vs the float64 C++ restatement oracle/cpu_port.cpp) over THOUSANDS of full-size trials.

BASELINE configs[1] shape (N=64, M=4096, Gr=64, G2=512), the trials of the configs[3] sweep: SNR points -15:3:12 dB
(sweep index = point index, as montecarlo.run_sweep keys them) x `--trials` realisations, plus the 256 trials of the
bench workload (5 dB, sweep index 0).  Per trial: NMSE of the HIP result (default path and a few switch settings that
isolate one approximation each), NMSE of the float64 port on the same inputs, convergence_error of both.

The float64 side is the expensive one (5-6 core-seconds per trial) and depends only on the inputs, which the library's
counter-based generator reproduces bit for bit on every box: the port's per-trial NMSE / convergence_error and a
fingerprint of the inputs go to `--fixture` (npz, a few MB) so that tests/test_gpu_parity_tail.py can check thousands of
trials against float64 in seconds.  Results are written chunk by chunk (a call cut short keeps what it has).

    python tools/parity_tail.py --trials 256 --out gpurun_out/parity_tail
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {                     # name -> (environment, want_ce)
    "default": ({}, True),
    "two_output": ({}, False),
    "toeplitz1": ({"JSTSP_TOEPLITZ": "1"}, True),        # old pass kernel on the compact image
    "toeplitz0": ({"JSTSP_TOEPLITZ": "0"}, True),        # unstructured path (full tile image)
    "rv_refresh1": ({"JSTSP_RV_REFRESH": "1"}, True),    # R v recomputed every iteration (experiments build only since round 6:
                                                         # JSTSP_EXPERIMENTS_LIB=1, else this variant IS the default)
    "unfused": ({"JSTSP_FUSED": "0"}, True),             # three-kernel iteration
    "fp32_mfma": ({"JSTSP_H2": "0"}, True),              # strict complex-fp32 MFMA path
}


def fingerprint(inp):
    """A few float64 numbers that pin a trial's inputs (sum of subY, of B, of Omega; tau_Y, tau_Z, rho)."""
    f = torch.stack([inp["subY"].abs().double().sum((1, 2)), inp["B"].abs().double().sum((1, 2)),
                     inp["Omega"].double().sum((1, 2))], 1).cpu().numpy()
    return np.concatenate([f, np.stack([inp[k].numpy() for k in ("tau_Y", "tau_Z", "rho")], 1)], 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=256, help="realisations per SNR point")
    ap.add_argument("--snrs", type=str, default="-15,-12,-9,-6,-3,0,3,6,9,12")
    ap.add_argument("--bench-trials", type=int, default=256, help="trials of the bench workload (5 dB, sweep index 0)")
    ap.add_argument("--angles-trials", type=int, default=64, help="proposed_algorithm_angles trials per point of --angles-snrs")
    ap.add_argument("--angles-snrs", type=str, default="-15,0,12", help="SNR points (members of --snrs) of the _angles trials")
    ap.add_argument("--seed", type=int, default=20190913, help="generator seed (the held-out fixture uses another one)")
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--variants", type=str, default=",".join(VARIANTS))
    ap.add_argument("--out", type=str, default=os.path.join(ROOT, "gpurun_out", "parity_tail"))
    ap.add_argument("--budget-s", type=float, default=1e9, help="stop starting new chunks after this many seconds")
    a = ap.parse_args()

    import jstsp19_amd as J
    from jstsp19_amd.system_model import SweepParams, build_trials
    from oracle import build_cpu_port as bp
    from oracle import solvers as O

    os.makedirs(a.out, exist_ok=True)
    dev = torch.device("cuda", 0)
    nthr = a.threads
    if not nthr:
        nthr = max(1, (os.cpu_count() or 2) // 2)
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                nthr = min(nthr, max(1, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
    import tempfile
    try:
        lib = bp.load(bp.build(native=True, out=os.path.join(tempfile.mkdtemp(prefix="jstsp_cpu_"), "libjstsp_cpu_port.so")))
    except (RuntimeError, OSError):
        lib = bp.load()
    variants = [v for v in a.variants.split(",") if v]
    snrs = [float(s) for s in a.snrs.split(",")]
    IMAX = 100
    # work list: (tag, solver, snr_db, sweep_idx, trial0, count)
    work = []
    for t0 in range(0, a.bench_trials, a.chunk):
        work.append(("bench", "proposed", 5.0, 0, t0, min(a.chunk, a.bench_trials - t0)))
    for t0 in range(0, a.angles_trials, a.chunk):
        for s in [float(x) for x in a.angles_snrs.split(",") if x]:
            if s in snrs:
                work.append(("sweep", "angles", s, snrs.index(s), t0, min(a.chunk, a.angles_trials - t0)))
    for t0 in range(0, a.trials, a.chunk):          # trial blocks outermost: every SNR point is covered early
        for i, s in enumerate(snrs):
            work.append(("sweep", "proposed", s, i, t0, min(a.chunk, a.trials - t0)))
    t_start = time.perf_counter()
    done = 0
    for tag, solver, snr, sidx, t0, cnt in work:
        name = "%s_%s_snr%+03d_t%04d" % (tag, solver, int(snr), t0)
        path = os.path.join(a.out, name + ".npz")
        if os.path.exists(path):
            continue
        if time.perf_counter() - t_start > a.budget_s:
            break
        p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=snr)
        inp = build_trials(p, t0, cnt, seed=a.seed, sweep_idx=sidx, device=dev)
        idx = inp["indx_S"] if solver == "angles" else None
        hyp = [inp[k].numpy() for k in ("tau_Y", "tau_Z", "rho")]
        zb = inp["Zbar"].cpu().numpy().astype(np.complex128)
        rec = {"snr_db": np.full(cnt, snr), "sweep_idx": np.full(cnt, sidx), "trial": np.arange(t0, t0 + cnt),
               "fingerprint": fingerprint(inp), "seed": np.full(cnt, a.seed, dtype=np.int64)}
        for v in variants:
            env, want_ce = VARIANTS[v]
            if solver == "angles" and v not in ("default", "two_output"):
                continue
            for k, val in env.items():
                os.environ[k] = val
            try:
                S, Y, ce = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], IMAX, *hyp, "approximate",
                                                indx_S=idx, want_ce=want_ce)
                torch.cuda.synchronize()
            finally:
                for k in env:
                    os.environ.pop(k, None)
            Sh = S.cpu().numpy().astype(np.complex128)
            rec["nmse_" + v] = np.array([O.nmse_capped(Sh[t], zb[t]) for t in range(cnt)])
            rec["S_" + v] = None if v != "default" else Sh
            if want_ce:
                rec["ce_" + v] = ce.cpu().numpy().astype(np.float64)
        Sd = rec.pop("S_default")
        for k in [k for k in rec if k.startswith("S_")]:
            rec.pop(k)
        # float64 port on the same inputs
        tc = time.perf_counter()
        Sc, _, cec, used = bp.proposed_algorithm(lib, inp["subY"].cpu().numpy(), inp["Omega"].cpu().numpy(), inp["A"].cpu().numpy(),
                                                 inp["B"].cpu().numpy(), IMAX, *hyp,
                                                 indx_S=None if idx is None else idx.cpu().numpy(), want_ce=True, threads=nthr)
        rec["port_seconds"] = np.array([time.perf_counter() - tc])
        rec["port_threads"] = np.array([used])
        rec["nmse_port"] = np.array([O.nmse_capped(Sc[t], zb[t]) for t in range(cnt)])
        rec["ce_port"] = cec
        rec["rel_dS_default"] = np.array([np.max(np.abs(Sd[t] - Sc[t])) / np.max(np.abs(Sc[t])) for t in range(cnt)])
        np.savez_compressed(path, **rec)
        done += cnt
        d = np.abs(rec["nmse_default"] - rec["nmse_port"])
        print("%s: %d trials, port %.1f s on %d threads, max|dNMSE| default %.2e (mean %.2e)  [%.0f s elapsed]"
              % (name, cnt, rec["port_seconds"][0], used, d.max(), d.mean(), time.perf_counter() - t_start), flush=True)
        del inp, Sc, Sd
    summarise(a.out)


def summarise(out):
    """Merge the chunk files into summary.json (statistics per variant and group) and fixture.npz (float64 side only)."""
    files = sorted(f for f in os.listdir(out) if f.endswith(".npz") and f.split("_")[0] in ("bench", "sweep"))
    groups = {}
    for f in files:
        tag, solver = f.split("_")[:2]
        z = np.load(os.path.join(out, f))
        g = groups.setdefault(tag + "_" + solver, {})
        for k in z.files:
            g.setdefault(k, []).append(z[k])
    summ = {}
    fix = {}
    for gname, g in groups.items():
        g = {k: np.concatenate(v, 0) for k, v in g.items()}
        port = g["nmse_port"]
        s = {"trials": int(len(port)), "snr_points": sorted(set(g["snr_db"].tolist())),
             "port_core_seconds_per_trial": float(np.sum(g["port_seconds"] * g["port_threads"]) / len(port)), "variants": {}}
        for k in sorted(g):
            if not k.startswith("nmse_") or k == "nmse_port":
                continue
            d = np.abs(g[k] - port)
            order = np.argsort(-d)[:5]
            s["variants"][k[5:]] = {
                "max_abs_dNMSE": float(d.max()), "p999": float(np.quantile(d, 0.999)), "p99": float(np.quantile(d, 0.99)),
                "mean": float(d.mean()), "rms": float(np.sqrt(np.mean(d ** 2))), "over_1e-6": int(np.sum(d > 1e-6)),
                "over_5e-7": int(np.sum(d > 5e-7)), "signed_mean": float(np.mean(g[k] - port)),
                "worst": [{"snr_db": float(g["snr_db"][i]), "sweep_idx": int(g["sweep_idx"][i]), "trial": int(g["trial"][i]),
                           "dNMSE": float(g[k][i] - port[i]), "nmse_port": float(port[i])} for i in order]}
            if ("ce_" + k[5:]) in g:
                cg, cp = g["ce_" + k[5:]], g["ce_port"]
                fin = np.isfinite(cp) & (np.abs(cp) > 0)
                s["variants"][k[5:]]["max_rel_dce"] = float(np.max(np.abs(cg[fin] - cp[fin]) / np.abs(cp[fin])))
        by_snr = {}
        d0 = np.abs(g["nmse_default"] - port)
        for snr in sorted(set(g["snr_db"].tolist())):
            m = g["snr_db"] == snr
            by_snr["%+d" % int(snr)] = {"trials": int(m.sum()), "max": float(d0[m].max()), "mean": float(d0[m].mean()),
                                        "mean_nmse_port": float(port[m].mean())}
        s["default_by_snr"] = by_snr
        s["max_rel_dS_default"] = float(g["rel_dS_default"].max())
        summ[gname] = s
        for k in ("snr_db", "sweep_idx", "trial", "fingerprint", "nmse_port", "ce_port", "seed"):
            if k in g:
                fix[gname + "/" + k] = g[k] if k != "ce_port" else g[k].astype(np.float64)
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(summ, f, indent=1)
    np.savez_compressed(os.path.join(out, "fixture.npz"), **fix)
    print(json.dumps({k: {v: s["variants"][v]["max_abs_dNMSE"] for v in s["variants"]} | {"trials": s["trials"]}
                      for k, s in summ.items()}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        main()
