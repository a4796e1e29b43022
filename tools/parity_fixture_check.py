#!/usr/bin/env python3
"""|dNMSE| of the HIP path against the committed float64 fixture (tests/golden/fullsize_port.npz) under environment variants -
seconds per variant, no CPU solve.   python tools/parity_fixture_check.py [--group sweep_proposed] [--n 640] "NAME=V,NAME2=V" ..."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--group", default="sweep_proposed")
    ap.add_argument("--n", type=int, default=0, help="rows (0 = all), taken with a stride that covers every SNR point")
    ap.add_argument("--two-output", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--fixture", default="fullsize_port", help="fullsize_port (the set defaults are chosen on) or fullsize_port_heldout")
    ap.add_argument("variants", nargs="*", default=[""])
    a = ap.parse_args()
    import time
    from oracle.fullsize_fixture import fixture, solve_group
    fx = fixture(a.fixture)
    total = len(fx[a.group + "/nmse_port"])
    rows = np.arange(total)
    if a.n and a.n < total:                       # blocks of 64 consecutive trials spread over the group
        nb = max(1, a.n // 64)
        starts = np.linspace(0, total - 64, nb).astype(int) // 64 * 64
        rows = np.unique(np.concatenate([np.arange(s, s + 64) for s in starts]))
    res, per_trial = {}, {}
    for v in a.variants:
        env = dict(kv.split("=") for kv in v.split(",") if kv)
        os.environ.update(env)
        t0 = time.time()
        try:
            nmse, _ = solve_group(fx, a.group, rows, want_ce=not a.two_output, angles=a.group.endswith("angles"))
        finally:
            for k in env:
                os.environ.pop(k, None)
        d = nmse - fx[a.group + "/nmse_port"][rows]
        res[v or "default"] = {"trials": int(len(rows)), "max": float(np.abs(d).max()), "rms": float(np.sqrt(np.mean(d ** 2))),
                               "p99": float(np.quantile(np.abs(d), 0.99)), "over_1e-6": int(np.sum(np.abs(d) > 1e-6)),
                               "over_5e-7": int(np.sum(np.abs(d) > 5e-7)), "signed_mean": float(d.mean()), "seconds": round(time.time() - t0, 1)}
        snr = fx[a.group + "/snr_db"][rows]
        res[v or "default"]["rms_by_snr"] = {"%+d" % int(x): round(float(np.sqrt(np.mean(d[snr == x] ** 2))) * 1e7, 3) for x in np.unique(snr)}
        res[v or "default"]["mean_by_snr"] = {"%+d" % int(x): round(float(np.mean(d[snr == x])) * 1e7, 3) for x in np.unique(snr)}
        per_trial[v or "default"] = d
        print(v or "default", json.dumps(res[v or "default"]), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"group": a.group, "results": res}, f, indent=1)
        np.savez_compressed(a.out.replace(".json", "") + "_per_trial.npz", rows=rows, **{k.replace("=", "_").replace(",", "__") or "default": v for k, v in per_trial.items()})


if __name__ == "__main__":
    main()
