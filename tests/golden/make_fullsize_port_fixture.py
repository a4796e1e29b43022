#!/usr/bin/env python3
"""tests/golden/fullsize_port.npz and fullsize_port_heldout.npz from the output of tools/parity_tail.py (run on a GPU box: the
inputs are built by the library's own counter-based generator, the float64 side by oracle/cpu_port.cpp on the box's host cores).

    # the fixture the round-4 defaults were chosen on (generator seed 20190913)
    python tools/parity_tail.py --trials 256 --out gpurun_out/parity_tail          # on the GPU box (about 17 core-hours / 16)
    python tests/golden/make_fullsize_port_fixture.py gpurun_out/parity_tail        # here
    # the HELD-OUT fixture (round 5; another generator seed, never used to choose a switch)
    python tools/parity_tail.py --seed 20260105 --trials 256 --bench-trials 0 --angles-trials 128 \
           --angles-snrs=-15,-12,-9,-6,-3,0,3,6,9,12 --variants default,two_output --out gpurun_out/parity_heldout
    python tests/golden/make_fullsize_port_fixture.py gpurun_out/parity_heldout heldout
    # the second held-out fixture (round 6; seed 20261003, generated ONCE after the numerics of the round were frozen, same recipe)
    python tools/parity_tail.py --seed 20261003 --trials 256 --bench-trials 0 --angles-trials 128 \
           --angles-snrs=-15,-12,-9,-6,-3,0,3,6,9,12 --variants default,two_output --out gpurun_out/parity_heldout2
    python tests/golden/make_fullsize_port_fixture.py gpurun_out/parity_heldout2 heldout2

Per group (bench_proposed: the 256 trials of the bench workload, 5 dB, sweep index 0; sweep_proposed: 10 SNR points x 256
trials of the BASELINE configs[3] sweep; sweep_angles: proposed_algorithm_angles - 64 trials at -15 / 0 / 12 dB in the first
fixture, 128 trials at each of the 10 points in the held-out one):
snr_db, sweep_idx, trial (the generator key), seed, fingerprint (sum|subY|, sum|B|, sum Omega, tau_Y, tau_Z, rho of the inputs - a
test that rebuilds a trial checks it reproduces them), nmse_port (float64 NMSE of the float64 solve) and, for a subset,
convergence_error of the float64 solve (float32 storage: it is compared at 2e-3).  Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main(src, name=""):
    z = np.load(os.path.join(src, "fixture.npz"))
    out = {}
    heldout = name.startswith("heldout")
    for g in ("bench_proposed", "sweep_proposed", "sweep_angles"):
        if g + "/nmse_port" not in z.files:
            continue
        for k in ("snr_db", "sweep_idx", "trial", "fingerprint", "nmse_port"):
            out[g + "/" + k] = z[g + "/" + k]
        n = len(out[g + "/nmse_port"])
        out[g + "/seed"] = z[g + "/seed"] if g + "/seed" in z.files else np.full(n, 20190913, dtype=np.int64)
        ce = z[g + "/ce_port"]
        if heldout:
            keep = z[g + "/trial"] < 32
        else:
            keep = np.ones(len(ce), bool) if g != "sweep_proposed" else (z[g + "/trial"] < 16)
        out[g + "/ce_rows"] = np.nonzero(keep)[0].astype(np.int32)
        out[g + "/ce_port"] = ce[keep].astype(np.float32)
    path = os.path.join(HERE, "fullsize_port%s.npz" % ("_" + name if name else ""))
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {k[:-10]: int(len(v)) for k, v in out.items() if k.endswith("/nmse_port")})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "..", "gpurun_out", "parity_tail"),
         sys.argv[2] if len(sys.argv) > 2 else "")
