#!/usr/bin/env python3
"""How the HIP result leaves the float64 one over the iterations: S after Imax = k iterations, HIP vs oracle/cpu_port.cpp, for a
few full-size trials (GPU box; the port costs k/100 of a solve per point)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import jstsp19_amd as J
    from jstsp19_amd.system_model import SweepParams, build_trials
    from oracle import build_cpu_port as bp
    from oracle import solvers as O
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    variants = sys.argv[2:] or [""]
    lib = bp.load()
    p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=3.0)
    inp = build_trials(p, 0, nt, sweep_idx=6)
    hyp = [inp[k].numpy() for k in ("tau_Y", "tau_Z", "rho")]
    h = {k: inp[k].cpu().numpy() for k in ("subY", "Omega", "B")}
    A = inp["A"].cpu().numpy()
    zb = inp["Zbar"].cpu().numpy().astype(np.complex128)
    ks = [1, 2, 3, 5, 8, 12, 20, 35, 60, 100]
    ref = {}
    for k in ks:
        ref[k] = bp.proposed_algorithm(lib, h["subY"], h["Omega"], A, h["B"], k, *hyp, want_ce=False, threads=nt)[0]
    for v in variants:
        env = dict(kv.split("=") for kv in v.split(",") if kv)
        os.environ.update(env)
        print("variant", v or "default")
        for k in ks:
            S, _, _ = J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], k, *hyp, "approximate", want_ce=False)
            torch.cuda.synchronize()
            Sh = S.cpu().numpy().astype(np.complex128)
            ds = [np.max(np.abs(Sh[t] - ref[k][t])) / max(np.max(np.abs(ref[k][t])), 1e-300) for t in range(nt)]
            fro = [np.linalg.norm(Sh[t] - ref[k][t]) / max(np.linalg.norm(ref[k][t]), 1e-300) for t in range(nt)]
            dn = [O.nmse_capped(Sh[t], zb[t]) - O.nmse_capped(ref[k][t], zb[t]) for t in range(nt)]
            sb = [float(np.real(np.vdot(ref[k][t], Sh[t] - ref[k][t])) / np.linalg.norm(ref[k][t]) ** 2) for t in range(nt)]
            print("  Imax %3d  max rel dS %.2e (median %.2e)  fro rel dS %.2e  rms dNMSE %.2e  max %.2e  mean dNMSE %+.2e  scale bias %+.2e +- %.1e" %
                  (k, max(ds), float(np.median(ds)), float(np.median(fro)), float(np.sqrt(np.mean(np.square(dn)))), float(np.max(np.abs(dn))),
                   float(np.mean(dn)), float(np.mean(sb)), float(np.std(sb) / np.sqrt(nt))), flush=True)
        for kk in env:
            os.environ.pop(kk, None)


if __name__ == "__main__":
    main()
