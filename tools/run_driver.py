#!/usr/bin/env python3
"""Any of the reference's sweep drivers at its own parameters on the HIP path:
  errorVSsnr | errorVSdelays | errorVSframelength | errorVSnrf | errorVSnt | errorVSpaths | rateVSframelength
  (columns proposed, proposed+angles, LS, VAMP, MMV-OMP), errorVSadmmiters (mean convergence curves, four panels),
  errorVSzy (Z vs Y estimate)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jstsp19_amd import montecarlo as mc

ap = argparse.ArgumentParser()
ap.add_argument("name")
ap.add_argument("--trials", type=int, default=None, help="default: the driver's own maxMCRealizations")
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
t0 = time.perf_counter()
if a.name == "errorVSadmmiters":
    n = a.trials or 20
    out = mc.run_convergence_curves(mc.admmiters_points(), n, batch=min(a.batch, n))
    torch.cuda.synchronize()
    print("mean convergence_error over %d realisations, %.1f s" % (n, time.perf_counter() - t0))
    for k, p in enumerate(mc.admmiters_points()):
        print("panel %d: Nt=%d frame=%d SNR=%g dB   iteration: eps1 eps2 (Algorithm) | eps1 eps2 (with angles), dB"
              % (k + 1, p.Nt, p.T_prop, p.snr_db))
        for it in (0, 1, 4, 9, 19, 49, 99):
            r = 10 * torch.log10(out[k, :, it, :])
            print("  %3d: %8.2f %8.2f | %8.2f %8.2f" % (it + 1, r[0, 0], r[0, 2], r[1, 0], r[1, 2]))
elif a.name == "errorVSzy":
    n = a.trials or 1
    out = mc.run_zy(None, n, batch=min(a.batch, n))
    torch.cuda.synchronize()
    print("capped NMSE, %d realisations, %.1f s:  Z %.6f   Y %.6f" % (n, time.perf_counter() - t0, out[0, 0], out[0, 1]))
else:
    d = mc.driver(a.name)
    n = a.trials or d["n_trials"]
    out = mc.run_driver(a.name, n, batch=min(a.batch, n))
    torch.cuda.synchronize()
    print("%s (%s), %d trials/point, %.1f s" % (a.name, d["metric"], n, time.perf_counter() - t0))
    print("%-8s proposed  +angles   LS        VAMP      MMV-OMP" % d["axis"])
    for v, row in zip(d["values"], out.tolist()):
        print("%-8s " % v + "  ".join("%.6f" % x for x in row))
