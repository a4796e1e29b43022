#!/usr/bin/env python3
"""Experiment: the 256 trials of the headline workload as ONE call vs as `parts` calls on separate contexts / streams issued
from concurrent host threads (software pipelining of independent sub-batches: one part's latency-bound window between
two passes under another part's HBM-bound pass).  usage: split_experiment.py [batch] [Imax] [parts ...]"""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J
from jstsp19_amd.system_model import SweepParams, build_trials
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Imax = int(sys.argv[2]) if len(sys.argv) > 2 else 100
parts_list = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
OFFSET_MS = float(os.environ.get("OFFSET_MS", "0"))
p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
inp = build_trials(p, 0, batch, seed=1)
tY, tZ, rh = inp["tau_Y"].numpy(), inp["tau_Z"].numpy(), inp["rho"].numpy()
torch.cuda.synchronize()
ref = None
for parts in parts_list:
    ctxs = [J.Context(0) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    bounds = [(batch * i // parts, batch * (i + 1) // parts) for i in range(parts)]
    outs = [None] * parts

    def work(i):
        lo, hi = bounds[i]
        with torch.cuda.stream(streams[i]):
            outs[i] = J.proposed_algorithm(inp["subY"][lo:hi], inp["Omega"][lo:hi], inp["A"], inp["B"][lo:hi], Imax, tY[lo:hi],
                                           tZ[lo:hi], rh[lo:hi], "approximate", ctx=ctxs[i])
            streams[i].synchronize()

    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
        for k, t in enumerate(th):
            t.start()
            if OFFSET_MS > 0 and k + 1 < len(th): time.sleep(OFFSET_MS * 1e-3)
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("parts %d rep %d: %.3f s  %.1f estimates/s  %.3f ms/iteration" % (parts, rep, dt, batch / dt, 1e3 * dt / Imax), flush=True)
    S = torch.cat([o[0] for o in outs], 0)
    if ref is None:
        ref = S
    else:
        print("   max |S - S(1 part)| / max|S| = %.2e" % float((S - ref).abs().max() / ref.abs().max()))
    del ctxs
