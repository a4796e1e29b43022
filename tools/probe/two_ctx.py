#!/usr/bin/env python3
"""Do two half-batches solved CONCURRENTLY on two contexts (two streams, two host threads) beat one full batch?
(One group's pass could run beside the other group's window between passes.)"""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
from jstsp19_amd import _lib
from jstsp19_amd.system_model import SweepParams
from bench import make_inputs
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ways = int(sys.argv[2]) if len(sys.argv) > 2 else 2
want_ce = (sys.argv[3] != "noce") if len(sys.argv) > 3 else True
dev = torch.device("cuda:0")
p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
inp = make_inputs(p, list(range(batch)), dev)
torch.cuda.synchronize()

def solve(sl, ctx, stream, out):
    with torch.cuda.stream(stream):
        S, Y, ce = J.proposed_algorithm(inp["subY"][sl], inp["Omega"][sl], inp["A"], inp["B"][sl], 100, inp["tau_Y"][sl], inp["tau_Z"][sl],
                                        inp["rho"][sl], "approximate", want_ce=want_ce, ctx=ctx)
        stream.synchronize()
    out.append(S)

def run(n):
    h = batch // n
    ctxs = [_lib.Context(0) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    best = None
    for rep in range(4):
        outs = [[] for _ in range(n)]
        th = [threading.Thread(target=solve, args=(slice(k * h, (k + 1) * h), ctxs[k], streams[k], outs[k])) for k in range(n)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    S = torch.cat([o[0] for o in outs])
    print("%d x %d trials concurrently: %.1f ms  -> %.1f channel-estimates/s" % (n, h, best * 1e3, batch / best), flush=True)
    return S

S1 = run(1)
Sn = run(ways)
print("max |S_concurrent - S_single| / max|S| = %.2e" % float((Sn - S1).abs().max() / S1.abs().max()))
