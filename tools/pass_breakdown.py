#!/usr/bin/env python3
"""Where the fused pass's time goes: the library built with -DJSTSP_FUSED_DBG_BUILD runs the pass with parts switched off
(JSTSP_FUSED_DBG bit mask: 1 phase-A products, 2 element-wise loads / stores, 4 phase-B products, 8 tile refill; results are
wrong, timing only).  Prints the average duration of fused_pass per setting, for the compact (block-Toeplitz) and the full
dictionary image.  args: batch [Imax]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jstsp19_amd as J
from jstsp19_amd.system_model import SweepParams, build_trials

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Imax = int(sys.argv[2]) if len(sys.argv) > 2 else 8
p = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8, snr_db=5.0)
inp = build_trials(p, 0, batch, seed=1)
ctx = J.default_context(0)
ctx.set_profiling(True)
last = [0, 0.0]
def run(dbg, toep):
    os.environ["JSTSP_FUSED_DBG"] = str(dbg)
    os.environ["JSTSP_TOEPLITZ"] = str(toep)
    J.proposed_algorithm(inp["subY"], inp["Omega"], inp["A"], inp["B"], Imax, inp["tau_Y"].numpy(), inp["tau_Z"].numpy(),
                         inp["rho"].numpy(), "approximate")
    torch.cuda.synchronize()
    n, ms = ctx.get_profile("fused_pass")
    dn, dms = n - last[0], ms - last[1]
    last[0], last[1] = n, ms
    return dms / max(dn, 1)
names = {0: "everything", 1: "no phase-A products", 2: "no element-wise loads/stores", 4: "no phase-B products", 8: "no refill",
         5: "no products", 7: "refill only", 13: "element-wise only", 15: "nothing (barriers, k fragments)", 10: "products only"}
for toep in (2, 1, 0):
    run(0, toep)
    for dbg in (0, 1, 2, 4, 8, 5, 10, 7, 13, 15):
        print("toeplitz=%d dbg=%2d  %-34s %.3f ms" % (toep, dbg, names[dbg], run(dbg, toep)), flush=True)
