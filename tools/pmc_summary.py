#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (average per dispatch)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in rows:
    k = r["Kernel_Name"][:72] + " grid=" + r.get("Grid_Size", "?")
    if pat and pat not in r["Kernel_Name"]:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in disp[k]:
        disp[k].add(r["Dispatch_Id"])
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in agg:
    n = len(disp[k])
    print(k, "dispatches", n, "avg_us %.1f" % (dur[k] / n / 1e3))
    for c, v in sorted(agg[k].items()):
        print("   %-28s %.4g" % (c, v / n))
    a = agg[k]
    if "GRBM_GUI_ACTIVE" in a and dur[k] > 0:
        cyc = a["GRBM_GUI_ACTIVE"] / n / 8.0
        print("   clock_GHz(est, /8 XCD)       %.3f" % (cyc / (dur[k] / n)))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            print("   mfma_util(est)               %.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (cyc * 1024)))
