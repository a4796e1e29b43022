#!/usr/bin/env python3
"""Duration quantiles per kernel from a rocprofv3 rocpd database (min / 10 % / median / 90 % / max, us): a launch that lasts as long as
its slowest workgroup shows as a long upper tail.   Usage: kquant_db.py DB [min_calls]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 50
per = {}
for name, d in c.execute("select name, (end-start)/1e3 from kernels"):
    per.setdefault(name, []).append(d)
print("%-70s %6s %8s %8s %8s %8s %8s" % ("kernel", "calls", "min", "p10", "median", "p90", "max"))
for name, ds in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if len(ds) < minc:
        continue
    ds.sort()
    q = lambda f: ds[min(len(ds) - 1, int(f * len(ds)))]
    print("%-70s %6d %8.1f %8.1f %8.1f %8.1f %8.1f" % (name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70], len(ds), ds[0], q(0.1), q(0.5), q(0.9), ds[-1]))
