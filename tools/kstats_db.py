#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes
DIR/NAME_results.db on ROCm 7.2): name, calls, average us, total ms, share.  Usage: kstats_db.py DB [min_share_%]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc").fetchall()
tot = sum(r[3] for r in rows) or 1.0
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
print("Name,Calls,AverageUs,TotalMs,Percentage")
for name, n, avg, total in rows:
    if 100 * total / tot >= thr:
        print('"%s",%d,%.1f,%.2f,%.2f' % (name[:110], n, avg, total, 100 * total / tot))
