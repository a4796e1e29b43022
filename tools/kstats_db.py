#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes
DIR/NAME_results.db on ROCm 7.2): name, calls, average us, total ms, share.  Usage: kstats_db.py DB [min_share_%] [split]
`split`: calls of one kernel whose durations differ by more than 2x are listed as separate rows (one kernel name launched with
two problem shapes, e.g. K B^H and the G_B apply through the same hgemm_kernel instance)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
split = len(sys.argv) > 3 and sys.argv[3] == "split"
if not split:
    rows = c.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc").fetchall()
else:
    per = {}
    for name, d in c.execute("select name, (end-start)/1e3 from kernels"):
        per.setdefault(name, []).append(d)
    rows = []
    for name, ds in per.items():
        ds.sort()
        grp = [ds[0]]
        k = 0
        for d in ds[1:]:
            if d > 2.0 * grp[0]:
                rows.append((name + (" [cluster %d]" % k), len(grp), sum(grp) / len(grp), sum(grp) / 1e3))
                grp, k = [d], k + 1
            else:
                grp.append(d)
        rows.append((name + (" [cluster %d]" % k if k else ""), len(grp), sum(grp) / len(grp), sum(grp) / 1e3))
    rows.sort(key=lambda r: -r[3])
tot = sum(r[3] for r in rows) or 1.0
print("Name,Calls,AverageUs,TotalMs,Percentage")
for name, n, avg, total in rows:
    if 100 * total / tot >= thr:
        print('"%s",%d,%.1f,%.2f,%.2f' % (name[:110] + name[110:][-12:] if "[cluster" in name else name[:110], n, avg, total, 100 * total / tot))
