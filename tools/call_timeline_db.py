#!/usr/bin/env python3
"""Kernels of ONE solver call from a rocprofv3 rocpd database: everything between two consecutive launches of the anchor kernel
(default: the first setup kernel of a call), start / duration in us.  Usage: call_timeline_db.py DB [anchor] [which]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "inv_d_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
a = idx[which]; b = idx[which + 1] if which + 1 < len(idx) else len(rows)
t0 = rows[a][1]; tot = 0.0
for r in rows[a:b]:
    d = (r[2] - r[1]) / 1e3; tot += d
    print("%9.1f %8.1f  %s" % ((r[1] - t0) / 1e3, d, r[0][:90]))
print("kernel time %.1f us, span %.1f us" % (tot, (rows[b - 1][2] - t0) / 1e3))
