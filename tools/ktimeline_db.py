#!/usr/bin/env python3
"""Timeline of one steady-state iteration from a rocprofv3 rocpd database: kernels between two consecutive launches of the
anchor kernel (default fused_pass), start / end relative to the first, queue id.  Usage: ktimeline_db.py DB [anchor] [which]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "fused_pass"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 150
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
a, b = idx[which], idx[which + 1]
t0 = rows[a][1]
for r in rows[a:b + 1]:
    print("%9.1f %9.1f  q=%s  %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, r[3] if qcol else "-", r[0][:70]))
