#!/usr/bin/env python3
"""Between-pass windows from a rocprofv3 rocpd database: for every pair of consecutive fused_pass launches the pass
duration, the window (end of one pass -> start of the next) and the busy time of each queue inside the window.
Usage: kwindows_db.py DB [first] [count]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
count = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = c.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
idx = [i for i, r in enumerate(rows) if "fused_pass" in r[0]]
wins = []
for n in range(len(idx) - 1):
    a, b = idx[n], idx[n + 1]
    wins.append(((rows[a][2] - rows[a][1]) / 1e3, (rows[b][1] - rows[a][2]) / 1e3))
import statistics
w = [x[1] for x in wins if x[1] < 5000]
print("passes %d; pass us mean %.1f; window us mean %.1f median %.1f min %.1f max %.1f" % (
    len(idx), statistics.mean(x[0] for x in wins), statistics.mean(w), statistics.median(w), min(w), max(w)))
for n in range(first, min(first + count, len(idx) - 1)):
    a, b = idx[n], idx[n + 1]
    t0 = rows[a][2]
    busy = {}
    names = {}
    for r in rows[a + 1:b]:
        busy[r[3]] = busy.get(r[3], 0) + (r[2] - r[1]) / 1e3
        names.setdefault(r[3], []).append("%s:%.0f-%.0f" % (r[0].split("(")[0].split("::")[-1][:14], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3))
    print("it %d: pass %.0f us, window %.0f us" % (n, wins[n][0], wins[n][1]))
    for q in sorted(busy):
        print("    q%s busy %.0f us: %s" % (q, busy[q], " ".join(names[q])))
