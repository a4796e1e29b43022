#!/usr/bin/env python3
"""Kernel sequence from a rocprofv3 rocpd database: start offset (us), duration (us), gap to the previous kernel's end, short name.
Usage: kseq_db.py DB [first] [count]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rows = c.execute("select name, start, end from kernels order by start").fetchall()[first:first + count]
t0, prev = rows[0][1], rows[0][1]
for name, s, e in rows:
    short = name.split("(")[0].split("::")[-1][:40]
    print("%10.1f %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short))
    prev = e
