#!/usr/bin/env python3
"""The kernels of a rocprofv3 --kernel-trace run in launch order, one line each (start offset, duration, grid, name): which kernel a gap or an
unexpected fill belongs to.   tools/kernel_timeline.py <results.db> [first] [count] [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
t0 = rows[0][1] if rows else 0
prev_end = t0
for i, (name, s, e, gx, wx) in enumerate(rows):
    if i < first or i >= first + count:
        prev_end = e
        continue
    if (e - s) / 1e3 >= min_us:
        print("%5d  +%10.1f us  gap %8.1f  dur %9.1f us  grid %9s wg %4s  %s" % (i, (s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, gx, wx, name[:100]))
    prev_end = e
