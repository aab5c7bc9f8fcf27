#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace/stats or a --pmc pass) as text:
per-kernel call count, total/avg/min/max duration, and — when counters were collected — per-kernel counter means.
Usage: tools/rocpd_summary.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("# rocprofv3 summary of", path)
    rows = cur.execute("select name, start, end from kernels").fetchall()
    regs = {r[0]: r[1:] for r in cur.execute("select name, max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name")}
    by = defaultdict(list)
    for r in rows:
        by[r[0]].append((r[2] - r[1]) / 1e3)
    tot = sum(sum(v) for v in by.values())
    print("%-90s %7s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print("%-90s %7d %12.1f %12.1f %12.1f %12.1f %6.2f%%" % (name[:90], len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / max(tot, 1e-9)))
        print("    vgpr %s sgpr %s lds %s grid %s wg %s" % regs.get(name, ("?",) * 5))
    try:
        pcols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
        if pcols:
            q = "select name, counter_name, avg(counter_value), sum(counter_value), count(*) from pmc_events group by name, counter_name"
            rows = cur.execute(q).fetchall()
            if rows:
                print("\n%-90s %-16s %16s %18s %7s" % ("kernel", "counter", "mean/dispatch", "sum", "n"))
                for r in rows:
                    print("%-90s %-16s %16.1f %18.1f %7d" % (r[0][:90], r[1], r[2], r[3], r[4]))
    except sqlite3.Error as e:
        print("# no pmc data:", e)


if __name__ == "__main__":
    main(sys.argv[1])
