#!/usr/bin/env python
"""Summarise rocprofv3 rocpd databases (ROCm 7.2 default output) into a small text report for profiles/.

usage: tools/prof_summary.py <dir-with-*_results.db>... > profiles/<name>.txt
"""
import glob
import os
import sqlite3
import sys


def summarise(db_path, out):
    db = sqlite3.connect(db_path)
    out.write(f"== {db_path}\n")
    try:
        rows = list(db.execute(
            "select name, count(*), avg(duration), min(duration), max(duration), sum(duration), max(vgpr_count), max(sgpr_count), "
            "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"))
        tot = sum(r[5] for r in rows) or 1
        out.write("kernel-trace stats (durations in us):\n")
        out.write(f"  {'kernel':40s} {'calls':>6s} {'avg':>10s} {'min':>10s} {'max':>10s} {'%':>6s} {'vgpr':>5s} {'sgpr':>5s} {'grid':>10s} {'wg':>4s}\n")
        for n, c, a, mn, mx, s, vg, sg, gx, wx in rows:
            out.write(f"  {n[:40]:40s} {c:6d} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f} {vg:5d} {sg:5d} {gx:10d} {wx:4d}\n")
    except sqlite3.Error as e:
        out.write(f"  (no kernel table: {e})\n")
    try:
        rows = list(db.execute(
            "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name, counter_name"))
        if rows:
            out.write("PMC counters per dispatch (avg / min / max):\n")
            for k, cn, c, a, mn, mx in rows:
                out.write(f"  {k[:40]:40s} {cn:24s} n={c:4d} avg={a:16.2f} min={mn:16.2f} max={mx:16.2f}\n")
    except sqlite3.Error:
        pass


def main():
    for d in sys.argv[1:]:
        paths = sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)) if os.path.isdir(d) else [d]
        for p in paths:
            summarise(p, sys.stdout)


if __name__ == "__main__":
    main()
