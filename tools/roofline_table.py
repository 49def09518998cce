#!/usr/bin/env python
"""Joins one tools/run_callbacks.py JSON with the rocprofv3 databases of the SAME command (a --kernel-trace --stats run and
separate --pmc passes, tools/refresh_profiles_r5.sh) into one table per config: for every generated kernel its average
duration, the algorithmic bytes of the callback it serves, achieved GB/s against the 8 TB/s HBM peak, HBM traffic from
the counters (WRITE_SIZE + 1.94 x FETCH_SIZE: the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, calibrated
in profiles/r1_pmc_calibration_store_bench.txt), VALU wavefront-instructions, the wait fractions and the busy cycles.
Every fraction quoted in DESIGN.md §5 is recomputable from this file.

usage: roofline_table.py callbacks.json STATS_DIR [PMC_DIR...] > profiles/r5_kernels_config<k>.md"""
import glob
import json
import os
import sqlite3
import sys

CORR = 1.9391
PEAK = 8000.0


def dbs(d):
    return sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True))


def kernel_stats(d):
    out = {}
    for p in dbs(d):
        db = sqlite3.connect(p)
        for n, c, a, mn, vg in db.execute("select name, count(*), avg(duration), min(duration), max(vgpr_count) from kernels group by name"):
            out[n.split("(")[0]] = {"calls": c, "avg_us": a / 1e3, "min_us": mn / 1e3, "vgpr": vg}
    return out


def counters(dirs):
    out = {}
    for d in dirs:
        for p in dbs(d):
            db = sqlite3.connect(p)
            try:
                rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name")
                for k, cn, c, a in rows:
                    out.setdefault(k.split("(")[0], {})[cn] = a
            except sqlite3.Error:
                pass
    return out


def main():
    cb = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks = kernel_stats(sys.argv[2])
    pm = counters(sys.argv[3:])
    w = sys.stdout.write
    w(f"# {cb['workload']}\n\nmodule `{cb['module']}`; nvar {cb['nvar']}, ncon {cb['ncon']}, nnzj {cb['nnzj']}, nnzh {cb['nnzh']}"
      + (f", compressed nnzj {cb['cnnzj']}, nnzh {cb['cnnzh']}" if "cnnzj" in cb else "") + "\n\n")
    w("Per callback: hipEvent time per call (tools/run_callbacks.py, 200 calls, NO profiler attached), algorithmic bytes (SURVEY §8d).\n"
      "Per kernel: rocprofv3 --kernel-trace average duration; `frac` = algorithmic bytes of the callback / kernel duration / 8 TB/s (dominant kernel only);\n"
      f"`traffic` = WRITE_SIZE + {CORR} x FETCH_SIZE per dispatch (separate --pmc passes); VALU = SQ_INSTS_VALU wavefront-instructions per dispatch;\n"
      "wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES, issue-stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; GUI = GRBM_GUI_ACTIVE cycles per dispatch (all XCDs).\n\n")
    w("| callback | ms/call (events) | alg. MB | kernel | calls | avg us | GB/s | frac | traffic MB | traffic/alg | VALU M | wait | issue-stall | GUI kcyc | vgpr |\n")
    w("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for name, c in cb["callbacks"].items():
        first = True
        for k in c["kernels"]:
            if k not in ks:
                continue
            s, p = ks[k], pm.get(k, {})
            gbs = c["algorithmic_bytes"] / s["avg_us"] / 1e3
            resident = c["algorithmic_bytes"] < 256 * 1024 * 1024      # fits the Infinity Cache: no HBM fraction, time against the launch floor
            tr = None
            if "WRITE_SIZE" in p and "FETCH_SIZE" in p:
                tr = 1024.0 * (p["WRITE_SIZE"] + CORR * p["FETCH_SIZE"])
            wc = p.get("SQ_WAVE_CYCLES")
            cells = [name if first else "", f"{c['ms']:.4f}" if first else "", f"{c['algorithmic_bytes'] / 1e6:.1f}" if first else "", f"`{k}`", str(s["calls"]),
                     f"{s['avg_us']:.2f}", f"{gbs:.0f}" if first else "", (f"mall, {s['avg_us'] / (1e3 * cb['launch_floor_ms']):.1f}x launch floor" if resident and cb.get("launch_floor_ms") else ("mall" if resident else f"{gbs / PEAK:.3f}")) if first else "",
                     f"{tr / 1e6:.1f}" if tr else "", f"{tr / c['algorithmic_bytes']:.3f}" if tr and first else "",
                     f"{p['SQ_INSTS_VALU'] / 1e6:.2f}" if "SQ_INSTS_VALU" in p else "",
                     f"{p['SQ_WAIT_ANY'] / wc:.2f}" if wc and "SQ_WAIT_ANY" in p else "", f"{p['SQ_WAIT_INST_ANY'] / wc:.2f}" if wc and "SQ_WAIT_INST_ANY" in p else "",
                     f"{p['GRBM_GUI_ACTIVE'] / 1e3:.1f}" if "GRBM_GUI_ACTIVE" in p else "", str(s["vgpr"])]
            w("| " + " | ".join(cells) + " |\n")
            first = False
    w("\nRaw counters per kernel (per-dispatch averages):\n\n```\n")
    for k in sorted(pm):
        if k.startswith("exa_"):
            w(f"{k:24s} " + "  ".join(f"{cn}={v:.1f}" for cn, v in sorted(pm[k].items())) + "\n")
    w("```\n")


if __name__ == "__main__":
    main()
