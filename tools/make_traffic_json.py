#!/usr/bin/env python
"""profiles/r5_traffic_config<k>.json from the PMC passes of tools/refresh_profiles_r5.sh: HBM bytes per launch of the
Hessian kernel = WRITE_SIZE + corrected FETCH_SIZE (separate rocprofv3 --pmc passes, per-dispatch averages), stamped with
the NAME OF THE MODULE it was measured on and the workload size — bench.py attaches the number only when both match.
usage: make_traffic_json.py CONFIG bench.json fetch_summary.txt write_summary.txt > profiles/r2_traffic_configK.json"""
import json
import re
import sys

config, bench, fetch, write = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
line = json.loads(open(bench).read().strip().splitlines()[-1])
kernel = line["roofline"]["kernel"]


def counter(path, name):
    for ln in open(path):
        m = re.match(r"\s*(\S+)\s+" + name + r"\s+n=\s*(\d+)\s+avg=\s*([\d.eE+-]+)", ln)
        if m and m.group(1).startswith(kernel):
            return float(m.group(3)), int(m.group(2))
    raise SystemExit(f"{name} of {kernel} not found in {path}")


def counter_of(path, name, kern):
    for ln in open(path):
        m = re.match(r"\s*(\S+)\s+" + name + r"\s+n=\s*(\d+)\s+avg=\s*([\d.eE+-]+)", ln)
        if m and m.group(1) == kern:
            return float(m.group(3))
    return None


f_kb, nf = counter(fetch, "FETCH_SIZE")
w_kb, nw = counter(write, "WRITE_SIZE")
CORR = 1.9391      # profiles/r1_pmc_calibration_store_bench.txt: a kernel reading a known 156250 KB in this 8-B/lane pattern reports 80578.89 KB
out = {
    "workload": line["config"]["workload"], "baseline_config": config, "points": line["config"].get("points"),
    "module": line["build"]["module_name"], "kernel": kernel,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/refresh_profiles_r5.sh), per-dispatch averages",
    "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "dispatches": [nf, nw],
    "fetch_correction": CORR,
    "fetch_correction_source": "profiles/r1_pmc_calibration_store_bench.txt (the guide's gfx950 'FETCH_SIZE reports 1/2', calibrated on this access pattern); WRITE_SIZE needs none",
    "hbm_bytes_per_launch": int(round(1024 * (w_kb + CORR * f_kb))),
    "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
}
# exa_tune launches every hess_coord! kernel of the module in the same process: their counters are in the same passes.  Recorded per
# kernel, so that a later process whose exa_tune picks the sibling (the chained pair is within 1 % of each other) still finds ITS number.
CORR_ = CORR
per = {}
for kern in ("exa_hess", "exa_hessc", "exa_hesscl"):
    fk, wk = counter_of(fetch, "FETCH_SIZE", kern), counter_of(write, "WRITE_SIZE", kern)
    if fk is not None and wk is not None:
        per[kern] = int(round(1024 * (wk + CORR_ * fk)))
out["hbm_bytes_per_launch_by_kernel"] = per
print(json.dumps(out, indent=1))
