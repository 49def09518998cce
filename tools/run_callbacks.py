#!/usr/bin/env python
"""Runs EVERY callback of one BASELINE config `reps` times on one MI355X and prints one JSON object: per callback the
kernels it launches, its algorithmic HBM bytes and the hipEvent time per call.  This is the command the
round-3 rocprofv3 passes wrap (tools/refresh_profiles_r3.sh): `--kernel-trace --stats` gives the per-kernel durations,
separate `--pmc` passes the counters; tools/roofline_table.py joins the three into profiles/r3_kernels_config<k>.md.

usage: run_callbacks.py CONFIG [--reps R] [--only a,b,...] [--points N] [--topology random|bus]
Matches benchmark/runbenchmark.jl:79-101 (all five callbacks) plus the products, the fused sweeps and the compressed COO."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import CompressedExaModel, ExaModel  # noqa: E402

# callback -> the generated kernels that carry its traffic (the first one is the dominant kernel)
KERNELS = {
    "obj": ["exa_obj"], "grad": ["exa_grad_pull", "exa_grad", "exa_gradw"], "cons": ["exa_cons1", "exa_cons", "exa_consl", "exa_aug_gather"],
    "jac": ["exa_jac", "exa_jacl"], "hess": ["exa_hess", "exa_hesscl", "exa_hessc"], "jprod": ["exa_jprod1", "exa_jprod"],
    "jtprod": ["exa_jtprodw", "exa_jtprods", "exa_jtprodx", "exa_jtprod"], "hprod": ["exa_hprodw", "exa_hprods", "exa_hprodx", "exa_hprod"],
    "fused": ["exa_fused"], "eval_all": ["exa_fused", "exa_grad_pull", "exa_grad"],
    "chess": ["exa_chessw", "exa_chessm", "exa_chessp", "exa_chesss", "exa_chessx"], "cjac": ["exa_cjacw", "exa_cjacp", "exa_cjacs", "exa_cjacx"],
}


def iterator_bytes(m):
    tot = 0
    for k in range(m.npatterns):
        pat = m.ir.patterns[k]
        for c in range(pat.n_cols):
            if pat.cols[c].type != 2:
                tot += 8 * pat.n
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", type=int, choices=[2, 3, 4, 5])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--points", type=float, default=None)
    ap.add_argument("--topology", default="random", choices=["random", "bus"])
    args = ap.parse_args()
    import bench
    if args.config == 4 and args.topology == "bus":
        from exahip import models
        core = models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0, topology="bus"))
    else:
        pts = int(args.points or {2: 1e7, 3: 1e6, 4: 0, 5: 1e8}[args.config])
        core = bench.build_core(args.config, pts)
    m = ExaModel(core)
    x, y = bench.eval_point(args.config, core, m)
    dev = torch.device("cuda:0")
    nvar, ncon, nnzj, nnzh = m.meta.nvar, m.meta.ncon, m.meta.nnzj, m.meta.nnzh
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    vd = torch.from_numpy(np.random.default_rng(2).standard_normal(nvar)).to(dev)
    wd = torch.from_numpy(np.random.default_rng(3).standard_normal(max(1, ncon))).to(dev)
    g = torch.empty(nvar, dtype=torch.float64, device=dev)
    c = torch.empty(max(1, ncon), dtype=torch.float64, device=dev)
    j = torch.empty(max(1, nnzj), dtype=torch.float64, device=dev)
    h = torch.empty(max(1, nnzh), dtype=torch.float64, device=dev)
    f = torch.zeros(1, dtype=torch.float64, device=dev)
    itb = iterator_bytes(m)
    only = [s for s in args.only.split(",") if s]
    want = lambda n: not only or n in only       # noqa: E731
    calls, alg = {}, {}
    if want("obj"):
        calls["obj"] = lambda: m._L.exa_obj_async(m.id, xd.data_ptr(), f.data_ptr()); alg["obj"] = 8 * nvar + itb
    if want("grad"):
        calls["grad"] = lambda: m.grad(xd, out=g); alg["grad"] = 16 * nvar + itb
    if want("cons"):
        calls["cons"] = lambda: m.cons(xd, out=c); alg["cons"] = 8 * (ncon + nvar) + itb
    if want("jac"):
        calls["jac"] = lambda: m.jac_coord(xd, out=j); alg["jac"] = 8 * (nnzj + nvar) + itb
    if want("hess"):
        calls["hess"] = lambda: m.hess_coord(xd, yd, 0.5, out=h); alg["hess"] = 8 * (nnzh + nvar + ncon) + itb
    if want("jprod"):
        calls["jprod"] = lambda: m.jprod(xd, vd, out=c); alg["jprod"] = 8 * (2 * nvar + ncon) + itb
    if want("jtprod"):
        calls["jtprod"] = lambda: m.jtprod(xd, wd, out=g); alg["jtprod"] = 8 * (2 * nvar + ncon) + itb
    if want("hprod"):
        calls["hprod"] = lambda: m.hprod(xd, yd, vd, 0.5, out=g); alg["hprod"] = 8 * (3 * nvar + ncon) + itb
    if want("fused"):
        calls["fused"] = lambda: m.eval_fused(xd, yd, 0.5, c=c, jac=j, hess=h, obj_out=f)
        alg["fused"] = 8 * (nnzh + nnzj + ncon) + 8 * (nvar + ncon) + itb
    if want("eval_all") and hasattr(m, "eval_all"):
        calls["eval_all"] = lambda: m.eval_all(xd, yd, 0.5, g=g, c=c, jac=j, hess=h, obj_out=f)
        alg["eval_all"] = 8 * (nnzh + nnzj + ncon + nvar) + 8 * (nvar + ncon) + itb
    cm = None
    if (want("chess") or want("cjac")) and args.config != 5:
        cm = CompressedExaModel(m)
        ch = torch.empty(max(1, cm.meta.nnzh), dtype=torch.float64, device=dev)
        cj = torch.empty(max(1, cm.meta.nnzj), dtype=torch.float64, device=dev)
        if want("chess"):
            calls["chess"] = lambda: cm.hess_coord(xd, yd, 0.5, out=ch); alg["chess"] = 8 * (cm.meta.nnzh + nvar + ncon) + itb
        if want("cjac"):
            calls["cjac"] = lambda: cm.jac_coord(xd, out=cj); alg["cjac"] = 8 * (cm.meta.nnzj + nvar) + itb
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {"config": args.config, "workload": bench.CONFIGS[args.config] + (" [bus-ordered topology]" if args.topology == "bus" else ""),
           "module": m._L.exa_module_name(m.id).decode(), "nvar": nvar, "ncon": ncon, "nnzj": nnzj, "nnzh": nnzh,
           "reps": args.reps, "callbacks": {}}
    if cm is not None:
        out["cnnzj"], out["cnnzh"] = cm.meta.nnzj, cm.meta.nnzh
        out["compressed_path"] = {"hess": cm.path("hess"), "jac": cm.path("jac")}
    # keep the clocks up before the first timed callback
    for _ in range(50):
        m.hess_coord(xd, yd, 0.5, out=h)
    floor_ms = m.time_callback("launch", 500, xd)      # an (almost) empty launch of the model's module on the same stream
    for name, fn in calls.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        # a launch whose inputs + outputs fit the 256 MiB Infinity Cache is not an HBM measurement: no fraction of the HBM peak for it
        # (round 3 printed 1.15 - 2.5 there), its time against the launch floor instead
        resident = alg[name] < 256 * 1024 * 1024
        out["callbacks"][name] = {"ms": ms, "algorithmic_bytes": alg[name], "GBps": alg[name] / ms / 1e6,
                                  "bound": "mall (cache-resident / launch-bound)" if resident else "hbm",
                                  "frac_of_8TBps": None if resident else alg[name] / ms / 1e6 / 8000.0,
                                  "x_launch_floor": ms / floor_ms, "kernels": KERNELS[name]}
    out["launch_floor_ms"] = floor_ms
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
