#!/usr/bin/env python
"""J'v and Hv of a data-indexed model (ACOPF 78 484 buses: every bus variable is reached through the branch table) by the three
implementations the library has for it: 0 FP64 atomics in the sweep, 1 COO + sorted gather (the reference's scheme), 3 owner pull
(exa_gen_pull.cpp: a thread per variable re-evaluates the items that land on it).  Event-bracketed ms per call, A/B rounds, the
results compared with each other; random and bus-ordered topology.  usage (GPU box): python tools/pull_ab.py [reps=300]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
for topo in ("random", "bus"):
    m = ExaModel(models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0, topology=topo)))
    nvar, ncon = m.meta.nvar, m.meta.ncon
    r = np.random.default_rng(0)
    x = torch.from_numpy(m.meta.x0 + 0.05 * r.uniform(-1, 1, nvar)).to(dev)
    y = torch.from_numpy(r.standard_normal(ncon)).to(dev)
    v = torch.from_numpy(r.standard_normal(nvar)).to(dev)
    w = torch.from_numpy(r.standard_normal(ncon)).to(dev)
    out = torch.empty(nvar, dtype=torch.float64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    floor = m.time_callback("launch", 500, x)
    print(f"ACOPF 78 484 buses, {topo} topology: nvar {nvar}, ncon {ncon}; launch floor {1e3 * floor:.1f} us; {m.product_info('jtprod')[1]}", flush=True)
    res, ref = {}, {}
    for mode in (0, 1, 3):
        try:
            m.set_product_mode(mode, mode)
        except Exception as e:      # noqa: BLE001
            print(f"  mode {mode}: not available ({e})")
            continue
        for name, call in (("jtprod", lambda: m.jtprod(x, w, out=out)), ("hprod", lambda: m.hprod(x, y, v, 0.7, out=out))):
            out.fill_(float("nan"))
            call()
            torch.cuda.synchronize()
            got = out.cpu().numpy().copy()
            if name not in ref:
                ref[name] = got
            err = float(np.max(np.abs(got - ref[name]) / np.maximum(1.0, np.abs(ref[name]))))
            call(); torch.cuda.synchronize()
            same = bool(np.array_equal(out.cpu().numpy(), got))
            ts = []
            for _ in range(5):
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / reps)
            res[(mode, name)] = min(ts)
            print(f"  mode {mode} {name:7s} {1e3 * min(ts):7.1f} us/call  (x{min(ts) / floor:4.1f} launch floor)  max difference from mode 0: {err:.1e}  run-to-run identical: {same}", flush=True)
    m.set_product_mode(-1, -1)
    del m
