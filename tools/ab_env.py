#!/usr/bin/env python
"""A/B/A/B of one callback under several settings of a code-generator knob IN ONE PROCESS (timings of separate gpurun
calls differ by more than the effects being looked for).  usage: ab_env.py KNOB v1 v2 ... [-- callback]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

args = sys.argv[1:]
cb = "hess"
if "--" in args:
    cb = args[args.index("--") + 1]
    args = args[:args.index("--")]
knob, values = args[0], args[1:]
N = int(float(os.environ.get("SWEEP_N", "1e7")))
which = os.environ.get("SWEEP_MODEL", "lv")
core = {"lv": lambda: models.luksan_vlcek_model(N), "rocket": lambda: models.rocket_model(1_000_000),
        "acopf": lambda: models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0))}[which]()
runs = {}
for v in values:
    os.environ[knob] = v
    m = ExaModel(core)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, m.meta.nvar)).to(dev)
    y = torch.from_numpy(np.random.default_rng(1).standard_normal(m.meta.ncon)).to(dev)
    n_out = {"hess": m.meta.nnzh, "jac": m.meta.nnzj, "cons": m.meta.ncon, "grad": m.meta.nvar}[cb]
    out = torch.empty(n_out, dtype=torch.float64, device=dev)
    m.time_callback(cb, 50, x, y, 0.5, out=out)
    runs[v] = (m, x, y, out, [])
for rnd in range(6):
    for v in values:
        m, x, y, out, acc = runs[v]
        acc.append(m.time_callback(cb, 200, x, y, 0.5, out=out))
print(which, knob, cb, N, {v: (round(min(r[4]), 5), round(float(np.median(r[4])), 5)) for v, r in runs.items()})
