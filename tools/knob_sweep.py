#!/usr/bin/env python
"""LV N=1e7: ms per call of the secondary callbacks under one setting of the code-generator knobs (environment).
Run once per setting (the knobs change the generated source, hence the module):
    EXAHIP_PPT_CONS=2 python tools/knob_sweep.py cons jac"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

which = sys.argv[1:] or ["obj", "cons", "grad", "jac", "hess"]
N = int(float(os.environ.get("SWEEP_N", "1e7")))
m = ExaModel(models.luksan_vlcek_model(N))
dev = torch.device("cuda:0")
x = torch.from_numpy(m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)).to(dev)
y = torch.from_numpy(np.random.default_rng(1).standard_normal(m.meta.ncon)).to(dev)
size = {"obj": 1, "cons": m.meta.ncon, "grad": N, "jac": m.meta.nnzj, "hess": m.meta.nnzh}
knobs = {k: v for k, v in os.environ.items() if k.startswith("EXAHIP_")}
res = {}
for cb in which:
    out = None if cb == "obj" else torch.empty(size[cb], dtype=torch.float64, device=dev)
    quick = os.environ.get("SWEEP_QUICK")                # under rocprofv3 --pmc every dispatch is serialised: keep it short
    for _ in range(1 if quick else 10):
        m.time_callback(cb, 2 if quick else 20, x, y, 0.5, out=out)      # also leaves the idle clock
    res[cb] = min(m.time_callback(cb, 5 if quick else 100, x, y, 0.5, out=out) for _ in range(1 if quick else 5))
print(knobs, {k: round(v, 5) for k, v in res.items()})
