#!/usr/bin/env python
"""hess_coord! of one benchmark model for ONE package directory (this tree's examodels.jl_amd or a copy of an older one: see tools/lv_callbacks_ab.py for how
to make one), by exa_time_callback (launches back to back from C), min and median over 8 x 200 — run alternately, a fresh process each, by the caller.
usage: model_hess_ab.py PKGDIR rocket|lv|acopf N"""
import os
import sys

pkg, which, N = sys.argv[1], sys.argv[2], int(float(sys.argv[3]))
sys.path.insert(0, pkg)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

core = {"lv": lambda: models.luksan_vlcek_model(N), "rocket": lambda: models.rocket_model(N),
        "acopf": lambda: models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0))}[which]()
m = ExaModel(core)
r = np.random.default_rng(0)
x = torch.from_numpy(np.asarray(m.meta.x0) + 0.005 * r.uniform(-1, 1, m.meta.nvar)).cuda()
y = torch.from_numpy(r.standard_normal(m.meta.ncon)).cuda()
h = torch.empty(m.meta.nnzh, dtype=torch.float64, device="cuda")
reps = 200 if m.meta.nnzh < 3e8 else 30
m.time_callback("hess", reps, x, y, 0.5, out=h)
ts = sorted(m.time_callback("hess", reps, x, y, 0.5, out=h) for _ in range(8))
print(f"{os.path.relpath(pkg):40s} {which} N={N:.0e}: hess min {ts[0]:.4f} median {ts[4]:.4f} ms  sum {float(h.sum()):.12e}", flush=True)
