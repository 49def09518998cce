#!/usr/bin/env python
"""obj of LV at three sizes (one / two-level arrival fold): value against numpy, run-to-run equality of five calls, ms per call by exa_time_callback.
usage (GPU box, repo root): python tools/obj_time.py"""
import sys
sys.path[:0]=["examodels.jl_amd"]
import numpy as np, torch
from exahip import ExaModel, models
for N in (1_000_000, 10_000_000, 100_000_000):
    m = ExaModel(models.luksan_vlcek_model(N))
    r = np.random.default_rng(0)
    xh = m.meta.x0 + 0.1 * r.uniform(-1, 1, N)
    x = torch.from_numpy(xh).cuda()
    vals = [m.obj(x) for _ in range(5)]
    ref = float(np.sum(100*(xh[:-1]**2 - xh[1:])**2 + (xh[:-1]-1)**2))
    ts = [m.time_callback("obj", 200, x) for _ in range(5)]
    print(N, "obj", vals[0], "rel err vs numpy", abs(vals[0]-ref)/abs(ref), "all equal", len(set(vals))==1, "ms", min(ts), flush=True)
