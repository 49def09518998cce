#!/usr/bin/env python
"""A/B of the prelude's lean FP64 math on LV N = 1e7 (every first-order callback is vector-issue bound there): for each setting of
(EXAHIP_FAST_EXP, EXAHIP_KTAB) a fresh model, min over 5 x 200 calls per callback (hipEvents inside exa_time_callback), and the
component-wise difference of hess_coord! / jac_coord! / cons_nln! to the first setting.  usage: math_ab.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ref = {}
for fe, kt in (("1", "0"), ("1", "1"), ("0", "0"), ("1", "0"), ("1", "1")):
    os.environ["EXAHIP_FAST_EXP"], os.environ["EXAHIP_KTAB"] = fe, kt
    m = ExaModel(models.luksan_vlcek_model(N))
    r = np.random.default_rng(0)
    x = torch.from_numpy(m.meta.x0 + 0.1 * r.uniform(-1, 1, N)).cuda()
    y = torch.from_numpy(r.standard_normal(m.meta.ncon)).cuda()
    v = torch.from_numpy(r.standard_normal(N)).cuda()
    c = torch.empty(m.meta.ncon, dtype=torch.float64, device="cuda")
    g = torch.empty(N, dtype=torch.float64, device="cuda")
    j = torch.empty(m.meta.nnzj, dtype=torch.float64, device="cuda")
    h = torch.empty(m.meta.nnzh, dtype=torch.float64, device="cuda")
    for _ in range(30):
        m.hess_coord(x, y, 0.5, out=h)
    t = {}
    t["cons"] = min(m.time_callback("cons", 200, x, out=c) for _ in range(5))
    t["jac"] = min(m.time_callback("jac", 200, x, out=j) for _ in range(5))
    t["hess"] = min(m.time_callback("hess", 200, x, y, 0.5, out=h) for _ in range(5))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, fn in (("jprod", lambda: m.jprod(x, v, out=c)), ("jtprod", lambda: m.jtprod(x, y, out=g)), ("hprod", lambda: m.hprod(x, y, v, 0.5, out=g))):
        best = 1e9
        for _ in range(5):
            fn(); torch.cuda.synchronize(); e0.record()
            for _ in range(100):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 100)
        t[name] = best
    out = {"cons": m.cons(x).cpu().numpy(), "jac": m.jac_coord(x).cpu().numpy(), "hess": m.hess_coord(x, y, 0.5).cpu().numpy()}
    if not ref:
        ref = out
    d = {k: float(np.max(np.abs(out[k] - ref[k]) / np.maximum(np.abs(ref[k]), 1e-300))) for k in out}
    print(f"FAST_EXP={fe} KTAB={kt}: " + "  ".join(f"{k} {v:.4f}" for k, v in t.items()) +
          "  | max rel diff to the first: " + "  ".join(f"{k} {v:.1e}" for k, v in d.items()), flush=True)
    del m
