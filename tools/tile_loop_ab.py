#!/usr/bin/env python
"""EXAHIP_TILE_LOOP A/B on LV: cons_nln! / jac_coord! by the one-tile kernels (0) against the tile-loop kernels exa_consl / exa_jacl with n tiles per
workgroup (unset = the library's rule: 4 / 8 where the block map is long).  One module, the launch differs.  ms per call, min over 5 x 100 calls;
outputs compared bit for bit with the first setting.  usage: tile_loop_ab.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ref = None
for tl in ("0", None, "2", "4", "8", "16", "0", None):
    os.environ.pop("EXAHIP_TILE_LOOP", None)
    if tl is not None:
        os.environ["EXAHIP_TILE_LOOP"] = tl
    m = ExaModel(models.luksan_vlcek_model(N))
    r = np.random.default_rng(0)
    x = torch.from_numpy(m.meta.x0 + 0.1 * r.uniform(-1, 1, N)).cuda()
    y = torch.from_numpy(r.standard_normal(m.meta.ncon)).cuda()
    c = torch.empty(m.meta.ncon, dtype=torch.float64, device="cuda")
    j = torch.empty(m.meta.nnzj, dtype=torch.float64, device="cuda")
    h = torch.empty(m.meta.nnzh, dtype=torch.float64, device="cuda")
    for _ in range(30):
        m.hess_coord(x, y, 0.5, out=h)
    t = {}
    for name, fn in (("cons", lambda: m.cons(x, out=c)), ("jac", lambda: m.jac_coord(x, out=j))):
        best = 1e9
        for _ in range(5):
            fn(); torch.cuda.synchronize(); e0.record()
            for _ in range(100):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 100)
        t[name] = best
    out = (c.clone(), j.clone())
    if ref is None:
        ref = out
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    print(f"N={N:.0e} EXAHIP_TILE_LOOP={'unset' if tl is None else tl}: " + "  ".join(f"{k} {v:.4f}" for k, v in t.items()) + f"  bitwise equal to the first: {same}", flush=True)
    del m
