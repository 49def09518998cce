#!/usr/bin/env python
"""EXAHIP_TILE_LOOP A/B on LV: cons_nln! / jac_coord! by the one-tile kernels (0) against the pipelined tile-loop kernels exa_consl / exa_jacl with n tiles
per workgroup (unset = the library's rule).  One module, the launch differs.  One model per setting, the settings ALTERNATE in 7 rounds of 200 launches
(exa_time_callback: back to back from C — a Python call per launch is host-bound below ~0.05 ms), minimum per setting; outputs compared bit for bit with
the first setting's.  usage: tile_loop_ab.py [N] [n1,n2,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
settings = ("0", None) + tuple(sys.argv[2].split(",") if len(sys.argv) > 2 else ("2", "4", "8", "16"))
ms = {}
core = models.luksan_vlcek_model(N)
for tl in settings:
    os.environ.pop("EXAHIP_TILE_LOOP", None)
    if tl is not None:
        os.environ["EXAHIP_TILE_LOOP"] = tl
    ms[tl] = ExaModel(core)
m0 = ms["0"]
r = np.random.default_rng(0)
x = torch.from_numpy(m0.meta.x0 + 0.1 * r.uniform(-1, 1, N)).cuda()
c = torch.empty(m0.meta.ncon, dtype=torch.float64, device="cuda")
j = torch.empty(m0.meta.nnzj, dtype=torch.float64, device="cuda")
same, ref = {}, None
for tl in settings:
    c.fill_(float("nan")); j.fill_(float("nan"))
    ms[tl].cons(x, out=c); ms[tl].jac_coord(x, out=j)
    torch.cuda.synchronize()
    if ref is None:
        ref = (c.clone(), j.clone())
    same[tl] = torch.equal(c, ref[0]) and torch.equal(j, ref[1])
t = {tl: {"cons": [], "jac": []} for tl in settings}
reps = 200 if N <= 3e7 else 40
for rnd in range(7):
    for tl in settings:
        for name, buf in (("cons", c), ("jac", j)):
            v = ms[tl].time_callback(name, reps, x, out=buf)
            if rnd:
                t[tl][name].append(v)
for tl in settings:
    print(f"N={N:.0e} EXAHIP_TILE_LOOP={'unset' if tl is None else tl:5s}: " + "  ".join(f"{k} min {min(v):.4f} median {sorted(v)[3]:.4f}" for k, v in t[tl].items())
          + f"  bitwise equal to the first: {same[tl]}", flush=True)
