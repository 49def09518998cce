#!/usr/bin/env python
"""Every callback of LV at N points for ONE package directory (argv[1]: this tree's examodels.jl_amd or a copy of an older one), min over
5 x 100 calls by hipEvents — run once per package in a fresh process, alternating, by the caller: the same-box A/B of two trees.
usage: lv_callbacks_ab.py PKGDIR N [lv|rocket|acopf]   (N: points of LV / nh of the rocket; the ACOPF network is the 78 484-bus synthetic one)"""
# An older tree to compare with:  mkdir tools/_old && git archive <commit> examodels.jl_amd include | tar -x -C tools/_old && make -C tools/_old/examodels.jl_amd/csrc
# (round 5 used ccdf976 as tools/_old_r5 and 396dbc3 as tools/_old_r5b; both removed after the measurements)
import os
import sys

pkg, N = sys.argv[1], int(float(sys.argv[2]))
sys.path.insert(0, pkg)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

which = sys.argv[3] if len(sys.argv) > 3 else "lv"
core = {"lv": lambda: models.luksan_vlcek_model(N), "rocket": lambda: models.rocket_model(N),
        "acopf": lambda: models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0))}[which]()
m = ExaModel(core)
r = np.random.default_rng(0)
nv = m.meta.nvar
x = torch.from_numpy(np.asarray(m.meta.x0) + 0.05 * r.uniform(-1, 1, nv) * (1.0 if which == "lv" else 0.1)).cuda()
y = torch.from_numpy(r.standard_normal(m.meta.ncon)).cuda()
v = torch.from_numpy(r.standard_normal(nv)).cuda()
c = torch.empty(m.meta.ncon, dtype=torch.float64, device="cuda")
g = torch.empty(nv, dtype=torch.float64, device="cuda")
j = torch.empty(m.meta.nnzj, dtype=torch.float64, device="cuda")
h = torch.empty(m.meta.nnzh, dtype=torch.float64, device="cuda")
for _ in range(30):
    m.hess_coord(x, y, 0.5, out=h)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = {}
for name, fn in (("obj", lambda: m.obj(x)), ("cons", lambda: m.cons(x, out=c)), ("grad", lambda: m.grad(x, out=g)), ("jac", lambda: m.jac_coord(x, out=j)),
                 ("hess", lambda: m.hess_coord(x, y, 0.5, out=h)), ("jprod", lambda: m.jprod(x, v, out=c)), ("jtprod", lambda: m.jtprod(x, y, out=g)),
                 ("hprod", lambda: m.hprod(x, y, v, 0.5, out=g)), ("eval_all", lambda: m.eval_all(x, y, 0.5))):
    if name == "obj":
        continue
    best = 1e9
    for _ in range(5):
        fn(); torch.cuda.synchronize(); e0.record()
        for _ in range(100):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    t[name] = best
print(f"{os.path.relpath(pkg)} {which} N={N:.0e}: " + "  ".join(f"{k} {v:.4f}" for k, v in t.items()), flush=True)
