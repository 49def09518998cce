#!/usr/bin/env python
"""What bounds the first-order callbacks of a stencil model?  cons_nln! / jac_coord! at N = 1e7 for constraint bodies of growing arithmetic on
the SAME access skeleton (one pattern over 1..N-2): 2 x[i] (one point) -> x[i] + 2 x[i+1] + 3 x[i+2] (the LV stencil, no functions) -> + exp ->
+ exp + 2 sin (LV's mix) -> the LV constraint itself.  Same bytes in and out from the second row on: the differences are arithmetic, the first
rows are the skeleton (launch, block map, loads, store).  ms per call, min over 5 x 100 calls."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaCore, ExaModel, models, rng  # noqa: E402
from exahip.graph import exp, sin  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
bodies = {
    "2 x[i]": lambda x: (lambda i: 2.0 * x[i]),
    "x[i] + 2 x[i+1] + 3 x[i+2]": lambda x: (lambda i: x[i] + 2.0 * x[i + 1] + 3.0 * x[i + 2]),
    "... + x[i] exp(x[i] - x[i+1])": lambda x: (lambda i: x[i] + 2.0 * x[i + 1] + 3.0 * x[i + 2] + x[i] * exp(x[i] - x[i + 1])),
    "... + exp + sin(a) sin(b)": lambda x: (lambda i: x[i] + 2.0 * x[i + 1] + 3.0 * x[i + 2] + x[i] * exp(x[i] - x[i + 1]) + sin(x[i + 1] - x[i + 2]) * sin(x[i + 1] + x[i + 2])),
}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def best(fn):
    b = 1e9
    for _ in range(5):
        fn(); torch.cuda.synchronize(); e0.record()
        for _ in range(100):
            fn()
        e1.record(); torch.cuda.synchronize()
        b = min(b, e0.elapsed_time(e1) / 100)
    return b


def run(name, core):
    m = ExaModel(core)
    x = torch.from_numpy(np.asarray(m.meta.x0) + 0.1 * np.random.default_rng(0).uniform(-1, 1, m.meta.nvar)).cuda()
    c = torch.empty(m.meta.ncon, dtype=torch.float64, device="cuda")
    j = torch.empty(m.meta.nnzj, dtype=torch.float64, device="cuda")
    for _ in range(20):
        m.jac_coord(x, out=j)
    tc, tj = best(lambda: m.cons(x, out=c)), best(lambda: m.jac_coord(x, out=j))
    audit = {a["kernel"]: a["vgpr"] for a in m.build_audit() if a["kernel"] in ("exa_cons", "exa_cons1", "exa_jac")}
    print(f"{name:34s} cons_nln! {tc:.4f} ms ({8 * (m.meta.nvar + m.meta.ncon) / tc / 1e6:5.0f} GB/s)   jac_coord! {tj:.4f} ms ({8 * (m.meta.nvar + m.meta.nnzj) / tj / 1e6:5.0f} GB/s, nnzj {m.meta.nnzj})  vgpr {audit}", flush=True)


for name, mk in bodies.items():
    c = ExaCore()
    x = c.add_var(N, start=0.5)
    c.add_con(mk(x), rng(1, N - 2))
    run(name, c)
run("LV (objective + constraint)", models.luksan_vlcek_model(N))
