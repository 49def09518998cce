#!/usr/bin/env python
"""A/B/A/B of one callback under several SETS of code-generator knobs in one process (timings of separate gpurun calls
differ by more than the effects being looked for), with a bitwise comparison of the outputs against the first variant.
usage: SWEEP_MODEL=lv SWEEP_N=3e7 ab_variants.py [--cb hess] name1:K=v,K2=v2 name2: ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

args = sys.argv[1:]
cb = "hess"
if args and args[0] == "--cb":
    cb = args[1]
    args = args[2:]
N = int(float(os.environ.get("SWEEP_N", "1e7")))
which = os.environ.get("SWEEP_MODEL", "lv")
reps = int(os.environ.get("SWEEP_REPS", "200" if N <= 2e7 else "40"))
core = {"lv": lambda: models.luksan_vlcek_model(N), "rocket": lambda: models.rocket_model(N if os.environ.get("SWEEP_N") else 1_000_000), "chain": lambda: models.cops_chain_model(N),
        "acopf": lambda: models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0))}[which]()
dev = torch.device("cuda:0")
runs = {}
x = y = None
ref = None
shared_out = None
for spec in args:
    name, _, kv = spec.partition(":")
    env = {k: v.replace("+", " ") for k, v in (p.split("=", 1) for p in kv.split(",") if p)}      # "+" stands for a blank
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = ExaModel(core)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if x is None:
        xh = models.acopf_start(core) if which == "acopf" else m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, m.meta.nvar)
        x = torch.from_numpy(xh).to(dev)
        y = torch.from_numpy(np.random.default_rng(1).standard_normal(m.meta.ncon)).to(dev)
    n_out = {"hess": m.meta.nnzh, "jac": m.meta.nnzj, "cons": m.meta.ncon, "grad": m.meta.nvar}[cb]
    # ONE output buffer for all variants: the same kernel differs by up to 9 % between two output buffers of one process
    # (physical placement), more than most of the effects looked for
    if shared_out is None:
        shared_out = torch.empty((n_out,), dtype=torch.float64, device=dev)
    out = shared_out
    out.fill_(float("nan"))
    m.time_callback(cb, 1, x, y, 0.5, out=out)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
        same = "ref"
    else:
        same = "bitwise-equal" if torch.equal(out, ref) else f"DIFFERS max|d|={float((out - ref).abs().max()):.3e} nan={int(torch.isnan(out).sum())}"
    m.time_callback(cb, 30, x, y, 0.5, out=out)
    runs[name] = (m, out, [], same)
for rnd in range(5):
    for name, (m, out, acc, _) in runs.items():
        acc.append(m.time_callback(cb, reps, x, y, 0.5, out=out))
print(f"== {which} N={N} {cb}")
for name, (m, out, acc, same) in runs.items():
    print(f"   {name:28s} min {min(acc):.5f}  med {float(np.median(acc)):.5f} ms   {same}", flush=True)
