#!/usr/bin/env python
"""Writes the generated HIP source of every zoo model (and the benchmark models) to a directory, plan-only (no device):
`diff -r` of two such dumps shows exactly what a generator change did to the kernels.  usage: dump_sources.py OUTDIR"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("examodels.jl_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from exahip import ExaModel, models  # noqa: E402
from zoo import ZOO  # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
cores = dict(ZOO)
cores["rocket1e4"] = lambda: models.rocket_model(10_000)
cores["acopf_big"] = lambda: models.ac_power_model(models.synthetic_power_data(2000, 3000, 300, seed=0))
for name, mk in cores.items():
    m = ExaModel(mk(), device=False)
    with open(os.path.join(out, name + ".hip"), "w") as fh:
        fh.write(m.kernel_source())
print(len(cores), "sources in", out)
