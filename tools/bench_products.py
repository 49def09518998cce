#!/usr/bin/env python
"""jtprod / hprod: times both implementations (0 = atomics in the sweep, 1 = COO + sorted gather) on configs 2-4."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402


def run(name, core):
    m = ExaModel(core)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, m.meta.nvar)).to(dev)
    y = torch.from_numpy(np.random.default_rng(1).standard_normal(m.meta.ncon)).to(dev)
    v = torch.from_numpy(np.random.default_rng(2).standard_normal(m.meta.nvar)).to(dev)
    w = torch.from_numpy(np.random.default_rng(3).standard_normal(m.meta.ncon)).to(dev)
    out = torch.empty(m.meta.nvar, dtype=torch.float64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    for mode in (0, 1):
        m.set_product_mode(mode, mode)
        for nm, fn in (("jtprod", lambda: m.jtprod(x, w, out=out)), ("hprod", lambda: m.hprod(x, y, v, 0.5, out=out))):
            for _ in range(3):
                fn()
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[f"{nm}_mode{mode}_ms"] = round(e0.elapsed_time(e1) / 20, 4)
    m.set_product_mode(-1, -1)
    m.jtprod(x, w, out=out)
    m.hprod(x, y, v, 0.5, out=out)
    res["auto"] = m.product_mode()
    print(name, res, flush=True)


if __name__ == "__main__":
    run("lv1e7", models.luksan_vlcek_model(10_000_000))
    run("rocket1e6", models.rocket_model(1_000_000))
    run("acopf78k", models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0)))
