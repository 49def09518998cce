#!/usr/bin/env python
"""Generates tests/golden/zoo_fixtures/<model>.{json,bin}: for every model of the test zoo and for Luksan-Vlcek N = 1e4
(BASELINE.json configs[0]) the evaluation point (x, y, sigma, u, v) and what the seven NLPModels callbacks return there
— obj, cons, grad, jac_structure + jac_coord, hess_structure + hess_coord, jprod, jtprod, hprod — computed by the TEST
ORACLE (oracle/exa_oracle.c, the CPU restatement of the reference's recurrences).

Three consumers:
  * tests/test_golden_zoo.py (-m "not gpu"): the oracle still reproduces them (a regression pin of the restatement);
  * tests/test_golden_zoo.py (-m gpu): the HIP path through the C ABI reproduces them (structure ==, values 1e-10);
  * tools/reference_check.jl: on any machine with Julia + ExaModels v0.12 the SAME files are compared with the real
    reference (backend = nothing): structure ==, values 1e-10 — the one command that turns "slot order pinned only by two
    re-readings of hessian.jl" into "confirmed by a Julia process".  The models are re-stated there with the reference's
    own macros; the ACOPF tables travel inside the fixture.

Format: <model>.bin = the arrays back to back, little endian; <model>.json = {"scalars": {...}, "arrays": {name:
[dtype, count, byte offset]}} with dtype in {"f8", "i4", "i8"}.  Run from the repo root: python tests/golden/make_zoo_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "examodels.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

OUT = os.path.join(HERE, "zoo_fixtures")


def models():
    from exahip import models as M
    from zoo import ZOO
    out = dict(ZOO)
    out["lv10000"] = lambda: M.luksan_vlcek_model(10_000)          # BASELINE.json configs[0]
    return out


def evaluate(core):
    import oracle
    from zoo import point
    ir = core.to_ir()
    o = oracle.OracleModel(ir)
    x, y, sigma = point(ir.x0, o.ncon, seed=0)
    u = np.random.default_rng(2).standard_normal(o.nvar)
    v = np.random.default_rng(3).standard_normal(o.ncon)
    jr, jc = o.jac_structure()
    hr, hc = o.hess_structure()
    arrays = {
        "x": x, "y": y, "u": u, "v": v,
        "cons": o.cons(x), "grad": o.grad(x),
        "jac_rows": jr.astype(np.int32), "jac_cols": jc.astype(np.int32), "jac_vals": o.jac_coord(x),
        "hess_rows": hr.astype(np.int32), "hess_cols": hc.astype(np.int32), "hess_vals": o.hess_coord(x, y, sigma),
        "jprod": o.jprod(x, u), "jtprod": o.jtprod(x, v), "hprod": o.hprod(x, y, u, sigma),
    }
    scalars = {"nvar": int(o.nvar), "ncon": int(o.ncon), "nnzj": int(o.nnzj), "nnzh": int(o.nnzh), "sigma": float(sigma),
               "obj": float(o.obj(x)), "minimize": bool(ir.desc.minimize)}
    return scalars, arrays


def write(name, scalars, arrays):
    man = {"scalars": scalars, "arrays": {}}
    off = 0
    with open(os.path.join(OUT, name + ".bin"), "wb") as fh:
        for k, a in arrays.items():
            a = np.ascontiguousarray(a)
            dt = {"float64": "f8", "int32": "i4", "int64": "i8"}[a.dtype.name]
            fh.write(a.astype("<" + dt).tobytes())
            man["arrays"][k] = [dt, int(a.size), off]
            off += a.nbytes
    with open(os.path.join(OUT, name + ".json"), "w") as fh:
        json.dump(man, fh, indent=1)


def load(name):
    with open(os.path.join(OUT, name + ".json")) as fh:
        man = json.load(fh)
    raw = open(os.path.join(OUT, name + ".bin"), "rb").read()
    arrays = {k: np.frombuffer(raw, dtype="<" + dt, count=n, offset=off) for k, (dt, n, off) in man["arrays"].items()}
    return man["scalars"], arrays


def acopf_tables():
    """The synthetic tables of zoo 'acopf30' (exahip.models.synthetic_power_data(30, 41, 6, seed=3)) as flat arrays, so
    that the Julia script can rebuild exactly this instance."""
    from exahip import models as M
    d = M.synthetic_power_data(nbus=30, nbr=41, ngen=6, seed=3)
    out = {}
    for tab in ("bus", "gen", "arc", "branch"):
        for col, vals in d[tab].cols.items():
            out[f"data_{tab}_{col}"] = np.asarray(vals)
    for k in ("ref_buses", "vmax", "vmin", "pmax", "pmin", "qmax", "qmin", "rate_a", "angmax", "angmin"):
        out["data_" + k] = np.asarray(d[k])
    return out


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, mk in models().items():
        scalars, arrays = evaluate(mk())
        if name == "acopf30":
            arrays.update(acopf_tables())
        write(name, scalars, arrays)
        print(f"{name:18s} nvar {scalars['nvar']:6d} ncon {scalars['ncon']:6d} nnzj {scalars['nnzj']:7d} nnzh {scalars['nnzh']:7d}")
