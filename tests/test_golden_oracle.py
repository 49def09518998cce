"""Pins the TEST ORACLE (oracle/exa_oracle.c) against the golden vectors of tests/golden/ad_golden.json
(symbolic derivatives by sympy, 40-digit evaluation; generator: tests/golden/make_golden.py).

This is the in-container stand-in for the reference's own differential test against ForwardDiff
(test/ADTest/ADTest.jl:344-373).  CPU only.
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from exprs import EXPRS, NPAR, NVAR, SPECIAL_EXPRS  # noqa: E402

from exahip import ExaCore, graph, rng  # noqa: E402

with open(os.path.join(HERE, "golden", "ad_golden.json")) as fh:
    GOLD = json.load(fh)
CASES = {c["name"]: c for c in GOLD["cases"]}
X0, T0 = np.array(GOLD["x"]), np.array(GOLD["theta"])
# the SpecialFunctions rows (ext/functionlist.jl; ADTest.jl:59-120): same point, tests/golden/make_special_golden.py
with open(os.path.join(HERE, "golden", "special_golden.json")) as fh:
    SPECIAL_GOLD = json.load(fh)
assert SPECIAL_GOLD["x"] == GOLD["x"] and SPECIAL_GOLD["theta"] == GOLD["theta"]
CASES.update({c["name"]: c for c in SPECIAL_GOLD["cases"]})
ALL_EXPRS = EXPRS + SPECIAL_EXPRS

TOL = 1e-11   # oracle (glibc libm, double) vs 40-digit symbolic truth


class NodeF:
    """exahip namespace for exprs.py."""
    abs = staticmethod(abs)

    def __getattr__(self, k):
        return getattr(graph, k)


def close(a, ref, tol=TOL):
    a, ref = np.asarray(a, dtype=float), np.asarray(ref, dtype=float)
    scale = np.maximum(np.abs(ref), max(1.0, float(np.max(np.abs(ref)))) * 1e-3)
    return float(np.max(np.abs(a - ref) / scale)) <= tol


def dense_lower(rows, cols, vals, n):
    H = np.zeros((n, n))
    np.add.at(H, (rows - 1, cols - 1), vals)
    return H


def build_case(f, as_constraint):
    c = ExaCore()
    x = c.add_var(NVAR, start=X0)
    th = c.add_par(NPAR, value=T0)
    e = f(x, th, NodeF())
    if as_constraint:
        c.add_con(e)
    else:
        c.add_obj(e)
    return c


@pytest.mark.parametrize("name", [n for n, _ in ALL_EXPRS])
def test_oracle_matches_symbolic_derivatives(libs, name):
    import oracle
    f = dict(ALL_EXPRS)[name]
    g = CASES[name]
    gold_H = np.tril(np.array(g["hess"]))
    # as an objective: obj / grad! / hess_coord!(obj_weight)
    o = oracle.OracleModel(build_case(f, False).to_ir())
    assert close(o.obj(X0), g["value"])
    assert close(o.grad(X0), g["grad"])
    r, c = o.hess_structure()
    assert np.all(r >= c)
    H = dense_lower(r, c, o.hess_coord(X0, np.zeros(0), 1.0), NVAR)
    assert close(H, gold_H)
    # as a constraint row: cons! / jac_coord! / hess_coord!(y)
    o = oracle.OracleModel(build_case(f, True).to_ir())
    assert close(o.cons(X0), [g["value"]])
    jr, jc = o.jac_structure()
    J = np.zeros(NVAR)
    np.add.at(J, jc - 1, o.jac_coord(X0))
    assert np.all(jr == 1) and close(J, g["grad"])
    r, c = o.hess_structure()
    H = dense_lower(r, c, o.hess_coord(X0, np.array([-1.75]), 0.0), NVAR)
    assert close(H, -1.75 * gold_H)


def test_the_quad_arbiter_and_the_double_oracle_agree_within_the_running_error_bound(libs):
    """oracle/exa_quad.h: cons_nln! evaluated in __float128 with the first-order running error bound of a double-precision evaluation
    (Higham 3.3).  On every zoo model and on an ACOPF network whose power-balance rows cancel, the double oracle is within the bound of the
    quad value, row by row — what lets the GPU tests use the pair (1e-10 of the quad value, or K x the bound) as north_star's bar
    (tests/conftest.py parity_cons).  Also pins the arbiter: where nothing cancels, quad and double agree to a few ulp."""
    import oracle
    from exahip import ExaModel, models
    from zoo import ZOO, point
    eps = np.finfo(np.float64).eps
    cases = [(name, mk(), None) for name, mk in ZOO.items()]
    core = models.ac_power_model(models.synthetic_power_data(3000, 4500, 300, seed=2))
    cases.append(("acopf3000", core, models.acopf_start(core)))
    seen_quad = 0
    for name, core, x0 in cases:
        m = ExaModel(core, device=False)
        if m.meta.ncon == 0:
            continue
        o = oracle.OracleModel(m.ir)
        x = x0 if x0 is not None else point(m.meta.x0, m.meta.ncon)[0]
        quad = o.cons_quad(x)
        if quad is None:
            assert name == "specialfn"           # the SpecialFunctions extension has no quad restatement
            continue
        seen_quad += 1
        q, e = quad
        c = o.cons(x)
        d = np.abs(c - q)
        assert np.all(d <= 2.0 * eps * e + 1e-300), (name, float(np.max(d / np.maximum(eps * e, 1e-300))))
        # rows that do not cancel (the result is at least a thousandth of what was summed): a few ulp apart
        big = np.abs(q) * 1e3 >= e
        assert np.all(d[big] <= 1e-12 * np.abs(q[big])), name
    assert seen_quad >= 14
