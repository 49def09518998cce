#!/usr/bin/env python
"""Generates tests/golden/ad_golden.json: value, gradient and dense Hessian of every expression in exprs.py at a
fixed pseudo-random point, by SYMBOLIC differentiation (sympy) evaluated with 40-digit mpmath.

    python tests/golden/make_golden.py

The reference cannot be executed in this container (pure Julia, no julia binary), so these vectors play the role
ForwardDiff plays in the reference's own AD tests (test/ADTest/ADTest.jl:344-373): an independent differentiator.
"""
import json
import os
import sys

import mpmath
import numpy as np
import sympy as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from exprs import EXPRS, NPAR, NVAR  # noqa: E402

mpmath.mp.dps = 40
D2R = sp.pi / 180


class SymF:
    """sympy namespace: every function defined from elementary sympy functions only."""
    inv = staticmethod(lambda a: 1 / a)
    abs = staticmethod(sp.Abs)
    sqrt = staticmethod(sp.sqrt)
    cbrt = staticmethod(lambda a: a ** sp.Rational(1, 3))
    abs2 = staticmethod(lambda a: a ** 2)
    sign = staticmethod(sp.sign)
    exp = staticmethod(sp.exp)
    exp2 = staticmethod(lambda a: 2 ** a)
    exp10 = staticmethod(lambda a: 10 ** a)
    expm1 = staticmethod(lambda a: sp.exp(a) - 1)
    log = staticmethod(sp.log)
    log2 = staticmethod(lambda a: sp.log(a) / sp.log(2))
    log1p = staticmethod(lambda a: sp.log(1 + a))
    log10 = staticmethod(lambda a: sp.log(a) / sp.log(10))
    sin, cos, tan = staticmethod(sp.sin), staticmethod(sp.cos), staticmethod(sp.tan)
    asin, acos = staticmethod(sp.asin), staticmethod(sp.acos)
    acot = staticmethod(lambda a: sp.atan(1 / a))
    csc = staticmethod(lambda a: 1 / sp.sin(a))
    sec = staticmethod(lambda a: 1 / sp.cos(a))
    cot = staticmethod(lambda a: 1 / sp.tan(a))
    sinh, cosh, tanh = staticmethod(sp.sinh), staticmethod(sp.cosh), staticmethod(sp.tanh)
    asinh, acosh, atanh = staticmethod(sp.asinh), staticmethod(sp.acosh), staticmethod(sp.atanh)
    acoth = staticmethod(lambda a: sp.atanh(1 / a))
    csch = staticmethod(lambda a: 1 / sp.sinh(a))
    sech = staticmethod(lambda a: 1 / sp.cosh(a))
    coth = staticmethod(lambda a: 1 / sp.tanh(a))
    sind = staticmethod(lambda a: sp.sin(a * D2R))
    cosd = staticmethod(lambda a: sp.cos(a * D2R))
    tand = staticmethod(lambda a: sp.tan(a * D2R))
    cscd = staticmethod(lambda a: 1 / sp.sin(a * D2R))
    secd = staticmethod(lambda a: 1 / sp.cos(a * D2R))
    cotd = staticmethod(lambda a: 1 / sp.tan(a * D2R))
    atand = staticmethod(lambda a: sp.atan(a) / D2R)
    acotd = staticmethod(lambda a: sp.atan(1 / a) / D2R)
    sinpi = staticmethod(lambda a: sp.sin(sp.pi * a))
    cospi = staticmethod(lambda a: sp.cos(sp.pi * a))
    sinc = staticmethod(lambda a: sp.sin(sp.pi * a) / (sp.pi * a))
    deg2rad = staticmethod(lambda a: a * D2R)
    rad2deg = staticmethod(lambda a: a / D2R)
    hypot = staticmethod(lambda a, b: sp.sqrt(a ** 2 + b ** 2))
    floor = staticmethod(sp.floor)
    ceil = staticmethod(sp.ceiling)

    @staticmethod
    def atan(a, b=None):
        return sp.atan(a) if b is None else sp.atan2(a, b)


class OneBased:
    def __init__(self, items):
        self.items = list(items)

    def __getitem__(self, i):
        return self.items[i - 1]


def main():
    rng = np.random.default_rng(20260928)
    xs = sp.symbols(f"x1:{NVAR + 1}", real=True)
    ts = sp.symbols(f"t1:{NPAR + 1}", real=True)
    out = {"nvar": NVAR, "npar": NPAR, "cases": []}
    x0 = rng.uniform(0.1, 0.9, NVAR)     # one common evaluation point (all rows are in-domain on (0.1, 0.9))
    t0 = rng.uniform(0.5, 1.5, NPAR)
    out["x"], out["theta"] = x0.tolist(), t0.tolist()
    for name, f in EXPRS:
        subs = {s: mpmath.mpf(float(v)) for s, v in zip(xs, x0)}
        subs.update({s: mpmath.mpf(float(v)) for s, v in zip(ts, t0)})

        class F(SymF):
            # max / min: take the branch that is active at the evaluation point (strict comparisons,
            # src/functionlist.jl:79-80)
            @staticmethod
            def maximum(a, b):
                return a if sp.N(a.subs(subs), 30) > sp.N(b.subs(subs), 30) else b

            @staticmethod
            def minimum(a, b):
                return a if sp.N(a.subs(subs), 30) < sp.N(b.subs(subs), 30) else b

        e = sp.sympify(f(OneBased(xs), OneBased(ts), F))

        def ev(expr):
            if expr == 0:
                return 0.0
            v = sp.N(expr.subs(subs), 40)
            # floor/ceil/sign/Abs derivatives are zero/one away from their kinks
            v = v.replace(lambda a: isinstance(a, (sp.Derivative, sp.Subs)), lambda a: sp.Integer(0))
            v = sp.N(v, 40)
            return float(v)

        used = sorted((s for s in e.free_symbols if s in xs), key=lambda s: xs.index(s))
        grad = [0.0] * NVAR
        hess = [[0.0] * NVAR for _ in range(NVAR)]
        g_sym = {s: sp.diff(e, s) for s in used}
        for s in used:
            grad[xs.index(s)] = ev(g_sym[s])
        for a in used:
            for b in used:
                if xs.index(b) <= xs.index(a):
                    h = ev(sp.diff(g_sym[a], b))
                    hess[xs.index(a)][xs.index(b)] = h
                    hess[xs.index(b)][xs.index(a)] = h
        out["cases"].append({"name": name, "value": ev(e), "grad": grad, "hess": hess})
        print(f"{name:24s} value={out['cases'][-1]['value']: .6e}")
    with open(os.path.join(HERE, "ad_golden.json"), "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
