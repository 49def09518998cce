#!/usr/bin/env python
"""Generates tests/golden/special_golden.json: value, gradient and dense Hessian of every SPECIAL_EXPRS row of exprs.py (the
SpecialFunctions rows of the reference's AD test list, test/ADTest/ADTest.jl:59-120, and one row per entry of
ext/functionlist.jl) at the evaluation point of ad_golden.json, by SYMBOLIC differentiation (sympy; erfcx, dawson and
invdigamma are defined here from erfc / erfi / the implicit function theorem) evaluated with 40-digit mpmath.

    python tests/golden/make_special_golden.py

Independent of both the C test oracle (oracle/exa_special.h) and the HIP routines (csrc/exa_gen_prelude.cpp)."""
import json
import os
import sys

import mpmath
import sympy as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from exprs import NPAR, NVAR, SPECIAL_EXPRS  # noqa: E402
from make_golden import OneBased, SymF  # noqa: E402

mpmath.mp.dps = 40


class invdigamma(sp.Function):
    """The positive solution t of digamma(t) = y (SpecialFunctions.invdigamma); dt/dy = 1 / trigamma(t)."""
    nargs = 1

    def fdiff(self, argindex=1):
        return 1 / sp.polygamma(1, invdigamma(self.args[0]))

    def _eval_evalf(self, prec):
        y = self.args[0]._to_mpmath(prec)
        with mpmath.workprec(prec + 20):
            t0 = mpmath.exp(y) + mpmath.mpf("0.5") if y >= -2.22 else -1 / (y - mpmath.digamma(1))
            t = mpmath.findroot(lambda t: mpmath.digamma(t) - y, t0)
        return sp.Float(t, precision=prec)


class SpecF(SymF):
    erf, erfc, erfi = staticmethod(sp.erf), staticmethod(sp.erfc), staticmethod(sp.erfi)
    erfcx = staticmethod(lambda a: sp.exp(a ** 2) * sp.erfc(a))
    digamma = staticmethod(lambda a: sp.polygamma(0, a))
    trigamma = staticmethod(lambda a: sp.polygamma(1, a))
    invdigamma = staticmethod(invdigamma)
    gamma = staticmethod(sp.gamma)
    airyai, airybi = staticmethod(sp.airyai), staticmethod(sp.airybi)
    airyaiprime, airybiprime = staticmethod(sp.airyaiprime), staticmethod(sp.airybiprime)
    besselj0 = staticmethod(lambda a: sp.besselj(0, a))
    bessely0 = staticmethod(lambda a: sp.bessely(0, a))
    besselj1 = staticmethod(lambda a: sp.besselj(1, a))
    bessely1 = staticmethod(lambda a: sp.bessely(1, a))
    dawson = staticmethod(lambda a: sp.sqrt(sp.pi) / 2 * sp.exp(-a ** 2) * sp.erfi(a))
    erfinv = staticmethod(sp.erfinv)
    erfcinv = staticmethod(lambda a: sp.erfinv(1 - a))
    # beta through gamma (valid at negative non-integer arguments too); logbeta = log|beta|
    beta = staticmethod(lambda a, b: sp.gamma(a) * sp.gamma(b) / sp.gamma(a + b))
    logbeta = staticmethod(lambda a, b: sp.log((sp.gamma(a) * sp.gamma(b) / sp.gamma(a + b)) ** 2) / 2)      # log|u|, derivative u'/u


def main():
    with open(os.path.join(HERE, "ad_golden.json")) as fh:
        base = json.load(fh)
    x0, t0 = base["x"], base["theta"]
    xs = sp.symbols(f"x1:{NVAR + 1}", real=True)
    ts = sp.symbols(f"t1:{NPAR + 1}", real=True)
    out = {"nvar": NVAR, "npar": NPAR, "x": x0, "theta": t0, "cases": []}
    subs = {s: sp.Float(repr(float(v)), 50) for s, v in zip(xs, x0)}
    subs.update({s: sp.Float(repr(float(v)), 50) for s, v in zip(ts, t0)})
    for name, f in SPECIAL_EXPRS:
        e = sp.sympify(f(OneBased(xs), OneBased(ts), SpecF))

        def ev(expr):
            if expr == 0:
                return 0.0
            v = expr.subs(subs)
            v = v.replace(lambda a: isinstance(a, sp.Derivative) and isinstance(a.expr, (sp.Abs, sp.sign)), lambda a: sp.Integer(0))
            return float(sp.N(v, 40))

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
        print(f"{name:24s} value={out['cases'][-1]['value']: .6e}  |grad|={max(abs(g) for g in grad):.3e}", flush=True)
    with open(os.path.join(HERE, "special_golden.json"), "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
