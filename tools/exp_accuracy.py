#!/usr/bin/env python
"""Accuracy of the generated kernels' FP64 exp (exa_exp of the prelude; EXAHIP_FAST_EXP=0: ocml's) against 40-digit mpmath,
through the normal product path: a one-pattern model c_i = exp(x_i) evaluated on the GPU (value path) and its Jacobian
(derivative path: the same exponential).  Reports the max error in ulps over uniform, log-spaced and special arguments."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import mpmath  # noqa: E402
import numpy as np  # noqa: E402

from exahip import ExaCore, ExaModel, rng  # noqa: E402
from exahip.graph import exp  # noqa: E402

mpmath.mp.dps = 40
r = np.random.default_rng(0)
xs = np.concatenate([
    r.uniform(-1, 1, 6000), r.uniform(-40, 40, 6000), r.uniform(-708, 709.7, 6000),
    10 ** r.uniform(-12, 2.8, 3000) * r.choice([-1, 1], 3000),
    (np.arange(-1000, 1001) + 0.5) * np.log(2) * (1 + r.uniform(-1e-13, 1e-13, 2001)),     # near the rounding boundaries of k
    np.array([0.0, -0.0, 1e-300, -1e-300, 709.782712893384, 709.7827128933841, 710.0, 1e300, -745.0, -745.2, -746.0, -1e300, np.inf, -np.inf]),
])
n = len(xs)
c = ExaCore()
x = c.add_var(n)
c.add_con(lambda i: exp(x[i]), rng(1, n))
m = ExaModel(c)
val = m.cons(xs)
jac = m.jac_coord(xs)
nan = m.cons(np.full(n, np.nan))
assert np.all(np.isnan(nan)), "exp(NaN) must be NaN"
worst, wx, bad = 0.0, None, 0
with np.errstate(over="ignore"):
    ref64 = np.exp(xs)
for got in (val, jac):
    for g, xv, rf in zip(got, xs, ref64):
        if not np.isfinite(rf) or rf == 0.0 or rf < 2.3e-308:        # inf / 0 / subnormal results: compared exactly / by absolute error
            if not np.isfinite(rf) or rf == 0.0:
                bad += int(g != rf)
            else:
                bad += int(abs(g - rf) > 5e-324 * 2)
            continue
        t = mpmath.exp(mpmath.mpf(float(xv)))
        u = float(abs(mpmath.mpf(float(g)) - t) / mpmath.mpf(2) ** (mpmath.floor(mpmath.log(t, 2)) - 52))
        if u > worst:
            worst, wx = u, float(xv)
print(f"EXAHIP_FAST_EXP={os.environ.get('EXAHIP_FAST_EXP', '1')}: max ulp error of exp over {n} arguments (value and derivative paths): {worst:.3f} at x = {wx!r}; "
      f"special / subnormal results off: {bad}")
