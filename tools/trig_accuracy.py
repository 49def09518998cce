#!/usr/bin/env python
"""Accuracy of the generated kernels' FP64 sin/cos (exa_sincos fast path + ocml fallback) against 40-digit mpmath,
through the normal product path: a one-pattern model c_i = sin(x_i) + 2 cos(x_i) evaluated on the GPU; jac gives
cos(x_i) - 2 sin(x_i).  Reports the max error in ulps over log-spaced and near-multiple-of-pi/2 arguments."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import mpmath  # noqa: E402
import numpy as np  # noqa: E402

from exahip import ExaCore, ExaModel, rng  # noqa: E402
from exahip.graph import cos, sin  # noqa: E402

mpmath.mp.dps = 40
r = np.random.default_rng(0)
xs = np.concatenate([
    r.uniform(-10, 10, 4000), 10 ** r.uniform(-8, 5.9, 4000) * r.choice([-1, 1], 4000),
    (np.arange(1, 2001) * (np.pi / 2)) * (1 + r.uniform(-1e-12, 1e-12, 2000)),        # near multiples of pi/2
    np.array([0.0, 1e-300, 823549.0, 823550.0, 1e6, 1e15, 3.0e20]),
])
n = len(xs)
c = ExaCore()
x = c.add_var(n)
c.add_con(lambda i: sin(x[i]), rng(1, n))
c.add_con(lambda i: cos(x[i]), rng(1, n))
m = ExaModel(c)
val = m.cons(xs)
jac = m.jac_coord(xs)          # [cos x_i ...][-sin x_i ...]
S, C = val[:n], val[n:]
Sj, Cj = -jac[n:], jac[:n]


def ulps(got, fn):
    worst = 0.0
    for g, xv in zip(got, xs):
        t = fn(mpmath.mpf(float(xv)))
        if t == 0:
            continue
        u = abs(mpmath.mpf(float(g)) - t) / mpmath.mpf(2) ** (mpmath.floor(mpmath.log(abs(t), 2)) - 52)
        worst = max(worst, float(u))
    return worst


print("max ulp error  sin (value path):", ulps(S, mpmath.sin), " cos:", ulps(C, mpmath.cos))
print("max ulp error  sin (sincos path):", ulps(Sj, mpmath.sin), " cos:", ulps(Cj, mpmath.cos))
