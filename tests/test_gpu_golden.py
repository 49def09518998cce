"""-m gpu: the HIP path against (a) the symbolic golden vectors and (b) the reference's known-answer tests.

All golden expressions are compiled into ONE model (one fused launch per callback): expression k becomes a constraint
pattern over 3 data points, data point i using its own copy of the 10 variables through SYMBOLIC indices
x[(i-1)*10 + k] — exercising index arithmetic, per-pattern COO offsets and the block->pattern dispatch at once.
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from conftest import has_gpu  # noqa: E402
from exprs import EXPRS, NPAR, NVAR, SPECIAL_EXPRS  # noqa: E402
from test_golden_oracle import CASES, T0, X0, NodeF, close, dense_lower  # noqa: E402
from test_known_answers import CONS_CASES, LSTAR, XSTAR, check_ipopt_logs, doc_parametric_lv10  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

NCOPY = 3
TOL = 1e-10


class Shifted:
    def __init__(self, X, i):
        self.X, self.i = X, i

    def __getitem__(self, k):
        return self.X[(self.i - 1) * NVAR + k]


def one_model(rows):
    from exahip import ExaCore, ExaModel, rng
    c = ExaCore()
    X = c.add_var(NVAR * NCOPY, start=np.tile(X0, NCOPY))
    th = c.add_par(NPAR, value=T0)
    for name, f in rows:
        c.add_con(lambda i, f=f: f(Shifted(X, i), th, NodeF()), rng(1, NCOPY))
    return ExaModel(c)


@pytest.fixture(scope="module")
def golden_model(libs):
    return one_model(EXPRS)


@pytest.fixture(scope="module")
def special_model(libs):
    """The SpecialFunctions rows (ext/functionlist.jl, ADTest.jl:59-120) in a model of their own: its module carries the special
    prelude (csrc/exa_gen_prelude.cpp kSpecialPrelude), the other one does not."""
    return one_model(SPECIAL_EXPRS)


def test_all_special_function_rows_in_one_model(special_model):
    check_rows(special_model, SPECIAL_EXPRS)


def test_all_golden_expressions_in_one_model(golden_model):
    check_rows(golden_model, EXPRS)


def check_rows(m, EXPRS):
    x = np.tile(X0, NCOPY)
    y = np.ones(m.meta.ncon)
    cons = m.cons(x)
    jr, jc = m.jac_structure()
    jv = m.jac_coord(x)
    hr, hc = m.hess_structure()
    hv = m.hess_coord(x, y, 0.0)
    assert np.all(hr >= hc)
    bad = []
    for k, (name, _) in enumerate(EXPRS):
        g = CASES[name]
        info = m.pattern_info(k)
        for i in range(NCOPY):
            row = info["o0"] + i
            ok = close([cons[row]], [g["value"]], TOL)
            s = slice(info["o1"] + info["o1step"] * i, info["o1"] + info["o1step"] * (i + 1))
            assert np.all(jr[s] == row + 1)
            J = np.zeros(NVAR * NCOPY)
            np.add.at(J, jc[s] - 1, jv[s])
            gold_g = np.zeros(NVAR * NCOPY)
            gold_g[i * NVAR:(i + 1) * NVAR] = g["grad"]
            ok &= close(J, gold_g, TOL)
            s = slice(info["o2"] + info["o2step"] * i, info["o2"] + info["o2step"] * (i + 1))
            H = dense_lower(hr[s], hc[s], hv[s], NVAR * NCOPY)
            gold_H = np.zeros((NVAR * NCOPY, NVAR * NCOPY))
            gold_H[i * NVAR:(i + 1) * NVAR, i * NVAR:(i + 1) * NVAR] = np.tril(np.array(g["hess"]))
            ok &= close(H, gold_H, TOL)
            if not ok:
                bad.append((name, i))
    assert not bad, bad


@pytest.mark.parametrize("name", list(CONS_CASES))
def test_reference_known_answers_on_hip(libs, name):
    from exahip import ExaModel
    core, x0, expected = CONS_CASES[name]()
    m = ExaModel(core)
    np.testing.assert_allclose(m.cons(x0), expected, rtol=0, atol=1e-13)


def test_lv10_published_kkt_point_on_hip(libs):
    from exahip import ExaModel, models
    m = ExaModel(models.luksan_vlcek_model(10))
    assert np.max(np.abs(m.cons(XSTAR))) < 1e-8
    jr, jc = m.jac_structure()
    J = np.zeros((m.meta.ncon, m.meta.nvar))
    np.add.at(J, (jr - 1, jc - 1), m.jac_coord(XSTAR))
    assert np.max(np.abs(m.grad(XSTAR) + J.T @ LSTAR)) < 1e-6


def test_reference_ipopt_logs_of_the_parametric_lv10_on_hip(libs):
    """docs/src/parameters.md:95-295 on the HIP path: nnzj = 24 / nnzh = 75, obj(x0) for the three parameter sets (exa_set_value
    between them, no rebuild), the six full-step Newton iterates of each log to the printed digits (each iterate is a function of
    hess_coord!'s VALUES at the one before), the three optima."""
    from exahip import ExaModel
    core, th = doc_parametric_lv10()
    m = ExaModel(core)
    x0 = np.array(m.meta.x0)
    assert m.obj(x0) == 2057.0
    assert "%.2e" % np.max(np.abs(m.cons(x0))) == "2.48e+01"
    check_ipopt_logs(m, x0, lambda v: m.set_value(th, v))
    m.set_value(th, [200.0, 1.0])
    assert abs(m.obj(x0) - 4089.8) < 1e-9
    m.set_value(th, [200.0, 0.5])
    assert abs(m.obj(x0) - 4081.05) < 1e-9


def test_objective_callbacks_on_golden_rows(libs):
    """A few rows as objectives: obj / grad! / hess_coord!(obj_weight) (exercises the atomic grad scatter)."""
    from exahip import ExaCore, ExaModel
    for name in ("lv-obj", "composite-1-5", "pow-negint", "table-atan2", "rocket-vel", "composite-1-1", "composite-1-14", "parameter-composite-4",
                 "sf-airyai", "sf-invdigamma", "sf-bessely1"):
        f = dict(EXPRS + SPECIAL_EXPRS)[name]
        c = ExaCore()
        x = c.add_var(NVAR, start=X0)
        th = c.add_par(NPAR, value=T0)
        c.add_obj(f(x, th, NodeF()))
        m = ExaModel(c)
        g = CASES[name]
        assert close([m.obj(X0)], [g["value"]], TOL), name
        assert close(m.grad(X0), g["grad"], TOL), name
        r, cc = m.hess_structure()
        H = dense_lower(r, cc, m.hess_coord(X0, np.zeros(0), 2.5), NVAR)
        assert close(H, 2.5 * np.tril(np.array(g["hess"])), TOL), name


def test_fast_sincos_accuracy_in_ulps(libs):
    """The generated kernels use a lean FP64 sincos (3-term Cody-Waite + fdlibm kernels, ocml beyond 8e5): <= 2 ulp
    against 40-digit mpmath over small, large, huge and near-multiple-of-pi/2 arguments (ocml itself: 0.71 ulp)."""
    import mpmath
    from exahip import ExaCore, ExaModel, rng
    from exahip.graph import cos, sin
    mpmath.mp.dps = 40
    r = np.random.default_rng(0)
    # (the WHOLE range of the lean path, |x| < 823549.6 — 20 000 log-uniform arguments up to the bound, and the neighbourhood of
    # multiples of pi/2 all the way up: k = 1 .. 1000 and 3 000 random k up to 524 287 (k pi/2 < 823 549.6), the double next to k pi/2 and
    # its two neighbours, where the three-term Cody-Waite reduction cancels the most)
    kk = np.concatenate([np.arange(1, 1001), r.integers(1000, 524288, 3000)]).astype(np.float64)
    near = kk * (np.pi / 2)
    xs = np.concatenate([
        r.uniform(-10, 10, 1500), 10 ** r.uniform(-8, 5.9, 1500) * r.choice([-1, 1], 1500),
        10 ** r.uniform(-3, np.log10(823549.5), 20000) * r.choice([-1, 1], 20000),
        (np.arange(1, 1001) * (np.pi / 2)) * (1 + r.uniform(-1e-12, 1e-12, 1000)),
        near, np.nextafter(near, np.inf), np.nextafter(near, -np.inf), -near,
        np.array([0.0, 1e-300, 823549.0, 823549.59, 823550.0, 1e6, 1e15, 3.0e20]),
    ])
    n = len(xs)
    c = ExaCore()
    x = c.add_var(n)
    c.add_con(lambda i: sin(x[i]), rng(1, n))
    c.add_con(lambda i: cos(x[i]), rng(1, n))
    m = ExaModel(c)
    val, jac = m.cons(xs), m.jac_coord(xs)
    for got, fn in ((val[:n], mpmath.sin), (val[n:], mpmath.cos), (jac[:n], mpmath.cos), (-jac[n:], mpmath.sin)):
        worst = 0.0
        for g, xv in zip(got, xs):
            t = fn(mpmath.mpf(float(xv)))
            if t != 0:
                worst = max(worst, float(abs(mpmath.mpf(float(g)) - t) / mpmath.mpf(2) ** (mpmath.floor(mpmath.log(abs(t), 2)) - 52)))
        assert worst <= 2.5, worst                      # (1.7 ulp for |x| < 1000, 2.4 up to the bound: csrc/exa_gen_prelude.cpp)
