"""Models of test/NLPTest/parameter_test.jl restated on the host mirror (the reference file holds the Julia
builders; here only their mathematical content and insertion order are reproduced)."""
import numpy as np

from exahip import ExaCore, models, product, rng
from exahip.graph import exp, sin

DEFAULT_THETA = [100.0, 1.0, 3.0, 2.0, 5.0, 4.0, 3.0]


def lv_parametric(N, M=1, use_parameters=True, param_values=None):
    """parameter_test.jl:21-74: the split Luksan-Vlcek model with its 7 literals either as θ[1..7] or inlined."""
    c = ExaCore()
    x0 = np.array([[models.luksan_vlcek_x0(i) for _ in range(M)] for i in range(1, N + 1)])
    x = c.add_var(N, M, start=x0)
    vals = list(DEFAULT_THETA if param_values is None else param_values)
    th = None
    if use_parameters:
        th = c.add_par(7, value=np.zeros(7))
        c.set_value(th, vals)
        p = [None] + [th[k] for k in range(1, 8)]
    else:
        p = [None] + vals

    def con1(d):
        i, j = d
        return p[3] * x[i + 1, j] ** 3 + p[4] * x[i + 2, j] - p[5]

    def con2(d):
        i, j = d
        return ((i, j), sin(x[i + 1, j] - x[i + 2, j]) * sin(x[i + 1, j] + x[i + 2, j]) + p[6] * x[i + 1, j]
                - x[i, j] * exp(x[i, j] - x[i + 1, j]) - p[7])

    def obj(d):
        i, j = d
        return p[1] * (x[i - 1, j] ** 2 - x[i, j]) ** 2 + (x[i - 1, j] - p[2]) ** 2

    s = c.add_con(con1, product(rng(1, N - 2), rng(1, M)))
    c.add_con_aug(s, con2, product(rng(1, N - 2), rng(1, M)))
    c.add_obj(obj, product(rng(2, N), rng(1, M)))
    return c, th


def real_only():
    """parameter_test.jl:227-243: constant base rows/objective + augmentation over (i, j) targeting row j."""
    c = ExaCore()
    x = c.add_var(10)
    c1 = c.add_con(lambda i: 1.0, rng(1, 2))
    c.add_obj(lambda i: 1.0, rng(1, 2))
    c.add_con_aug(c1, lambda d: (d[1], -(x[d[0]] - 1) ** 2), product(rng(1, 10), rng(1, 2)))
    return c


def param_only(theta):
    """parameter_test.jl:245-264: the same with θ[i] in place of the literal 1.0."""
    c = ExaCore()
    x = c.add_var(10)
    th = c.add_par(2, value=np.asarray(theta, float))
    c1 = c.add_con(lambda i: th[i], rng(1, 2))
    c.add_obj(lambda i: th[i], rng(1, 2))
    c.add_con_aug(c1, lambda d: (d[1], -(x[d[0]] - 1) ** 2), product(rng(1, 10), rng(1, 2)))
    return c


PARAM_SETS = {
    "default": None,
    "objective": [200.0, 2.0, 3.0, 2.0, 5.0, 4.0, 3.0],
    "constraints": [100.0, 1.0, 6.0, 4.0, 10.0, -8.0, 6.0],
    "all": [150.0, 0.5, 2.5, 1.5, 7.5, 3.5, 4.5],
}
AFTER_BUILD = [75.0, 1.5, 4.0, 3.0, 6.0, 5.0, 2.0]


def dense(rows, cols, vals, shape, symmetric=False):
    a = np.zeros(shape)
    np.add.at(a, (np.asarray(rows) - 1, np.asarray(cols) - 1), vals)
    if symmetric:
        a = a + np.tril(a, -1).T
    return a
