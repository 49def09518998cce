"""Degenerate shapes: empty iterators, no constraints, no objective, all-linear (nnzh = 0), constant rows, unused
variables, single points, wavefront/tile boundary sizes.  Every callback must still fully overwrite its output."""
import numpy as np

from exahip import ExaCore, Table, rng
from exahip.graph import exp, sin


def no_constraints(n=37):
    c = ExaCore()
    x = c.add_var(n, start=np.linspace(-1, 1, n))
    c.add_obj(lambda i: (x[i] - x[i + 1]) ** 2 + sin(x[i]), rng(1, n - 1))
    return c


def no_objective(n=29):
    c = ExaCore()
    x = c.add_var(n, start=np.linspace(0.2, 1.2, n))
    c.add_con(lambda i: x[i] * x[i + 1] - exp(x[i]), rng(1, n - 1))
    return c


def empty_iterators():
    """patterns with zero data points between non-empty ones; an empty table"""
    c = ExaCore()
    x = c.add_var(12, start=np.linspace(0.5, 1.5, 12))
    c.add_obj(lambda i: x[i] ** 3, rng(1, 0))
    c.add_obj(lambda i: x[i] ** 4, rng(2, 9))
    c.add_con(lambda i: x[i] * x[i + 1], rng(5, 4))
    g = c.add_con(lambda i: sin(x[i]) + x[i + 2], rng(1, 10))
    c.add_con_aug(g, lambda t: (t.r, t.w * x[t.i] ** 2), Table(r=np.zeros(0, dtype=np.int64), i=np.zeros(0, dtype=np.int64), w=np.zeros(0)))
    c.add_con_aug(g, lambda t: (t.r, t.w * x[t.i] ** 2), Table(r=np.array([2, 2, 9]), i=np.array([1, 12, 5]), w=np.array([0.5, -1.0, 2.0])))
    return c


def all_linear():
    """nnzh == 0: hess_coord! has nothing to write; Jacobian entries are constants"""
    c = ExaCore()
    x = c.add_var(9, start=np.arange(1.0, 10.0))
    c.add_obj(lambda i: 2.0 * x[i] - x[i + 1], rng(1, 8))
    c.add_con(lambda i: x[i] + 3 * x[i + 1] - 1.5, rng(1, 8))
    return c


def constants_and_unused():
    """rows without any variable (o1step = 0), variables no pattern touches, a single-point objective"""
    c = ExaCore()
    x = c.add_var(10, start=0.3)
    th = c.add_par(2, value=[2.0, -1.0])
    c.add_obj(x[3] ** 2 * x[7])
    c.add_con(lambda i: th[1] * i + 0.5, rng(1, 4))
    c.add_con(lambda i: x[2 * i] ** 2 - th[2], rng(1, 3))
    return c


def boundary_sizes(n):
    """n data points around the wavefront (64), workgroup (256) and staging-tile boundaries"""
    def make():
        c = ExaCore()
        x = c.add_var(n + 2, start=np.linspace(0.1, 1.9, n + 2))
        c.add_obj(lambda i: (x[i] - 1) ** 2 * x[i + 1], rng(1, n))
        c.add_con(lambda i: sin(x[i] * x[i + 1]) + exp(x[i + 2] - x[i]), rng(1, n))
        return c
    return make


EDGE = {
    "no_constraints": no_constraints,
    "no_objective": no_objective,
    "empty_iterators": empty_iterators,
    "all_linear": all_linear,
    "constants_and_unused": constants_and_unused,
    **{f"n{n}": boundary_sizes(n) for n in (1, 63, 64, 65, 255, 256, 257, 4097)},
}
