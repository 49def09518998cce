"""Model zoo shared by the parity tests: (name, core builder, evaluation point builder)."""
import numpy as np

from exahip import ExaCore, Table, models, product, rng
from exahip.graph import cos, exp, log, sin, sqrt, tanh


def point(meta_x0, ncon, seed=0, spread=0.1):
    """x = x0 + 0.1 u, u ~ U(-1,1) PCG64 seed 0; y ~ N(0,1) seed 1; sigma = 0.5."""
    x = np.asarray(meta_x0) + spread * np.random.default_rng(seed).uniform(-1, 1, len(meta_x0))
    y = np.random.default_rng(seed + 1).standard_normal(ncon)
    return x, y, 0.5


def mixed_model():
    """Small model exercising parameters, data columns (Int and Float), non-unit ranges, several patterns."""
    c = ExaCore()
    x = c.add_var(12, start=np.linspace(0.5, 1.5, 12))
    z = c.add_var(rng(0, 5), start=0.7)
    th = c.add_par(rng(2, 4), value=[10.0, 20.0, 30.0])
    tab = Table(i=np.array([1, 3, 5, 7]), j=np.array([2, 4, 6, 8]), w=np.array([0.5, 1.5, -2.0, 3.0]), e=np.array([2, 3, 4, 5]))
    c.add_obj(lambda d: d.w * (x[d.i] - x[d.j]) ** 2 + sin(x[d.i] * x[d.j]), tab)
    c.add_obj(lambda i: exp(-z[i]) * z[i] ** 3 + th[2] * z[i], rng(0, 5))
    c.add_con(lambda j: th[j] * x[1] * x[j] + log(x[j + 1]), rng(2, 4))
    g = c.add_con(lambda d: x[d.i] / x[d.j] - d.w * sqrt(x[d.j]) + x[d.i] ** d.e, tab, lcon=-1.0, ucon=1.0)
    c.add_con_aug(g, lambda k: (k, tanh(x[k] * x[k + 4]) - 3 * x[k + 8]), rng(1, 4))
    c.add_con(lambda i: z[i] * z[i - 1] - cos(z[i] - x[12]) + 2.0 ** z[i], rng(1, 5))
    return c


def conaug2d_model():
    """test/NLPTest/conaug_test.jl:200-213 shape: two augmentations on a 2-D empty constraint."""
    N, M = 4, 5
    c = ExaCore()
    x = c.add_var(N, M, start=np.arange(1.0, N * M + 1))
    g = c.add_con(N, M, lcon=-np.inf, ucon=np.inf)
    fwd = [(i, j) for j in range(1, M + 1) for i in range(1, N)]
    bwd = [(i, j) for j in range(1, M + 1) for i in range(2, N + 1)]
    c.add_con_aug(g, lambda p: g[p[0], p[1]] + (x[p[0], p[1]] * x[p[0] + 1, p[1]]), fwd)
    c.add_con_aug(g, lambda p: g[p[0], p[1]] + (x[p[0] - 1, p[1]] - x[p[0], p[1]] ** 2), bwd)
    c.add_obj(lambda p: x[p[0], p[1]] ** 2, product(rng(1, N), rng(1, M)))
    return c


def stepped_model():
    """StepRange iterators (2:3:N), exa_sum / exa_prod nodes, Constant algebra, a parameterised power."""
    from exahip.graph import Constant, exa_prod, exa_sum
    N = 50
    c = ExaCore()
    x = c.add_var(N, start=np.linspace(0.4, 1.6, N))
    th = c.add_par(2, value=[1.5, 0.25])
    c.add_obj(lambda i: (x[i] - x[i + 1]) ** 2 + Constant(1) * x[i] * Constant(0) + th[1] * x[i + 1], rng(1, N - 1, 2))
    c.add_obj(lambda i: exa_sum(x[i + k] ** 2 for k in range(3)) * 0.5, rng(2, N - 2, 3))
    c.add_con(lambda i: exa_prod(1 + x[i + k] for k in range(3)) - x[i] ** th[2] + Constant(2) ** x[i + 1], rng(3, N - 2, 3))
    c.add_con(lambda i: exa_sum([x[i], -x[i - 1], 2 * x[i - 2]]) / (1 + x[i] ** 2), rng(N, 3, -4))
    return c


def small_acopf():
    return models.ac_power_model(models.synthetic_power_data(nbus=30, nbr=41, ngen=6, seed=3))


def trivialmax_model(n=6):
    """test/NLPTest/trivialmax.jl:3-10: a maximisation with scalar (iterator-free) objective and constraint."""
    c = ExaCore(minimize=False)
    x = c.add_var(n, start=0.3)
    c.add_con(x[1], lcon=0.0, ucon=1.0)
    c.add_obj(x[1] ** 2)
    return c


def specialfn_model(n=40):
    """Every entry of the SpecialFunctions extension (ext/functionlist.jl:6-124) inside ordinary patterns: range-indexed objective, a
    table-indexed constraint, beta / logbeta with both operands variable and with a fixed one, an augmentation."""
    from exahip import graph as G
    c = ExaCore()
    x = c.add_var(n, start=np.linspace(0.45, 1.35, n))
    th = c.add_par(2, value=[1.25, 0.5])
    tab = Table(i=np.arange(1, n - 1, 3), j=np.arange(3, n + 1, 3)[: len(np.arange(1, n - 1, 3))], w=np.linspace(0.5, 2.0, len(np.arange(1, n - 1, 3))))
    c.add_obj(lambda i: G.erf(x[i] - x[i + 1]) * G.gamma(x[i] + 1.0) + G.beta(x[i] + 0.5, x[i + 1] + th[1]) + G.airyai(2.0 * x[i] - 3.0 * x[i + 1])
              + G.besselj0(4.0 * x[i]) + G.dawson(x[i] * x[i + 1]) + G.erfcx(x[i] - 2.0) + G.digamma(x[i] + 0.2), rng(1, n - 1))
    g = c.add_con(lambda i: G.erfinv(x[i] * 0.5) + G.invdigamma(x[i] - x[i + 1]) + G.logbeta(x[i] + 0.1, 2.0) + G.airybiprime(x[i + 1] - 2.0)
                  + G.bessely1(x[i] + 0.5) + G.erfi(x[i]) + G.trigamma(x[i] + 0.3) + G.erfcinv(x[i]) + G.erfc(x[i] * x[i + 2]), rng(1, n - 2))
    c.add_con(lambda d: d.w * G.besselj1(x[d.i] * 3.0) * G.bessely0(x[d.j] + 1.0) + G.airybi(-x[d.i] * x[d.j]) + G.airyaiprime(x[d.j])
              + G.beta(th[2] + 1.0, x[d.i]) * G.logbeta(x[d.i], x[d.j]), tab, lcon=-5.0, ucon=5.0)
    c.add_con_aug(g, lambda k: (k, G.erf(x[k] * x[k + 5]) + G.gamma(x[k + 2])), rng(1, 6))
    return c


def docparam_model(N=10):
    """docs/src/parameters.md:24-82 — the reference's own documented parametric model (θ = [100, 1], objective added FIRST): the model whose
    Ipopt logs pin counts and Hessian values (tests/test_known_answers.py)."""
    from exahip.graph import exp, sin
    c = ExaCore()
    th = c.add_par([100.0, 1.0])
    x = c.add_var(N, start=np.array([models.luksan_vlcek_x0(i) for i in range(1, N + 1)]))
    c.add_obj(lambda i: th[1] * (x[i - 1] ** 2 - x[i]) ** 2 + (x[i - 1] - th[2]) ** 2, rng(2, N))
    c.add_con(lambda i: 3 * x[i + 1] ** 3 + 2 * x[i + 2] - 5 + sin(x[i + 1] - x[i + 2]) * sin(x[i + 1] + x[i + 2]) + 4 * x[i + 1]
              - x[i] * exp(x[i] - x[i + 1]) - 3, rng(1, N - 2))
    return c


ZOO = {
    "lv10_docparam": docparam_model,
    "lv3": lambda: models.luksan_vlcek_model(3),
    "lv20": lambda: models.luksan_vlcek_model(20),
    "lv20_objfirst": lambda: models.luksan_vlcek_model(20, obj_first=True),
    "lv_split_20x1": lambda: models.luksan_vlcek_split_model(20, 1),
    "lv_split_20x2": lambda: models.luksan_vlcek_split_model(20, 2),
    "lv_struct_20x2": lambda: models.luksan_vlcek_struct_model(20, 2),      # nested access paths p.i.i[k] (test/NLPTest/luksan_struct.jl)
    "lv1000": lambda: models.luksan_vlcek_model(1000),
    "rocket50": lambda: models.rocket_model(50),
    "acopf30": small_acopf,
    "mixed": mixed_model,
    "conaug2d": conaug2d_model,
    "stepped": stepped_model,
    "cops_chain": lambda: models.cops_chain_model(200),
    "cops_elec": lambda: models.cops_elec_model(25),
    "trivialmax": trivialmax_model,
    "specialfn": specialfn_model,
}
