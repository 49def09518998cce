"""Recipe models of ExaModelsCompiler/test/runtests.jl restated on the host mirror, each as a function of its
instantiation values so that the same builder yields the recipe (example values) and the concrete reference."""
import numpy as np

from exahip import ExaCore, Table, product, rng

INF = float("inf")


def knob(n):
    """runtests.jl:21-27 `build`: min sum (x_i - 2)^2  s.t.  x_i + x_{i+1} >= 3; one integer placeholder."""
    return (n,)


def build_knob(c, N):
    x = c.add_var(N, start=1.0, name="x")
    c.add_obj(lambda i: (x[i] - 2.0) ** 2, rng(1, N))
    c.add_con(lambda i: x[i] + x[i + 1], rng(1, N - 1), lcon=3.0, ucon=INF)


def build_struct(c, sz, dat, tab):
    """runtests.jl:60-66 `sbuild`: a bare size, a NamedTuple carrying a start and a bound, and a table."""
    x = c.add_var(sz, start=dat["v0"], lvar=dat["lo"], name="x")
    c.add_obj(lambda t: t.w * (x[t.i] - t.s) ** 2, tab)
    c.add_con(lambda i: x[i] + x[i + 1], rng(1, sz - 1), lcon=-100.0, ucon=100.0)


def build_layout(c, N):
    """runtests.jl:43-55 `pbuild`: every block NAMED — a variable, a parameter, two constraints — and a data-axis
    constraint whose iterator is a collection of points (here already flattened to rows (t1, t2, j), first axis
    fastest, so its block reports dims (4,) where the reference reports (2, 2))."""
    y = c.add_var(N, 2, start=0.5, name="y")
    w = c.add_par(3, value=1.0, name="w")
    c.add_con(lambda i: y[i, 1] + y[i, 2], rng(1, N), lcon=0.0, ucon=4.0, name="link")
    rows = [(t1, t2, j) for j in (1, 2) for (t1, t2) in ((1, 0.5), (2, 1.5))]
    c.add_con(lambda p: y[p[0], 1] - p[1], rows, lcon=-10.0, ucon=10.0, name="dax")
    c.add_obj(lambda p: w[1] * (y[p[0], p[1]] - 2.0) ** 2, product(rng(1, N), rng(1, 2)))


def make(builder, values, recipe):
    """recipe=True: a recipe core traced on `values` as EXAMPLES; False: the concrete core at `values`."""
    if recipe:
        c = ExaCore(examples=values)
        args = list(c.args)
    else:
        c = ExaCore()
        args = list(values)
    # NamedTuple placeholders index by key like dicts
    builder(c, *args)
    return c


S_EX = (4, dict(v0=np.full(4, 0.5), lo=np.full(4, -10.0)),
        Table(i=np.array([1, 3]), w=np.array([2.0, 1.0]), s=np.array([1.0, 0.5])))
S_N = 6
S_ARGS = (S_N, dict(v0=np.linspace(0.1, 0.6, S_N), lo=np.full(S_N, -5.0)),
          Table(i=np.array([2, 5, 6]), w=np.array([1.5, 3.0, 0.5]), s=np.array([2.0, -1.0, 0.0])))


def lv_recipe_builder(c, N):
    """The headline model with its size left open: BASELINE configs[0..1] from ONE recipe."""
    from exahip.graph import exp, sin
    from exahip.models import luksan_vlcek_x0
    x = c.add_var(N, start=0.0, name="x")        # x0 alternates by index: not a recipe-expressible start; set by caller
    c.add_con(lambda i: 3 * x[i + 1] ** 3 + 2 * x[i + 2] - 5 + sin(x[i + 1] - x[i + 2]) * sin(x[i + 1] + x[i + 2])
              + 4 * x[i + 1] - x[i] * exp(x[i] - x[i + 1]) - 3, rng(1, N - 2), name="s")
    c.add_obj(lambda i: 100 * (x[i - 1] ** 2 - x[i]) ** 2 + (x[i - 1] - 1) ** 2, rng(2, N))
