"""Deterministic random expression patterns over the reference's function tables (src/functionlist.jl:6-81) — a
property-style parity generator: every tree is built with the exahip front-end exactly as a user would write it
(operators, literal Int/Float constants, parameters, data fields, symbolic variable indices with offsets)."""
import numpy as np

from exahip import ExaCore, Table, graph, rng

UN_SAFE = ["sin", "cos", "tanh", "atan", "asinh", "abs2", "exp", "sqrt", "log", "inv", "cbrt", "sinh", "cosh", "tan",
           "log1p", "expm1", "sech", "atand", "sind", "cospi", "exp2", "abs", "-", "+"]
BIN = ["+", "-", "*", "/", "^i", "^f", "atan2", "hypot", "max", "min", "^v"]
NVAR, NPTS = 40, 7


# the SpecialFunctions extension (ext/functionlist.jl), each with the map that keeps its argument inside the domain
SPECIAL = {
    "erf": lambda a: a, "erfc": lambda a: a, "erfcx": lambda a: a, "dawson": lambda a: a, "erfi": lambda a: 1.5 * graph.tanh(a),
    "airyai": lambda a: 4.0 * graph.tanh(a), "airybi": lambda a: 3.0 * graph.tanh(a), "airyaiprime": lambda a: 4.0 * graph.tanh(a),
    "airybiprime": lambda a: 3.0 * graph.tanh(a), "besselj0": lambda a: a, "besselj1": lambda a: a,
    "bessely0": lambda a: 1.5 + a * a, "bessely1": lambda a: 1.5 + a * a, "gamma": lambda a: 1.5 + graph.tanh(a),
    "digamma": lambda a: 1.25 + a * a, "trigamma": lambda a: 1.25 + a * a, "invdigamma": lambda a: 2.0 * graph.tanh(a),
    "erfinv": lambda a: 0.9 * graph.tanh(a), "erfcinv": lambda a: 1.0 + 0.9 * graph.tanh(a),
}


class Gen:
    block_stride = 0      # > 0: symbolic indices also pick one of three variable blocks this far apart (x, u, z arrays)
    special = 0.0         # > 0: this share of the univariate / bivariate picks comes from the SpecialFunctions extension (the draws of
                          # the default generator are untouched: recorded seeds keep their models)

    user = None           # {table name: constructor}: these entries are taken from the caller (user-registered twins of table functions,
                          # tests/test_registered_functions.py) — the draws do not change, so the tree is the table's tree

    def __init__(self, seed):
        self.r = np.random.default_rng(seed)

    def fn(self, name):
        return (self.user or {}).get(name) or getattr(graph, name)

    def leaf(self, x, th, d):
        k = self.r.integers(0, 10)
        if k < 5:
            off = int(self.r.integers(0, 4))                      # symbolic index with a literal offset
            if self.block_stride:
                off += self.block_stride * int(self.r.integers(0, 3))
            return x[d.i + off]
        if k == 5:
            return x[int(self.r.integers(1, NVAR + 1))]          # constant index
        if k == 6:
            return th[int(self.r.integers(1, 4))]
        if k == 7:
            return d.w                                            # Float data field
        if k == 8:
            return float(np.round(self.r.uniform(0.5, 2.0), 3))
        return int(self.r.integers(1, 4))

    def tree(self, x, th, d, depth):
        if depth == 0 or self.r.uniform() < 0.15:
            return self.leaf(x, th, d)
        if self.special and self.r.uniform() < self.special:
            names = list(SPECIAL) + ["beta", "logbeta"]
            f = names[self.r.integers(0, len(names))]
            a = self.tree(x, th, d, depth - 1)
            if not isinstance(a, graph.Node):
                a = a + x[d.i]
            if f in ("beta", "logbeta"):
                b = self.tree(x, th, d, depth - 1)
                b = 1.25 + b * b if isinstance(b, graph.Node) else abs(float(b)) + 0.5      # a fixed second operand now and then
                return getattr(graph, f)(1.5 + a * a, b)
            return getattr(graph, f)(SPECIAL[f](a))
        if self.r.uniform() < 0.45:
            f = UN_SAFE[self.r.integers(0, len(UN_SAFE))]
            a = self.tree(x, th, d, depth - 1)
            if not isinstance(a, graph.Node):
                a = a + x[d.i]
            if f in ("sqrt", "log", "cbrt", "log1p", "inv"):
                a = 1.5 + a * a                                   # keep the argument in the domain
            if f in ("exp", "sinh", "cosh", "exp2", "expm1", "tan"):
                a = 0.3 * graph.tanh(a)
            if f == "-":
                return -a
            if f == "+":
                return +a
            if f == "abs":
                return abs(a)
            return self.fn(f)(a)
        op = BIN[self.r.integers(0, len(BIN))]
        a = self.tree(x, th, d, depth - 1)
        b = self.tree(x, th, d, depth - 1)
        if not isinstance(a, graph.Node) and not isinstance(b, graph.Node):
            a = a * x[d.j]
        if op == "+":
            return a + b
        if op == "-":
            return a - b
        if op == "*":
            return a * b
        if op == "/":
            return a / (2.0 + b * b) if isinstance(b, graph.Node) else a / b
        if op == "^i":
            base = a if isinstance(a, graph.Node) else b
            return base ** int(self.r.integers(-2, 6))
        if op == "^f":
            base = a if isinstance(a, graph.Node) else b
            return (1.0 + base * base) ** float(np.round(self.r.uniform(-1.5, 2.5), 2))
        if op == "^v":
            base = a if isinstance(a, graph.Node) else b
            ex = b if isinstance(a, graph.Node) else a
            return (1.0 + base * base) ** (graph.tanh(ex) if isinstance(ex, graph.Node) else ex)
        if op == "atan2":
            return graph.atan(a, 1.5 + b * b if isinstance(b, graph.Node) else b)
        if op == "hypot":
            return self.fn("hypot")(a, 1.0 + b)
        if op == "max":
            return graph.maximum(a, b)
        return graph.minimum(a, b)


def registered_twins():
    """User-registered twins of eleven table entries (exa_register_univariate / _fused / _bivariate; rules = the derivative columns of
    src/functionlist.jl in HIP spelling): pass as `user=` and the model is the table's model with these nodes registered instead."""
    R = graph.register_univariate
    return {
        "sin": R("rt_sin", fused="exa_sincos($1, &$2, &$3); $4 = -$2;"),
        "cos": R("rt_cos", fused="double s_; exa_sincos($1, &s_, &$2); $3 = -s_; $4 = -$2;"),
        "tanh": R("rt_tanh", "tanh($1)", "1.0 - $2 * $2", "-2.0 * $2 * $3"),
        "atan": R("rt_atan", "atan($1)", "1.0 / (1.0 + $1 * $1)", "-2.0 * $1 * $3 * $3"),
        "exp": R("rt_exp", fused="$2 = exp($1); $3 = $2; $4 = $2;"),
        "asinh": R("rt_asinh", "asinh($1)", "1.0 / sqrt($1 * $1 + 1.0)", "-$1 * $3 * $3 * $3"),
        "sinh": R("rt_sinh", "sinh($1)", "cosh($1)", "$2"),
        "cosh": R("rt_cosh", "cosh($1)", "sinh($1)", "$2"),
        "log1p": R("rt_log1p", "log1p($1)", "1.0 / (1.0 + $1)", "-$3 * $3"),
        "sqrt": R("rt_sqrt", "sqrt($1)", "0.5 / $2", "-0.5 * $3 / $1"),
        "hypot": graph.register_bivariate("rt_hypot", "hypot($1, $2)", "$1 / $3", "$2 / $3", "($2 * $2) / ($3 * $3 * $3)",
                                          "-($1 * $2) / ($3 * $3 * $3)", "($1 * $1) / ($3 * $3 * $3)"),
    }


def build_model(seed, npat=12, depth=4, special=0.0, user=None):
    """One model with `npat` random objective/constraint/augmentation patterns over a small table iterator."""
    g = Gen(seed)
    c = ExaCore()
    x = c.add_var(NVAR + 4, start=np.linspace(0.3, 1.1, NVAR + 4))
    th = c.add_par(3, value=[0.7, 1.3, -0.4])
    r = np.random.default_rng(seed + 1000)
    tab = Table(i=r.integers(1, NVAR, NPTS), j=r.integers(1, NVAR, NPTS), w=r.uniform(0.5, 1.5, NPTS),
                t=r.integers(1, NPTS + 1, NPTS))
    base = None
    for k in range(npat):
        kind = k % 4
        def fn(d, s=int(g.r.integers(0, 2**31))):
            gen = Gen(s)
            gen.special = special
            gen.user = user
            return gen.tree(x, th, d, depth)
        if kind == 0:
            c.add_obj(fn, tab)
        elif kind in (1, 2) or base is None:
            base = c.add_con(fn, tab)
        else:
            c.add_con_aug(base, lambda d, fn=fn: (d.t, fn(d)), tab)      # shared target rows
    return c


class _RangePoint:
    """Adapter: a range iterator's data point `i` presented with the fields the tree generator reads."""

    def __init__(self, i, step):
        self.i, self.j, self.w, self.t = i, i + 1, 0.75, i if step == 1 else (i - 1) * 0 + 1


def build_range_model(seed, npts=1000, npat=8, depth=4, unit=False, blocks=False):
    """As build_model, but every pattern iterates a RANGE (unit or stepped), so indices are `range + c`: the
    gathered gradient, the LDS-window scatter of J'v / Hv, the multi-tile flush and partial wavefronts are exercised."""
    g = Gen(seed)
    c = ExaCore()
    nvar = 3 * npts + 8
    x = c.add_var(nvar, start=np.linspace(0.3, 1.1, nvar))
    th = c.add_par(3, value=[0.7, 1.3, -0.4])
    base = None
    global NVAR
    saved = NVAR
    NVAR = 40                      # constant-index leaves stay inside the variable block
    try:
        for k in range(npat):
            step = 1 + int(g.r.integers(0, 3)) * (k % 3 == 2)
            lo = 1 + int(g.r.integers(0, 3))
            n = npts - int(g.r.integers(0, 70))
            if unit:                   # stencil-like: unit ranges of (almost) equal length — regular sorted structure
                step, n = 1, npts - (k % 2)
            itr = rng(lo, lo + step * (n - 1), step)
            def fn(i, s=int(g.r.integers(0, 2**31)), st=step):
                gen = Gen(s)
                gen.block_stride = npts if blocks else 0
                return gen.tree(x, th, _RangePoint(i, st), depth)
            kind = k % 4
            if kind == 0:
                c.add_obj(fn, itr)
            elif kind in (1, 2) or base is None:
                base = c.add_con(fn, itr)
                base_n = n
            else:
                m = min(n, base_n)
                c.add_con_aug(base, lambda i, fn=fn: (i, fn(i)), rng(1, m))
    finally:
        NVAR = saved
    return c
