"""Symbolic expression tree — host-side mirror of the reference's node types.

Restates /root/reference/src/graph.jl:37-300 (node types), src/register.jl:56-74,123-160 (operator
overloads that build Node1/Node2, plain Reals stored directly as Node2 children),
src/specialization.jl:193-202 (`x^2 -> abs2`, `x^1 -> x`, `x^P -> Node2(^, x, Val{P})`) and
:311-339 (Constant{0/1/2} algebra).  It builds the SAME tree shape the Julia closures build, because the
COO slot order of the Jacobian/Hessian is the first-appearance order of a traversal of that tree
(src/simdfunction.jl:78-100) — canonicalising here would change the layout.

Only tree construction lives here; evaluation/differentiation is done by the HIP kernels generated in
csrc/ (product) or by oracle/ (test checker).
"""
from __future__ import annotations

import math
import numbers

import numpy as np

# --- function tables: names in the order of include/exahip_ir.h (== src/functionlist.jl:6-81) -------------
UN_FNS = [
    "+", "-", "inv", "sqrt", "cbrt", "abs", "abs2", "sign", "exp", "exp2", "exp10", "expm1", "log", "log2",
    "log1p", "log10", "sin", "cos", "tan", "asin", "acos", "atan", "acot", "csc", "sec", "cot", "sinh", "cosh",
    "tanh", "asinh", "acosh", "csch", "sech", "coth", "sind", "cosd", "tand", "cscd", "secd", "cotd", "atand",
    "acotd", "sinpi", "cospi", "sinc", "deg2rad", "rad2deg", "signbit", "floor", "ceil", "atanh", "acoth",
    # the SpecialFunctions extension (ext/functionlist.jl:6-102)
    "erf", "erfc", "erfi", "erfcx", "digamma", "trigamma", "invdigamma", "gamma", "airyai", "airybi", "airyaiprime",
    "airybiprime", "besselj0", "bessely0", "besselj1", "bessely1", "dawson", "erfinv", "erfcinv",
]
BIN_FNS = ["+", "-", "*", "/", "^", "atan", "hypot", "max", "min", "beta", "logbeta"]
SPECIAL_UN = UN_FNS[UN_FNS.index("erf"):]
UN_ID = {n: i for i, n in enumerate(UN_FNS)}
BIN_ID = {n: i for i, n in enumerate(BIN_FNS)}


def _is_real(v):
    return isinstance(v, numbers.Real) and not isinstance(v, bool)


def _norm_real(v):
    """numpy scalars -> python int/float (Int stays Int, cf. replace_T: simdfunction.jl:170-172)."""
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    return v


class Node:
    """AbstractNode (graph.jl:11)."""

    __slots__ = ()
    __array_priority__ = 1000  # keep numpy scalars from swallowing the overloads

    # -- bivariate overloads (register.jl:123-160) --
    def __add__(self, o):
        return _bin("+", self, o)

    def __radd__(self, o):
        return _bin("+", o, self)

    def __sub__(self, o):
        return _bin("-", self, o)

    def __rsub__(self, o):
        return _bin("-", o, self)

    def __mul__(self, o):
        return _bin("*", self, o)

    def __rmul__(self, o):
        return _bin("*", o, self)

    def __truediv__(self, o):
        return _bin("/", self, o)

    def __rtruediv__(self, o):
        return _bin("/", o, self)

    def __pow__(self, o):
        return _pow(self, o)

    def __rpow__(self, o):
        return _bin("^", o, self)

    def __neg__(self):
        return _un("-", self)

    def __pos__(self):
        return _un("+", self)

    def __abs__(self):
        return _un("abs", self)


class Constant(Node):
    """Constant{v} (graph.jl:89-91): value-in-type constant that triggers the algebraic rules below."""

    __slots__ = ("v",)

    def __init__(self, v):
        self.v = _norm_real(v)

    def __repr__(self):
        return f"Constant({self.v})"


class Null(Node):
    """Null(v) (graph.jl:37-40): constant row; value None == zero."""

    __slots__ = ("v",)

    def __init__(self, v=None):
        self.v = None if v is None else float(v)

    def __repr__(self):
        return f"Null({self.v})"


class DataSource(Node):
    """The data point p = itr[I] itself (graph.jl:179).  `template` describes one element so that tuple
    destructuring (`for (i, j) in itr`, Base.indexed_iterate graph.jl:275) knows its arity."""

    __slots__ = ("template",)

    def __init__(self, template=None):
        self.template = template

    def path(self):
        return ()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return DataIndexed(self, name)

    def __getitem__(self, k):
        return DataIndexed(self, _norm_real(k))

    def __iter__(self):
        return _iter_fields(self)

    def __repr__(self):
        return "p"


class DataIndexed(Node):
    """DataIndexed{inner, J}: field J of `inner` (graph.jl:190-196)."""

    __slots__ = ("inner", "key")

    def __init__(self, inner, key):
        object.__setattr__(self, "inner", inner)
        object.__setattr__(self, "key", key)

    def path(self):
        return self.inner.path() + (self.key,)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return DataIndexed(self, name)

    def __getitem__(self, k):
        return DataIndexed(self, _norm_real(k))

    def __iter__(self):
        return _iter_fields(self)

    def __repr__(self):
        return "p" + "".join(f".{k}" for k in self.path())


def _template_at(node):
    """Sub-template of the element template at this node's path (None if unknown)."""
    root = node
    while isinstance(root, DataIndexed):
        root = root.inner
    t = root.template
    for k in node.path():
        if t is None:
            return None
        t = _tmpl_get(t, k)
    return t


def _tmpl_get(t, k):
    if isinstance(t, dict):
        return t.get(k)
    if isinstance(t, (tuple, list)):
        return t[k] if isinstance(k, int) and 0 <= k < len(t) else None
    if hasattr(t, "_fields") and isinstance(k, str):
        return getattr(t, k, None)
    return None


def _iter_fields(node):
    t = _template_at(node)
    if isinstance(t, (tuple, list)):
        return iter([DataIndexed(node, k) for k in range(len(t))])
    raise TypeError("cannot destructure a data point whose element shape is unknown or scalar")


class Var(Node):
    """Var(i): i-th decision variable; `i` is an int or an integer-valued Node (graph.jl:138-140)."""

    __slots__ = ("i",)

    def __init__(self, i):
        self.i = _norm_real(i)

    def __repr__(self):
        return f"x[{self.i!r}]"


class ParameterNode(Node):
    """ParameterNode(i): theta[i] (graph.jl:143-145); a constant for differentiation."""

    __slots__ = ("i",)

    def __init__(self, i):
        self.i = _norm_real(i)

    def __repr__(self):
        return f"θ[{self.i!r}]"


class Node1(Node):
    __slots__ = ("fn", "inner")

    def __init__(self, fn, inner):
        self.fn = fn
        self.inner = inner

    def __repr__(self):
        return f"{self.fn}({self.inner!r})"


class Node2(Node):
    """Node2{F}(inner1, inner2); either child may be a plain Real (graph.jl:222-225)."""

    __slots__ = ("fn", "a", "b")

    def __init__(self, fn, a, b):
        self.fn = fn
        self.a = a
        self.b = b

    def __repr__(self):
        return f"({self.a!r} {self.fn} {self.b!r})"


# --- construction rules --------------------------------------------------------------------------------------
_PY_BIN = {
    "+": lambda a, b: a + b,
    "-": lambda a, b: a - b,
    "*": lambda a, b: a * b,
    "/": lambda a, b: a / b,
    "^": lambda a, b: a**b,
    "atan": math.atan2,
    "hypot": math.hypot,
    "max": max,
    "min": min,
    "beta": lambda a, b: _scipy_special("beta")(a, b),
    "logbeta": lambda a, b: _scipy_special("betaln")(a, b),
}


def _scipy_special(name):
    """Plain-number arguments of the SpecialFunctions entries (constant folding at build time only): scipy.special."""
    import scipy.special as sp

    return getattr(sp, name)


def _bin(fn, a, b):
    a = _norm_real(a)
    b = _norm_real(b)
    if not isinstance(a, Node) and not isinstance(b, Node):
        return _PY_BIN[fn](a, b)
    if not (isinstance(a, Node) or _is_real(a)) or not (isinstance(b, Node) or _is_real(b)):
        return NotImplemented
    # Constant folding + Constant{0/1/2} identities (register.jl:141-142; specialization.jl:311-339)
    ca = isinstance(a, Constant)
    cb = isinstance(b, Constant)
    if ca and cb:
        return Constant(_PY_BIN[fn](a.v, b.v))
    if ca and isinstance(b, Node):
        if a.v == 0:
            if fn == "+":
                return b
            if fn == "-":
                return -b
            if fn in ("*", "/"):
                return Constant(0)
            if fn == "^":
                return Constant(0)
        if a.v == 1:
            if fn == "*":
                return b
            if fn == "/":
                return _un("inv", b)
            if fn == "^":
                return Constant(1)
    if cb and isinstance(a, Node):
        if b.v == 0:
            if fn in ("+", "-"):
                return a
            if fn == "*":
                return Constant(0)
            if fn == "^":
                return Constant(1)
        if b.v == 1:
            if fn in ("*", "/", "^"):
                return a
        if b.v == -1 and fn == "^":
            return _un("inv", a)
        if b.v == 2 and fn == "^":
            return _un("abs2", a)
    return Node2(fn, a, b)


def _un(fn, a):
    a = _norm_real(a)
    if isinstance(a, Constant):
        return Constant(_PY_UN[fn](a.v))
    if isinstance(a, Node):
        return Node1(fn, a)
    return _PY_UN[fn](a)


class _IntExp:
    """Marks an exponent that the reference carries as Val{P} (integer literal power)."""


def _pow(x, p):
    p = _norm_real(p)
    if isinstance(p, Node):
        return _bin("^", x, p)
    if isinstance(p, int):
        if hasattr(p, "sym"):
            raise TypeError("a size placeholder as a literal exponent has no recipe form; use powi(x, n)")
        # Base.literal_pow -> _pow_val (specialization.jl:193-202)
        if p == 1:
            return x
        if p == 2:
            return Node1("abs2", x)
        return Node2("^", x, p)  # Val{P}: integer exponent kept as an Int child
    return Node2("^", x, float(p))


def powi(x, n: int):
    """`x^n` with a RUN-TIME Int exponent (not a literal): Node2(^, x, n) without the abs2 rewrite
    (specialization.jl:196)."""
    return Node2("^", x, int(n))


def _cot(x):
    return 1.0 / math.tan(x)


_PY_UN = {
    "+": lambda x: +x,
    "-": lambda x: -x,
    "inv": lambda x: 1.0 / x,
    "sqrt": math.sqrt,
    "cbrt": lambda x: math.copysign(abs(x) ** (1.0 / 3.0), x),
    "abs": abs,
    "abs2": lambda x: x * x,
    "sign": lambda x: (x > 0) - (x < 0),
    "exp": math.exp,
    "exp2": lambda x: 2.0**x,
    "exp10": lambda x: 10.0**x,
    "expm1": math.expm1,
    "log": math.log,
    "log2": math.log2,
    "log1p": math.log1p,
    "log10": math.log10,
    "sin": math.sin,
    "cos": math.cos,
    "tan": math.tan,
    "asin": math.asin,
    "acos": math.acos,
    "atan": math.atan,
    "acot": lambda x: math.atan(1.0 / x),
    "csc": lambda x: 1.0 / math.sin(x),
    "sec": lambda x: 1.0 / math.cos(x),
    "cot": _cot,
    "sinh": math.sinh,
    "cosh": math.cosh,
    "tanh": math.tanh,
    "asinh": math.asinh,
    "acosh": math.acosh,
    "csch": lambda x: 1.0 / math.sinh(x),
    "sech": lambda x: 1.0 / math.cosh(x),
    "coth": lambda x: 1.0 / math.tanh(x),
    "sind": lambda x: math.sin(math.radians(x)),
    "cosd": lambda x: math.cos(math.radians(x)),
    "tand": lambda x: math.tan(math.radians(x)),
    "cscd": lambda x: 1.0 / math.sin(math.radians(x)),
    "secd": lambda x: 1.0 / math.cos(math.radians(x)),
    "cotd": lambda x: 1.0 / math.tan(math.radians(x)),
    "atand": lambda x: math.degrees(math.atan(x)),
    "acotd": lambda x: math.degrees(math.atan(1.0 / x)),
    "sinpi": lambda x: math.sin(math.pi * x),
    "cospi": lambda x: math.cos(math.pi * x),
    "sinc": lambda x: 1.0 if x == 0 else math.sin(math.pi * x) / (math.pi * x),
    "deg2rad": math.radians,
    "rad2deg": math.degrees,
    "signbit": lambda x: float(math.copysign(1.0, x) < 0),
    "floor": math.floor,
    "ceil": math.ceil,
    "atanh": math.atanh,
    "acoth": lambda x: math.atanh(1.0 / x),
    "erf": math.erf,
    "erfc": math.erfc,
    "erfi": lambda x: float(_scipy_special("erfi")(x)),
    "erfcx": lambda x: float(_scipy_special("erfcx")(x)),
    "digamma": lambda x: float(_scipy_special("digamma")(x)),
    "trigamma": lambda x: float(_scipy_special("polygamma")(1, x)),
    "invdigamma": lambda x: _invdigamma(x),
    "gamma": math.gamma,
    "airyai": lambda x: float(_scipy_special("airy")(x)[0]),
    "airyaiprime": lambda x: float(_scipy_special("airy")(x)[1]),
    "airybi": lambda x: float(_scipy_special("airy")(x)[2]),
    "airybiprime": lambda x: float(_scipy_special("airy")(x)[3]),
    "besselj0": lambda x: float(_scipy_special("j0")(x)),
    "bessely0": lambda x: float(_scipy_special("y0")(x)),
    "besselj1": lambda x: float(_scipy_special("j1")(x)),
    "bessely1": lambda x: float(_scipy_special("y1")(x)),
    "dawson": lambda x: float(_scipy_special("dawsn")(x)),
    "erfinv": lambda x: float(_scipy_special("erfinv")(x)),
    "erfcinv": lambda x: float(_scipy_special("erfcinv")(x)),
}


def _invdigamma(y):
    x = math.exp(y) + 0.5 if y >= -2.22 else -1.0 / (y + 0.5772156649015329)
    dg, pg = _scipy_special("digamma"), _scipy_special("polygamma")
    for _ in range(25):
        xn = x - (float(dg(x)) - y) / float(pg(1, x))
        if abs(xn - x) <= 1e-15 * abs(xn):
            return xn
        x = xn
    return x


def register_univariate(name, f=None, df=None, ddf=None, helpers="", py=None, fused=None):
    """The reference's `@register_univariate(f, df, ddf)` (src/register.jl:56-74) across the C ABI: the three rules are HIP device
    expressions — `f` in `$1` (the argument), `df` in `$1 $2` (= f), `ddf` in `$1 $2 $3` (= df); `"=0"`-style strings are exact constants;
    `helpers` is device code the expressions may call (include/exahip.h: exa_register_univariate).  Returns the node constructor:

        softplus = register_univariate("softplus", "log1p(exp($1))", "1.0 / (1.0 + exp(-$1))", "$3 * (1.0 - $3)")
        c.add_obj(lambda i: softplus(x[i] - x[i + 1]), rng(1, N - 1))

    `py` (optional): the same function on plain Python numbers, for arguments that are literal constants at build time.
    `fused` (instead of f / df / ddf): ONE device statement that computes all three — `$1` the argument, `$2 $3 $4` receive f, f', f'' —
    for functions whose derivatives share work with the value: `fused="exa_sincos($1, &$2, &$3); $4 = -$2;"` (exa_register_univariate_fused)."""
    from . import capi
    enc = lambda t: None if t is None else str(t).encode()
    if fused is not None:
        if f is not None or df is not None or ddf is not None:
            raise ValueError("give either the three rules or `fused`")
        fid = capi.lib().exa_register_univariate_fused(enc(name), enc(fused), enc(helpers) if helpers else None)
    else:
        fid = capi.lib().exa_register_univariate(enc(name), enc(f), enc(df), enc(ddf), enc(helpers) if helpers else None)
    if fid < 0:
        raise ValueError(capi.lib().exa_last_error().decode())
    UN_ID[name] = fid
    if py is not None:
        _PY_UN[name] = py
    elif name not in _PY_UN:
        _PY_UN[name] = lambda x, _n=name: (_ for _ in ()).throw(TypeError(f"`{_n}` of a literal constant: give register_univariate a `py` callable"))
    return _make_un(name)


def register_bivariate(name, f, d1, d2, d11, d12, d22, helpers="", py=None):
    """`@register_bivariate(f, d1, d2, d11, d12, d22)` (src/register.jl:123-276): rules in `$1 $2` (the arguments) and, for the partials,
    `$3` (= f).  With one operand constant the node becomes the reference's FirstFixed / SecondFixed form, which takes d2/d22 or d1/d11
    of these same rules.  Returns the node constructor `g(a, b)`."""
    from . import capi
    enc = lambda t: None if t is None else str(t).encode()
    fid = capi.lib().exa_register_bivariate(enc(name), enc(f), enc(d1), enc(d2), enc(d11), enc(d12), enc(d22), enc(helpers) if helpers else None)
    if fid < 0:
        raise ValueError(capi.lib().exa_last_error().decode())
    BIN_ID[name] = fid
    if py is not None:
        _PY_BIN[name] = py
    elif name not in _PY_BIN:
        _PY_BIN[name] = lambda a, b, _n=name: (_ for _ in ()).throw(TypeError(f"`{_n}` of two literal constants: give register_bivariate a `py` callable"))

    def g(a, b):
        return _bin(name, a, b)

    g.__name__ = name
    return g


def _make_un(name):
    def f(x):
        return _un(name, x)

    f.__name__ = name
    f.__doc__ = f"`{name}` registered as a univariate node function (src/functionlist.jl:6-60)."
    return f


# module-level math functions usable on nodes and on plain numbers
inv = _make_un("inv")
sqrt = _make_un("sqrt")
cbrt = _make_un("cbrt")
abs2 = _make_un("abs2")
sign = _make_un("sign")
exp = _make_un("exp")
exp2 = _make_un("exp2")
exp10 = _make_un("exp10")
expm1 = _make_un("expm1")
log = _make_un("log")
log2 = _make_un("log2")
log1p = _make_un("log1p")
log10 = _make_un("log10")
sin = _make_un("sin")
cos = _make_un("cos")
tan = _make_un("tan")
asin = _make_un("asin")
acos = _make_un("acos")
acot = _make_un("acot")
csc = _make_un("csc")
sec = _make_un("sec")
cot = _make_un("cot")
sinh = _make_un("sinh")
cosh = _make_un("cosh")
tanh = _make_un("tanh")
asinh = _make_un("asinh")
acosh = _make_un("acosh")
csch = _make_un("csch")
sech = _make_un("sech")
coth = _make_un("coth")
sind = _make_un("sind")
cosd = _make_un("cosd")
tand = _make_un("tand")
cscd = _make_un("cscd")
secd = _make_un("secd")
cotd = _make_un("cotd")
atand = _make_un("atand")
acotd = _make_un("acotd")
sinpi = _make_un("sinpi")
cospi = _make_un("cospi")
sinc = _make_un("sinc")
deg2rad = _make_un("deg2rad")
rad2deg = _make_un("rad2deg")
signbit = _make_un("signbit")
floor = _make_un("floor")
ceil = _make_un("ceil")
atanh = _make_un("atanh")
acoth = _make_un("acoth")
for _name in SPECIAL_UN:      # erf, erfc, ..., erfcinv as module-level functions (docstring: ext/functionlist.jl)
    globals()[_name] = _make_un(_name)
    globals()[_name].__doc__ = f"`{_name}` of the SpecialFunctions extension, registered as a univariate node function (ext/functionlist.jl:6-102)."
uplus = _make_un("+")
uminus = _make_un("-")


def atan(y, x=None):
    """1-arg: univariate atan; 2-arg: bivariate atan(y, x) (functionlist.jl:27,77)."""
    if x is None:
        return _un("atan", y)
    return _bin("atan", y, x)


def hypot(a, b):
    return _bin("hypot", a, b)


def beta(a, b):
    """Bivariate `beta` of the SpecialFunctions extension (ext/functionlist.jl:109-116)."""
    return _bin("beta", a, b)


def logbeta(a, b):
    """Bivariate `logbeta` (log|B(a, b)|) of the SpecialFunctions extension (ext/functionlist.jl:117-124)."""
    return _bin("logbeta", a, b)


def maximum(a, b):
    return _bin("max", a, b)


def minimum(a, b):
    return _bin("min", a, b)


def exa_sum(terms):
    """SumNode: left fold with binary + (graph.jl:549-567 — `reduce(+, ...)` in both adjoint modes)."""
    terms = list(terms)
    if not terms:
        return Null(None)
    acc = terms[0]
    for t in terms[1:]:
        acc = acc + t
    return acc


def exa_prod(terms):
    """ProdNode: left fold with binary * (graph.jl:549-567)."""
    terms = list(terms)
    if not terms:
        return Null(1.0)
    acc = terms[0]
    for t in terms[1:]:
        acc = acc * t
    return acc
