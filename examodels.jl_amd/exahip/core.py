"""ExaCore — host-side mirror of the reference's model builder, emitting the pattern table (IR).

Restates the bookkeeping of /root/reference/src/nlp.jl: `add_var` (variable blocks + `getindex`,
:900-926), `add_par` (:954-962), `add_obj` (:1448-1482), `add_con` (:1551-1611), `add_con!`
(:1679-1738), the `g[idx] += expr` sugar (:196-264), offsets `idxx` (:2012-2015).  Counter order is
insertion order; the evaluator library recomputes o0/o1/o2 from that order (include/exahip_ir.h).

This is NOT a re-implementation of the modeling macros / JuMP / recipes (out of scope, SURVEY §2 rows
12,13,20); it is the minimum needed to state the reference's test and benchmark models in Python and to
hand them to libexahip.so (and to the test oracle) as data.
"""
from __future__ import annotations

import ctypes
import itertools

import numpy as np

from . import graph as G
from . import recipe as R
from .recipe import ArgTable, TArray, TInt, keep_int, max0
from .graph import (BIN_ID, UN_ID, Constant, DataIndexed, DataSource, Node, Node1, Node2, Null,
                    ParameterNode, Var, _is_real, _norm_real)

# ---------------------------------------------------------------------------------------------------------------
# ctypes view of include/exahip_ir.h
# ---------------------------------------------------------------------------------------------------------------
OP_CONST_F, OP_CONST_I, OP_DATA, OP_PAR, OP_VAR, OP_UN, OP_BIN, OP_NULLV = range(8)
COL_I64, COL_F64, COL_RANGE = range(3)
PAT_OBJ, PAT_CON, PAT_CONAUG = range(3)


class CNode(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int32), ("fn", ctypes.c_int32), ("a", ctypes.c_int32), ("b", ctypes.c_int32),
                ("fval", ctypes.c_double), ("ival", ctypes.c_int64)]


class CColumn(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("_pad", ctypes.c_int32), ("data", ctypes.c_void_p),
                ("start", ctypes.c_int64), ("step", ctypes.c_int64)]


class CPattern(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("n_nodes", ctypes.c_int32), ("nodes", ctypes.POINTER(CNode)),
                ("root", ctypes.c_int32), ("target", ctypes.c_int32), ("base", ctypes.c_int32),
                ("n_cols", ctypes.c_int32), ("cols", ctypes.POINTER(CColumn)), ("n", ctypes.c_int64)]


class CModelDesc(ctypes.Structure):
    _fields_ = [("nvar", ctypes.c_int64), ("npar", ctypes.c_int64),
                ("x0", ctypes.c_void_p), ("lvar", ctypes.c_void_p), ("uvar", ctypes.c_void_p),
                ("theta0", ctypes.c_void_p),
                ("n_patterns", ctypes.c_int32), ("minimize", ctypes.c_int32),
                ("patterns", ctypes.POINTER(CPattern)),
                ("y0", ctypes.c_void_p), ("lcon", ctypes.c_void_p), ("ucon", ctypes.c_void_p)]


# ---------------------------------------------------------------------------------------------------------------
# iterators (the `itr` of a pattern), normalised to struct-of-arrays columns
# ---------------------------------------------------------------------------------------------------------------
class URange:
    """Julia-style inclusive range a:b or a:s:b."""

    def __init__(self, start, stop, step=1):
        self.start, self.stop, self.step = keep_int(start), keep_int(stop), keep_int(step)

    @property
    def length(self):
        """len(self) that keeps a recipe's size expression (Python's len() must return a plain int)."""
        if self.step > 0:
            return max0((self.stop - self.start) // self.step + 1)
        return max0((self.start - self.stop) // (-self.step) + 1)

    def __len__(self):
        return int(self.length)

    def __iter__(self):
        return iter(range(int(self.start), int(self.start) + int(self.step) * len(self), int(self.step)))

    def __repr__(self):
        return f"{self.start}:{self.stop}" if self.step == 1 else f"{self.start}:{self.step}:{self.stop}"

    @property
    def first(self):
        return self.start


def rng(a, b, step=1):
    return URange(a, b, step)


def _as_urange(r):
    if isinstance(r, URange):
        return r
    if isinstance(r, range):
        if len(r) == 0:
            return URange(r.start, r.start - 1, 1)
        return URange(r.start, r[-1], r.step)
    raise TypeError(r)


class Table:
    """Struct-of-arrays iterator: dict field -> 1-D numpy array (int64 or float64).  Equivalent to a Julia
    Vector of NamedTuples, already transposed."""

    def __init__(self, **cols):
        self.cols = {}
        n = None
        for k, v in cols.items():
            a = np.asarray(v)
            if a.dtype.kind in "iu":
                a = np.ascontiguousarray(a, dtype=np.int64)
            else:
                a = np.ascontiguousarray(a, dtype=np.float64)
            if n is None:
                n = len(a)
            assert len(a) == n, "Table columns must have equal length"
            self.cols[k] = a
        self.n = 0 if n is None else n

    def __len__(self):
        return self.n


class Product:
    """Iterators.product(r1, r2, ...) / multi-`for` generator: first index fastest (column-major)."""

    def __init__(self, *axes):
        self.axes = [a if isinstance(a, (URange, range, TArray)) else np.asarray(a) for a in axes]

    def __len__(self):
        n = 1
        for a in self.axes:
            n *= len(a)
        return n


def product(*axes):
    return Product(*axes)


class _Iter:
    """Normalised iterator: n, element template, column lookup by access path."""

    def __init__(self, itr):
        self.src = itr
        self._cache = {}
        if isinstance(itr, (URange, range)):
            r = _as_urange(itr)
            self.kind = "range"
            self.n = r.length
            self.r = r
            self.template = 0
            self.dims = (r,)
        elif isinstance(itr, ArgTable):            # recipe placeholder: a table whose rows arrive at instantiation
            self.kind = "table"
            self.src = Table(**itr.cols)
            self.arg = itr
            self.n = itr.nrows
            self.template = {k: (0 if v.dtype.kind == "i" else 0.0) for k, v in itr.cols.items()}
            self.dims = (self.n,)
        elif isinstance(itr, TArray) and itr.src is not None:   # recipe placeholder array iterated directly
            self.kind = "list"
            self.lst = np.asarray(itr)
            self.arg = itr
            self.n = itr.n
            self.dims = (self.n,)
            self.template = _template_of(self.lst[0]) if len(self.lst) else 0
        elif isinstance(itr, Table):
            self.kind = "table"
            self.n = itr.n
            self.template = {k: (0 if v.dtype.kind == "i" else 0.0) for k, v in itr.cols.items()}
            self.dims = (self.n,)
        elif isinstance(itr, Product):
            self.kind = "product"
            self.axes = itr.axes
            self.dims = tuple(_axis_len(a) for a in itr.axes)
            self.n = 1
            for d in self.dims:
                self.n = self.n * d
            self.template = tuple(0 for _ in itr.axes)
        elif isinstance(itr, np.ndarray) and itr.dtype.names:
            self.kind = "table"
            t = Table(**{k: itr[k] for k in itr.dtype.names})
            self.src = t
            self.n = t.n
            self.template = {k: (0 if v.dtype.kind == "i" else 0.0) for k, v in t.cols.items()}
            self.dims = (self.n,)
        elif isinstance(itr, np.ndarray) and itr.dtype == object and itr.ndim > 1:
            # a Matrix of structs (test/NLPTest/luksan_struct.jl: data[1:end-2, :]): Base.size(itr) = its shape, elements in
            # column-major order (Julia's linear index)
            self.kind = "list"
            self.lst = list(itr.flatten(order="F"))
            self.n = len(self.lst)
            self.dims = tuple(itr.shape)
            self.template = _template_of(self.lst[0]) if self.n else 0
        else:
            lst = itr if isinstance(itr, (list, tuple, np.ndarray)) else list(itr)
            self.kind = "list"
            self.lst = lst
            self.n = len(lst)
            self.dims = tuple(np.shape(lst)[:1]) if not isinstance(lst, np.ndarray) else lst.shape[:1]
            if self.n:
                self.template = _template_of(lst[0])
            else:
                self.template = 0

    def column(self, path):
        """-> (coltype, ndarray|None, start, step)"""
        if path in self._cache:
            return self._cache[path]
        if self.kind == "range":
            assert path == (), "a range element has no fields"
            out = (COL_RANGE, None, self.r.start, self.r.step)
        elif self.kind == "table":
            assert len(path) == 1, f"bad field path {path}"
            a = self.src.cols[path[0]]
            out = (COL_I64 if a.dtype.kind == "i" else COL_F64, a, 0, 0)
        elif self.kind == "product":
            assert len(path) == 1 and isinstance(path[0], int)
            k = path[0]
            inner = 1
            for a in self.axes[:k]:
                inner *= len(a)
            outer = int(self.n) // (inner * len(self.axes[k])) if int(self.n) else 0
            ax = self.axes[k]
            vals = np.fromiter(iter(ax), dtype=np.int64, count=len(ax)) if isinstance(ax, (URange, range)) \
                else np.asarray(ax)
            a = np.tile(np.repeat(vals, inner), outer)
            if a.dtype.kind in "iu":
                a = np.ascontiguousarray(a, dtype=np.int64)
                out = (COL_I64, a, 0, 0)
            else:
                a = np.ascontiguousarray(a, dtype=np.float64)
                out = (COL_F64, a, 0, 0)
        else:
            vals = [_get_path(e, path) for e in self.lst]
            a = np.asarray(vals)
            if a.dtype.kind in "iu":
                a = np.ascontiguousarray(a, dtype=np.int64)
                out = (COL_I64, a, 0, 0)
            elif a.dtype.kind == "b":
                raise TypeError("boolean data fields are not supported")
            else:
                a = np.ascontiguousarray(a, dtype=np.float64)
                out = (COL_F64, a, 0, 0)
        self._cache[path] = out
        return out


def _axis_len(a):
    if isinstance(a, (URange, range)):
        return _as_urange(a).length
    if isinstance(a, TArray) and a.src is not None:
        return a.n
    return len(a)


def _rcolumn(it, path):
    """Recipe form of the column `path` of iterator `it` (include/exahip_recipe.h, column kinds)."""
    if it.kind == "range":
        return (R.RCOL_RANGE, it.r.start, it.r.step)
    arg = getattr(it, "arg", None)
    if it.kind == "table" and arg is not None:
        return (R.RCOL_COL, arg.field, list(arg.cols).index(path[0]))
    if it.kind == "list" and arg is not None:
        assert path == (), "a placeholder array element has no fields"
        return (R.RCOL_FIELD, arg.src[1]) if arg.src[0] == R.SRC_FIELD else (R.RCOL_COL, arg.src[1], arg.src[2])
    if it.kind == "product":
        k = path[0]
        inner = 1
        for a in it.axes[:k]:
            inner = inner * _axis_len(a)
        ax = it.axes[k]
        tagged = any(isinstance(v, TInt) for v in it.dims)
        if isinstance(ax, (URange, range)):
            r = _as_urange(ax)
            if tagged or isinstance(r.start, TInt) or isinstance(r.step, TInt):
                return (R.RCOL_AXIS_RANGE, r.start, r.step, r.length, inner)
        elif isinstance(ax, TArray) and ax.src is not None:
            col = -1 if ax.src[0] == R.SRC_FIELD else ax.src[2]
            return (R.RCOL_AXIS_FIELD, ax.src[1], col, ax.n, inner)
        elif tagged:
            a = np.asarray(ax)
            kind = R.RCOL_AXIS_INLINE_I64 if a.dtype.kind in "iu" else R.RCOL_AXIS_INLINE_F64
            return (kind, a, len(a), inner)
    ct, arr, st, sp = it.column(path)          # size-independent data: inline
    if ct == COL_RANGE:
        return (R.RCOL_RANGE, st, sp)
    if isinstance(it.n, TInt):
        raise R.RecipeError("an iterator of placeholder length needs placeholder data (a table / array field)")
    return (R.RCOL_INLINE_I64 if ct == COL_I64 else R.RCOL_INLINE_F64, arr)


def _template_of(e):
    if isinstance(e, dict):
        return {k: _template_of(v) for k, v in e.items()}
    if hasattr(e, "_fields"):
        return {k: _template_of(getattr(e, k)) for k in e._fields}
    if isinstance(e, (tuple, list)):
        return tuple(_template_of(v) for v in e)
    if isinstance(e, np.ndarray) and e.ndim >= 1:
        return tuple(_template_of(v) for v in e)
    return _norm_real(e)


def _get_path(e, path):
    for k in path:
        if isinstance(e, dict):
            e = e[k]
        elif isinstance(k, str):
            e = getattr(e, k)
        else:
            e = e[k]
    return e


# ---------------------------------------------------------------------------------------------------------------
# blocks
# ---------------------------------------------------------------------------------------------------------------
def _start(s):
    return s.start if isinstance(s, URange) else 1


def _length(s):
    return s.length if isinstance(s, URange) else keep_int(s)


def _idxx(coord, sizes):
    """idxx(coord, si) = 1 + sum_d stride_d (coord_d - 1), column-major (nlp.jl:2012-2015).  Written with the
    same operator sequence so that symbolic coords yield the same index tree."""

    def rec(c, s, a):
        if not c:
            return 0
        return a * (c[0] - 1) + rec(c[1:], s[1:], a * s[0] if s else a)

    return rec(tuple(coord), tuple(sizes), 1) + 1


class Variable:
    """Variable block (nlp.jl:66-75): `x[i]`, `x[i, j]` build Var nodes (nlp.jl:900-926)."""

    def __init__(self, size, length, offset):
        self.size, self.length, self.offset = size, length, offset

    def __getitem__(self, i):
        if isinstance(i, tuple):
            assert len(i) == len(self.size), "Variable index dimension error"
            adj = tuple(_sub(ii, _start(s) - 1) for ii, s in zip(i, self.size))
            return Var(self.offset + _idxx(adj, tuple(_length(s) for s in self.size)))
        o = self.offset - _start(self.size[0]) + 1
        i = _norm_real(i)
        if isinstance(i, Node):
            return Var(Node2("+", i, o))        # _indexed_var(i::AbstractNode, o::Int)
        _bound_check(self.size[0], i)
        return Var(i + o)


def _sub(a, b):
    a = _norm_real(a)
    if isinstance(a, Node):
        return Node2("-", a, b)      # `is .- (start .- 1)`: Node - Int is a plain Node2 (no Int-zero rule)
    return a - b


def _bound_check(s, i):
    if isinstance(s, URange):
        assert s.start <= i <= s.stop, "Variable index bound error"
    else:
        assert 1 <= i <= s, "Variable index bound error"


class Parameter:
    """Parameter block (nlp.jl:954-962)."""

    def __init__(self, size, length, offset):
        self.size, self.length, self.offset = size, length, offset

    def __getitem__(self, i):
        if isinstance(i, tuple):
            assert len(i) == len(self.size), "Parameter index dimension error"
            adj = tuple(_sub(ii, _start(s) - 1) for ii, s in zip(i, self.size))
            return ParameterNode(self.offset + _idxx(adj, tuple(_length(s) for s in self.size)))
        return ParameterNode(i + (self.offset - _start(self.size[0]) + 1))


class Expression:
    """Subexpression (`add_expr`): spliced in wherever it is indexed (nlp.jl:928-952)."""

    def __init__(self, fn, size):
        self.fn, self.size = fn, size
        self.length = 1
        for s in size:
            self.length *= _length(s)

    def __getitem__(self, i):
        # the body sees its index exactly as an iterator element would arrive: a scalar for 1-D, a tuple otherwise
        return self.fn(i)


class ConstraintSlot:
    """`g[idx]` handle for the `g[idx] += expr` sugar (nlp.jl:207-248)."""

    def __init__(self, con, idx):
        self.con, self.idx = con, idx

    def _mk(self, expr):
        if _is_real(expr):
            expr = Null(expr)
        return ConAugPair(self.con, self.idx, expr)

    def __add__(self, expr):
        return self._mk(expr)

    def __radd__(self, expr):
        return self._mk(expr)

    def __iadd__(self, expr):
        return self._mk(expr)

    def __sub__(self, expr):
        return self._mk(-expr)

    def __isub__(self, expr):
        return self._mk(-expr)


class ConAugPair:
    def __init__(self, con, idx, expr):
        self.con, self.idx, self.expr = con, idx, expr


class Constraint:
    """Constraint block handle (nlp.jl:137-143)."""

    def __init__(self, pat_index, offset, n, size, dims):
        self.pat_index = pat_index   # position in core.patterns of the BASE pattern
        self.offset = offset         # o: first row is offset+1
        self.n = n
        self.size = size             # tuple of ints / URanges (range start info)
        self.dims = dims             # tuple of lengths: Base.size(c.itr)

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            adj = tuple(_con_adjust(ii, _start(s)) for ii, s in zip(idx, self.size))
            return ConstraintSlot(self, adj)
        return ConstraintSlot(self, _con_adjust(idx, _start(self.size[0])))


class ConstraintAugmentation:
    """ConstraintAugmentation handle (nlp.jl:171-177); indexing does not re-adjust (nlp.jl:240-241)."""

    def __init__(self, base: Constraint, pat_index):
        self.base = base
        self.pat_index = pat_index
        self.dims = base.dims

    def __getitem__(self, idx):
        return ConstraintSlot(self.base, idx)


def _con_adjust(idx, start):
    idx = _norm_real(idx)
    if isinstance(idx, Node):
        return Node2("-", idx, start - 1)
    return idx - start + 1


class Objective:
    def __init__(self, pat_index):
        self.pat_index = pat_index


class _Pattern:
    def __init__(self, kind, expr, itr: _Iter, target=None, base=-1):
        self.kind, self.expr, self.itr, self.target, self.base = kind, expr, itr, target, base


# ---------------------------------------------------------------------------------------------------------------
# ExaCore
# ---------------------------------------------------------------------------------------------------------------
class ExaCore:
    """Mutable accumulator (the reference's ExaCore is immutable and returned anew, nlp.jl:328-366; the
    counters and their order are the same)."""

    def __init__(self, minimize=True, examples=None):
        """`examples` makes the core a RECIPE (the reference's `ExaCore(nargs = Val(N))`, nlp.jl:507-523): one
        placeholder per example value is available as `core.args` — see exahip/recipe.py."""
        self.minimize = bool(minimize)
        self.schema = None
        self.args = ()
        if examples is not None:
            self.schema = R.Schema()
            self.args = tuple(R.make_placeholders(self.schema, tuple(examples)))
        self.blocks = []          # named blocks (name, kind 0 var / 1 con / 2 par, offset, length, dims): cnlp P_block
        self._segs = {k: [] for k in ("x0", "lvar", "uvar", "theta", "y0", "lcon", "ucon")}
        self.nvar = 0
        self.npar = 0
        self.ncon = 0
        self.nconaug = 0
        self.nobj = 0
        self.x0, self.lvar, self.uvar = [], [], []
        self.theta = []
        self.y0, self.lcon, self.ucon = [], [], []
        self.patterns: list[_Pattern] = []

    # -- variables / parameters ---------------------------------------------------------------------------
    def add_var(self, *ns, start=0.0, lvar=-np.inf, uvar=np.inf, name=None):
        size = tuple(_as_size(n) for n in ns)
        length = 1
        for s in size:
            length = length * _length(s)
        o = self.nvar
        self.nvar = self.nvar + length
        self._vec("x0", self.x0, start, length)
        self._vec("lvar", self.lvar, lvar, length)
        self._vec("uvar", self.uvar, uvar, length)
        self._block(name, 0, o, length, size)
        return Variable(size, length, o)

    def _vec(self, which, parts, value, n):
        """append one block of a start/bound vector, remembering where it came from (recipe segments)"""
        if _is_real(value) and not isinstance(value, TArray):
            parts.append(_ConstVec(float(value), int(n)))      # materialised only if somebody asks for the array
        else:
            parts.append(_fill(value, int(n)))
        if isinstance(value, TArray) and value.src is not None:
            src = value.src
        elif isinstance(value, TArray) or np.ndim(value) > 0 or callable(value):
            if isinstance(n, TInt):
                raise R.RecipeError(f"{which}: a block of placeholder length needs a scalar or a placeholder array")
            src = (R.SRC_INLINE, parts[-1])
        else:
            src = (R.SRC_CONST, float(value))
        self._segs[which].append((n, src))

    def _block(self, name, kind, offset, length, size):
        if name is not None:
            self.blocks.append((str(name), kind, offset, length, tuple(_length(s) for s in size)))

    def add_par(self, *ns, value=0.0, name=None):
        if len(ns) == 1 and not isinstance(ns[0], (int, np.integer, URange, range)):
            value = ns[0]
            ns = (R.length(value),)
        size = tuple(_as_size(n) for n in ns)
        length = 1
        for s in size:
            length = length * _length(s)
        o = self.npar
        self.npar = self.npar + length
        self._vec("theta", self.theta, value, length)
        self._block(name, 2, o, length, size)
        return Parameter(size, length, o)

    def set_value(self, par: Parameter, values):
        """set_value!(core, θ, vals) before the model is built (nlp.jl:1279-1287)."""
        if any(isinstance(n, TInt) or src[0] in (R.SRC_FIELD, R.SRC_COL) for n, src in self._segs["theta"]):
            raise R.RecipeError("set_value on a recipe core: set the instance's values after instantiation")
        flat = np.concatenate(self.theta) if self.theta else np.zeros(0)
        flat[par.offset:par.offset + par.length] = _fill(values, par.length)
        self.theta = [flat]
        self._segs["theta"] = [(len(flat), (R.SRC_INLINE, flat))]

    # -- subexpressions ------------------------------------------------------------------------------------
    def add_expr(self, fn, itr):
        it = _Iter(itr)
        if it.kind == "product":        # size = the axes themselves when they are ranges (subexpr_test.jl:81)
            size = tuple(_as_urange(a) if isinstance(a, (URange, range)) else len(a) for a in it.axes)
        else:
            size = it.dims
        return Expression(fn, size)

    # -- objective ------------------------------------------------------------------------------------------
    def add_obj(self, fn, itr=None):
        """add_obj(core, f(p) for p in itr).  `fn` is a callable of the symbolic data point, or a Node
        (then itr defaults to 1:1, nlp.jl:1468)."""
        it = _Iter(URange(1, 1) if itr is None else itr)
        expr = fn(DataSource(it.template)) if callable(fn) else fn
        self.nobj = self.nobj + it.n
        self.patterns.append(_Pattern(PAT_OBJ, expr, it))
        return Objective(len(self.patterns) - 1)

    # -- constraints -----------------------------------------------------------------------------------------
    def add_con(self, *args, start=0.0, lcon=0.0, ucon=0.0, name=None):
        """add_con(core, f, itr; ...) or add_con(core, dims...; ...) (empty rows, nlp.jl:1570-1581)."""
        if args and (callable(args[0]) or isinstance(args[0], Node)):
            fn = args[0]
            it = _Iter(args[1] if len(args) > 1 else URange(1, 1))
            expr = fn(DataSource(it.template)) if callable(fn) else fn
            size = _infer_dims(it)
        else:
            size = tuple(_as_size(n) for n in args)
            if len(size) == 1:
                it = _Iter(URange(1, _length(size[0])))
            else:
                it = _Iter(Product(*[URange(1, _length(s)) for s in size]))
            expr = Null(None)
        if _is_real(expr):
            expr = Null(expr)
        o = self.ncon
        n = it.n
        self.ncon = self.ncon + n
        self._vec("y0", self.y0, start, n)
        self._vec("lcon", self.lcon, lcon, n)
        self._vec("ucon", self.ucon, ucon, n)
        self.patterns.append(_Pattern(PAT_CON, expr, it))
        dims = tuple(_length(s) for s in size)
        self._block(name, 1, o, n, size)
        return Constraint(len(self.patterns) - 1, o, n, size, dims)

    def add_con_aug(self, c1, fn, itr):
        """add_con!(core, c1, idx => expr for p in itr) (nlp.jl:1679-1687).  `fn(p)` returns `(idx, expr)` with
        idx an index / tuple of indices into c1, or a ConAugPair from the `g[idx] += expr` sugar (then c1 may be
        None, nlp.jl:1704-1722)."""
        it = _Iter(itr)
        res = fn(DataSource(it.template))
        if isinstance(res, ConAugPair):
            base = res.con
            idx, expr = res.idx, res.expr
        else:
            idx, expr = res
            base = c1.base if isinstance(c1, ConstraintAugmentation) else c1
        if _is_real(expr):
            expr = Null(expr)
        if isinstance(idx, tuple):
            target = _idxx(tuple(idx), base.dims)     # offset0(...) = o0 + idxx(coord, dims) (nlp.jl:2000-2001)
        else:
            target = idx                               # o0 + idx(p)                         (nlp.jl:1996-1997)
        self.nconaug = self.nconaug + it.n
        self.patterns.append(_Pattern(PAT_CONAUG, expr, it, target=target, base=base.pat_index))
        return ConstraintAugmentation(base, len(self.patterns) - 1)

    # -- IR emission --------------------------------------------------------------------------------------------
    def to_ir(self):
        return ModelIR(self)

    def to_recipe(self) -> bytes:
        """Wire form (include/exahip_recipe.h): sizes/data deferred to the schema fields for a recipe core, fully
        inline for an ordinary one."""
        return R.dumps(self)


def _as_size(n):
    if isinstance(n, (URange, range)):
        return _as_urange(n)
    return keep_int(n)


def _infer_dims(it: _Iter):
    if it.kind == "range":
        return (it.r,)
    if it.kind == "product":
        return tuple(a if isinstance(a, URange) else (_as_urange(a) if isinstance(a, range) else len(a))
                     for a in it.axes)
    if it.kind == "list" and len(it.dims) > 1:      # a Matrix of data points: the constraint block has its shape (nlp.jl:1588-1590)
        return tuple(int(d) for d in it.dims)
    return (it.n,)


class _ConstVec:
    """n copies of one value.  A model at N = 1e8 has six such vectors of 0.8 GB each; the library fills in its own
    defaults (0, -Inf, +Inf) when it is handed no array at all (include/exahip_ir.h), so they are never built."""
    __slots__ = ("value", "n")

    def __init__(self, value, n):
        self.value, self.n = value, n

    def __array__(self, dtype=None, copy=None):
        return np.full(self.n, self.value, dtype=dtype or np.float64)

    def __len__(self):
        return self.n


def _fill(v, n):
    if callable(v):
        return np.array([float(v(i)) for i in range(1, n + 1)], dtype=np.float64)
    a = np.asarray(v, dtype=np.float64)
    if a.ndim == 0:
        return np.full(n, float(a))
    a = a.reshape(-1, order="F")
    assert a.size == n, f"expected {n} values, got {a.size}"
    return np.ascontiguousarray(a)


def lower_pattern(p: _Pattern, symbolic=False):
    """Post-order node list + column list of one pattern -> (nodes, cols, root, target).
    nodes: (op, fn, a, b, fval, ival); with `symbolic` the ival of an Int leaf may be a tagged TInt and the
    columns are in recipe form (exahip/recipe.py), otherwise (coltype, array, start, step)."""
    nodes = []
    cols = []
    colid = {}

    def col_of(path):
        if path not in colid:
            colid[path] = len(cols)
            cols.append(_rcolumn(p.itr, path) if symbolic else p.itr.column(path))
        return colid[path]

    def emit(e):
        e = _norm_real(e)
        if isinstance(e, bool):
            raise TypeError("bool in expression")
        if isinstance(e, int):
            nodes.append((OP_CONST_I, 0, -1, -1, 0.0, e if symbolic else int(e)))
        elif isinstance(e, float):
            nodes.append((OP_CONST_F, 0, -1, -1, e if symbolic else float(e), 0))     # a deferred real keeps its tag
        elif isinstance(e, Constant):
            return emit(e.v)
        elif isinstance(e, Null):
            nodes.append((OP_NULLV, 0, -1, -1, 0.0 if e.v is None else e.v, 0))
        elif isinstance(e, (DataSource, DataIndexed)):
            nodes.append((OP_DATA, 0, col_of(e.path()), -1, 0.0, 0))
        elif isinstance(e, Var):
            a = emit(e.i)
            nodes.append((OP_VAR, 0, a, -1, 0.0, 0))
        elif isinstance(e, ParameterNode):
            a = emit(e.i)
            nodes.append((OP_PAR, 0, a, -1, 0.0, 0))
        elif isinstance(e, Node1):
            a = emit(e.inner)
            nodes.append((OP_UN, UN_ID[e.fn], a, -1, 0.0, 0))
        elif isinstance(e, Node2):
            a = emit(e.a)
            b = emit(e.b)
            nodes.append((OP_BIN, BIN_ID[e.fn], a, b, 0.0, 0))
        else:
            raise TypeError(f"cannot lower {type(e).__name__} into the pattern IR")
        return len(nodes) - 1

    root = emit(p.expr)
    target = emit(p.target) if p.kind == PAT_CONAUG else -1
    return nodes, cols, root, target


class ModelIR:
    """Owns the ctypes arrays of one exa_model_desc_t (keeps every buffer alive)."""

    def __init__(self, core: ExaCore):
        self.core = core
        self._keep = []
        pats = (CPattern * max(1, len(core.patterns)))()
        for k, p in enumerate(core.patterns):
            self._emit_pattern(p, pats[k])
        self.patterns = pats
        d = CModelDesc()
        d.nvar, d.npar = core.nvar, core.npar
        # None = "all default": the descriptor then carries a NULL pointer and the library fills the constant in
        self._vecs = {"x0": _cat(core.x0, 0.0), "lvar": _cat(core.lvar, -np.inf), "uvar": _cat(core.uvar, np.inf),
                      "theta0": _cat(core.theta), "y0": _cat(core.y0, 0.0), "lcon": _cat(core.lcon, 0.0), "ucon": _cat(core.ucon, 0.0)}
        self._defaults = {"x0": (0.0, core.nvar), "lvar": (-np.inf, core.nvar), "uvar": (np.inf, core.nvar),
                          "y0": (0.0, core.ncon), "lcon": (0.0, core.ncon), "ucon": (0.0, core.ncon)}
        v = self._vecs
        d.x0, d.lvar, d.uvar = _ptr(v["x0"]), _ptr(v["lvar"]), _ptr(v["uvar"])
        d.theta0 = _ptr(v["theta0"])
        d.n_patterns = len(core.patterns)
        d.minimize = 1 if core.minimize else 0
        d.patterns = ctypes.cast(pats, ctypes.POINTER(CPattern))
        d.y0, d.lcon, d.ucon = _ptr(v["y0"]), _ptr(v["lcon"]), _ptr(v["ucon"])
        self.desc = d

    def __getattr__(self, name):
        # x0 / lvar / uvar / theta0 / y0 / lcon / ucon as arrays (an all-default vector is built on first use)
        vecs = self.__dict__.get("_vecs")
        if vecs is not None and name in vecs:
            if vecs[name] is None:
                value, n = self._defaults[name]
                vecs[name] = np.full(int(n), value)      # the descriptor keeps its NULL: same contents by definition
            return vecs[name]
        raise AttributeError(name)

    def _emit_pattern(self, p: _Pattern, out: CPattern):
        nodes, cols, root, target = lower_pattern(p)
        cn = (CNode * len(nodes))()
        for i, t in enumerate(nodes):
            cn[i].op, cn[i].fn, cn[i].a, cn[i].b, cn[i].fval, cn[i].ival = t
        cc = (CColumn * max(1, len(cols)))()
        for i, (ct, arr, st, sp) in enumerate(cols):
            cc[i].type = ct
            cc[i].data = arr.ctypes.data if arr is not None else None
            cc[i].start, cc[i].step = st, sp
            if arr is not None:
                self._keep.append(arr)
        self._keep += [cn, cc]
        out.kind = p.kind
        out.n_nodes = len(nodes)
        out.nodes = ctypes.cast(cn, ctypes.POINTER(CNode))
        out.root = root
        out.target = target
        out.base = p.base
        out.n_cols = len(cols)
        out.cols = ctypes.cast(cc, ctypes.POINTER(CColumn))
        out.n = p.itr.n


def _cat(parts, default=None):
    """one contiguous array, or None when every block is the library's default constant for this vector"""
    if not parts:
        return np.zeros(0, dtype=np.float64)
    if default is not None and all(isinstance(a, _ConstVec) and a.value == default for a in parts):
        return None
    if len(parts) == 1:            # a single block (the benchmark models at N = 1e8): no second copy
        return np.ascontiguousarray(np.asarray(parts[0], dtype=np.float64))
    return np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64) for a in parts]))


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None
