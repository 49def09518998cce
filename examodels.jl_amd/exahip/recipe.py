"""Recipes: models whose sizes and data are left open until instantiation.

Reference: a core built against `ArgSource` placeholders (src/argument.jl:67-185, `ExaCore(nargs = Val(N))`
src/nlp.jl:507-523) is a *recipe*; `ExaModel(core, args...)` (nlp.jl:809-863) instantiates it, and
ExaModelsCompiler publishes it behind the cnlp schema/builder ABI (ExaModelsCompiler.jl:1197-1330).

Here the placeholders are *example values that remember where they came from*: `TInt` is an `int` whose value is
the example's and whose `.sym` is the expression (over schema fields) that produced it; `TArray` / `ArgTable`
carry the schema field they stand for.  The ordinary ExaCore bookkeeping runs unchanged on them — every offset,
stride, range bound and block length it computes comes out tagged — and `Recipe.dumps()` writes the pattern
table with those expressions in place of the numbers (wire format: include/exahip_recipe.h).  libexahip
evaluates them against the data bound through the builder ABI and plans/compiles the concrete model.

The example fixes types (which fields are ints, float arrays, tables with which columns), never values — the
same role it has in `compile_library(out, core, example...)` (ExaModelsCompiler.jl:138-160).
"""
from __future__ import annotations

import struct

import numpy as np

# symbolic integer expression opcodes (include/exahip_recipe.h)
SYM_CONST, SYM_SCALAR, SYM_LEN, SYM_ADD, SYM_SUB, SYM_MUL, SYM_FLOORDIV, SYM_MAX0, SYM_NEG = range(9)
# ... and REAL-valued expressions of the sizes (a coefficient such as h = 1 / (N + 1), ArgumentTest.jl:231-280)
SYM_FCONST, SYM_ITOF, SYM_FADD, SYM_FSUB, SYM_FMUL, SYM_FDIV, SYM_FNEG = range(9, 16)
FIELD_SCALAR, FIELD_ARRAY, FIELD_TABLE = range(3)
TYPE_I64, TYPE_F64 = range(2)
SRC_CONST, SRC_INLINE, SRC_FIELD, SRC_COL = range(4)
(RCOL_RANGE, RCOL_INLINE_I64, RCOL_INLINE_F64, RCOL_FIELD, RCOL_COL, RCOL_AXIS_RANGE, RCOL_AXIS_FIELD,
 RCOL_AXIS_INLINE_I64, RCOL_AXIS_INLINE_F64) = range(9)
MAGIC = b"EXARCP01"


class RecipeError(TypeError):
    pass


class TInt(int):
    """An example integer that remembers the expression (over schema fields) it stands for."""

    def __new__(cls, value, sym):
        o = int.__new__(cls, int(value))
        o.sym = sym
        return o

    # value semantics (hash / == / ordering) stay int's: the example decides control flow, as in the reference
    @staticmethod
    def _s(v):
        if isinstance(v, TInt):
            return v.sym
        if isinstance(v, (bool, float)) or not isinstance(v, (int, np.integer)):
            return None
        return (SYM_CONST, int(v))

    def _bin(self, o, op, f, swap=False):
        so = TInt._s(o)
        if so is None:
            if isinstance(o, (float, np.floating)):          # mixed with a real: the result is a deferred real
                fop = {SYM_ADD: "__add__", SYM_SUB: "__sub__", SYM_MUL: "__mul__"}.get(op)
                if fop is None:
                    raise RecipeError("floor division of a size placeholder by a real has no recipe form")
                me, other = TFloat.of(self), TFloat.of(o)
                return getattr(other, fop)(me) if swap else getattr(me, fop)(other)
            return NotImplemented
        a, b = (so, self.sym) if swap else (self.sym, so)
        va, vb = (int(o), int(self)) if swap else (int(self), int(o))
        return TInt(f(va, vb), (op, a, b))

    def __add__(self, o): return self._bin(o, SYM_ADD, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, SYM_ADD, lambda a, b: a + b, True)
    def __sub__(self, o): return self._bin(o, SYM_SUB, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, SYM_SUB, lambda a, b: a - b, True)
    def __mul__(self, o): return self._bin(o, SYM_MUL, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, SYM_MUL, lambda a, b: a * b, True)
    def __floordiv__(self, o): return self._bin(o, SYM_FLOORDIV, lambda a, b: a // b)
    def __rfloordiv__(self, o): return self._bin(o, SYM_FLOORDIV, lambda a, b: a // b, True)
    def __neg__(self): return TInt(-int(self), (SYM_NEG, self.sym, None))
    def __pos__(self): return self

    def _no(self, *_):
        raise RecipeError("this operation on a size placeholder has no recipe form (deferred: + - * // and / )")

    # true division leaves the integers: the result is a deferred REAL (h = 1 / (N + 1))
    def __truediv__(self, o): return TFloat.of(self) / o
    def __rtruediv__(self, o): return o / TFloat.of(self)

    __mod__ = __rmod__ = __pow__ = __rpow__ = _no
    __lshift__ = __rshift__ = __and__ = __or__ = __xor__ = _no

    def __repr__(self):
        return f"TInt({int(self)})"


class TFloat(float):
    """An example real that remembers the expression over the sizes it stands for (a deferred coefficient)."""

    def __new__(cls, value, sym):
        o = float.__new__(cls, float(value))
        o.sym = sym
        return o

    @staticmethod
    def of(v):
        if isinstance(v, TFloat):
            return v
        if isinstance(v, TInt):
            return TFloat(int(v), (SYM_ITOF, v.sym, None))
        if isinstance(v, (bool,)) or not isinstance(v, (int, float, np.integer, np.floating)):
            return None
        return TFloat(v, (SYM_FCONST, float(v), None))

    def _bin(self, o, op, f, swap=False):
        b = TFloat.of(o)
        if b is None:
            return NotImplemented
        x, y = (b, self) if swap else (self, b)
        return TFloat(f(float(x), float(y)), (op, x.sym, y.sym))

    def __add__(self, o): return self._bin(o, SYM_FADD, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, SYM_FADD, lambda a, b: a + b, True)
    def __sub__(self, o): return self._bin(o, SYM_FSUB, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, SYM_FSUB, lambda a, b: a - b, True)
    def __mul__(self, o): return self._bin(o, SYM_FMUL, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, SYM_FMUL, lambda a, b: a * b, True)
    def __truediv__(self, o): return self._bin(o, SYM_FDIV, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, SYM_FDIV, lambda a, b: a / b, True)
    def __neg__(self): return TFloat(-float(self), (SYM_FNEG, self.sym, None))
    def __pos__(self): return self

    def _no(self, *_):
        raise RecipeError("this operation on a deferred real has no recipe form (deferred: + - * /)")

    __pow__ = __rpow__ = __mod__ = __rmod__ = __floordiv__ = __rfloordiv__ = _no

    def __repr__(self):
        return f"TFloat({float(self)})"


def keep_int(v):
    """int(v) that does not strip the tag."""
    return v if isinstance(v, TInt) else int(v)


def max0(v):
    """max(0, v) whose recipe form is evaluated at instantiation (range lengths)."""
    if isinstance(v, TInt):
        return TInt(max(0, int(v)), (SYM_MAX0, v.sym, None))
    return max(0, int(v))


def is_tagged(v):
    return isinstance(v, TInt)


class TArray(np.ndarray):
    """Example array standing for an `array` field or a table column of the schema."""

    def __new__(cls, values, src, n):
        o = np.asarray(values).view(cls)
        o.src = src          # (SRC_FIELD, field) | (SRC_COL, field, col)
        o.n = n              # TInt length
        return o

    def __array_finalize__(self, obj):
        # any derived array (slice, arithmetic, copy) is ordinary data: it no longer IS the field
        self.src = None
        self.n = None


def length(v):
    """Julia's `length`: the deferred length of a placeholder array/table, else len(v)."""
    if isinstance(v, TArray) and v.src is not None:
        return v.n
    if isinstance(v, ArgTable):
        return v.nrows
    return len(v)


class ArgTable:
    """Example table (struct of arrays) standing for a `table` field: iterate it (`for t in tab`) or use its
    columns as start/bound vectors."""

    def __init__(self, field, cols, nrows):
        self.field, self.cols, self.nrows = field, cols, nrows   # cols: name -> TArray
        self.n = int(nrows)

    def __getattr__(self, name):
        cols = self.__dict__.get("cols", {})
        if name in cols:
            return cols[name]
        raise AttributeError(name)

    def __getitem__(self, name):
        return self.cols[name]

    def __len__(self):
        return self.n


class ArgStruct:
    """Example NamedTuple argument: entries by key (`dat.v0`)."""

    def __init__(self, entries):
        self.__dict__.update(entries)

    def __getitem__(self, k):
        return self.__dict__[k]


class Schema:
    """The flattened example (ExaModelsCompiler.jl:1130-1148 publishes exactly this as `P_schema`): bare values by
    position (`arg1`, ...), NamedTuple entries by key, a table with its typed columns."""

    def __init__(self):
        self.fields = []      # dicts: name, kind, type, columns [(name, type)]

    def add(self, name, kind, typ=TYPE_I64, columns=()):
        if any(f["name"] == name for f in self.fields):
            raise RecipeError(f"duplicate schema field {name!r}")
        self.fields.append({"name": name, "kind": kind, "type": typ, "columns": list(columns)})
        return len(self.fields) - 1

    def json(self):
        tn = {TYPE_I64: "i64", TYPE_F64: "f64"}
        parts = []
        for f in self.fields:
            if f["kind"] == FIELD_TABLE:
                cols = ",".join('{"name":"%s","type":"%s"}' % (n, tn[t]) for n, t in f["columns"])
                parts.append('{"name":"%s","kind":"table","columns":[%s]}' % (f["name"], cols))
            else:
                kind = "scalar" if f["kind"] == FIELD_SCALAR else "array"
                parts.append('{"name":"%s","kind":"%s","type":"%s"}' % (f["name"], kind, tn[f["type"]]))
        return '{"fields":[' + ",".join(parts) + "]}"


def _np_type(a):
    return TYPE_I64 if np.asarray(a).dtype.kind in "iu" else TYPE_F64


def _table_columns(ex):
    """example table (a Vector of NamedTuples in the reference) -> ordered {name: ndarray}: a Table, a structured
    array or a list of dicts.  A plain dict is a NamedTuple of fields, not a table."""
    from .core import Table
    if isinstance(ex, Table):
        return dict(ex.cols)
    if isinstance(ex, np.ndarray) and ex.dtype.names:
        return {k: ex[k] for k in ex.dtype.names}
    if isinstance(ex, (list, tuple)) and ex and isinstance(ex[0], dict):
        return {k: np.asarray([r[k] for r in ex]) for k in ex[0]}
    return None


def make_placeholders(schema: Schema, examples):
    """One placeholder per example argument (the k-th is the reference's ArgSource{k})."""
    out = []
    for k, ex in enumerate(examples, 1):
        name = f"arg{k}"
        cols = _table_columns(ex)
        if isinstance(ex, (int, np.integer)) and not isinstance(ex, bool):
            f = schema.add(name, FIELD_SCALAR, TYPE_I64)
            out.append(TInt(ex, (SYM_SCALAR, f)))
        elif isinstance(ex, (float, np.floating)):
            raise RecipeError("a bare real argument has no recipe form; pass it inside an array or as a parameter value")
        elif cols is not None:
            out.append(_table_placeholder(schema, name, cols))
        elif isinstance(ex, dict):            # NamedTuple of scalars / arrays / tables: entries by key
            entries = {}
            for key, v in ex.items():
                c2 = _table_columns(v)
                if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
                    f = schema.add(key, FIELD_SCALAR, TYPE_I64)
                    entries[key] = TInt(v, (SYM_SCALAR, f))
                elif c2 is not None:
                    entries[key] = _table_placeholder(schema, key, c2)
                else:
                    entries[key] = _array_placeholder(schema, key, v)
            out.append(ArgStruct(entries))
        else:
            out.append(_array_placeholder(schema, name, ex))
    return out


def _array_placeholder(schema, name, v):
    a = np.asarray(v)
    if a.ndim != 1:
        raise RecipeError(f"schema field {name!r}: only 1-D arrays can be deferred")
    t = _np_type(a)
    f = schema.add(name, FIELD_ARRAY, t)
    a = np.ascontiguousarray(a, dtype=np.int64 if t == TYPE_I64 else np.float64)
    return TArray(a, (SRC_FIELD, f), TInt(len(a), (SYM_LEN, f)))


def _table_placeholder(schema, name, cols):
    n = None
    typed = []
    for cn, a in cols.items():
        a = np.asarray(a)
        n = len(a) if n is None else n
        if len(a) != n:
            raise RecipeError(f"table {name!r}: columns differ in length")
        typed.append((cn, _np_type(a)))
    f = schema.add(name, FIELD_TABLE, columns=typed)
    nrows = TInt(n or 0, (SYM_LEN, f))
    tc = {}
    for ci, (cn, t) in enumerate(typed):
        a = np.ascontiguousarray(cols[cn], dtype=np.int64 if t == TYPE_I64 else np.float64)
        tc[cn] = TArray(a, (SRC_COL, f, ci), nrows)
    return ArgTable(f, tc, nrows)


# ---------------------------------------------------------------------------------------------------------------
# serialisation (include/exahip_recipe.h)
# ---------------------------------------------------------------------------------------------------------------
class _W:
    def __init__(self):
        self.b = bytearray()
        self.syms = []          # (op, a, b) with a/b literal or earlier ids
        self._memo = {}

    def i32(self, v): self.b += struct.pack("<i", int(v))
    def i64(self, v): self.b += struct.pack("<q", int(v))
    def f64(self, v): self.b += struct.pack("<d", float(v))

    def s(self, text):
        raw = text.encode()
        self.i32(len(raw))
        self.b += raw

    def arr(self, a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        self.i64(a.size)
        self.b += a.tobytes()

    def sym(self, t):
        """intern a TInt expression tree, post-order, into the SSA table -> id"""
        if t in self._memo:
            return self._memo[t]
        op = t[0]
        if op in (SYM_CONST, SYM_SCALAR, SYM_LEN):
            rec = (op, t[1], 0)
        elif op == SYM_FCONST:
            rec = (op, struct.unpack("<q", struct.pack("<d", t[1]))[0], 0)      # the double's bit pattern
        elif op in (SYM_MAX0, SYM_NEG, SYM_ITOF, SYM_FNEG):
            rec = (op, self.sym(t[1]), 0)
        else:
            rec = (op, self.sym(t[1]), self.sym(t[2]))
        self.syms.append(rec)
        self._memo[t] = len(self.syms) - 1
        return self._memo[t]

    def ival(self, v):
        """a size-like integer: literal, or a reference into the symbol table"""
        if isinstance(v, TInt):
            self.i32(1)
            self.i64(self.sym(v.sym))
        else:
            self.i32(0)
            self.i64(int(v))


USER_FN_BASE = 1000                 # exa_register_univariate / _bivariate ids start here (include/exahip.h)


def _user_text(bivariate, fn, which):
    """One text of a registration, read back from the library (exa_user_function: 0 name, 1 f, 2 d1, 3 d2, 4 d11, 5 d12, 6 d22, 7 helpers, 8 fused)."""
    import ctypes
    from . import capi
    L = capi.lib()
    n = L.exa_user_function(bivariate, fn, which, None, 0)
    if n < 0:
        raise ValueError(f"function id {fn} is not registered in this process")
    buf = ctypes.create_string_buffer(n + 1)
    L.exa_user_function(bivariate, fn, which, buf, n + 1)
    return buf.value.decode()


def dumps(core) -> bytes:
    """Serialise `core` (a recipe, or a fully concrete core = a recipe with no fields) to the wire format."""
    from . import core as C
    body = _W()
    body.ival(core.nvar)
    body.ival(core.npar)
    for name in ("x0", "lvar", "uvar", "theta", "y0", "lcon", "ucon"):
        segs = core._segs[name]
        body.i32(len(segs))
        for n, src in segs:
            body.ival(n)
            body.i32(src[0])
            if src[0] == SRC_CONST:
                body.f64(src[1])
            elif src[0] == SRC_INLINE:
                body.arr(src[1], np.float64)
            elif src[0] == SRC_FIELD:
                body.i32(src[1])
            else:
                body.i32(src[1])
                body.i32(src[2])
    body.i32(len(core.blocks))
    for name, kind, off, ln, dims in core.blocks:
        body.s(name)
        body.i32(kind)
        body.ival(off)
        body.ival(ln)
        body.i32(len(dims))
        for d in dims:
            body.ival(d)
    body.i32(len(core.patterns))
    user = set()                    # (bivariate, fn) of the user-registered functions the patterns use
    for p in core.patterns:
        nodes, cols, root, target = C.lower_pattern(p, symbolic=True)
        user.update((1 if op == C.OP_BIN else 0, fn) for op, fn, *_ in nodes if op in (C.OP_UN, C.OP_BIN) and fn >= USER_FN_BASE)
        body.i32(p.kind)
        body.i32(root)
        body.i32(target)
        body.i32(p.base)
        body.ival(p.itr.n)
        body.i32(len(nodes))
        for op, fn, a, b, fval, ival in nodes:
            body.i32(op)
            body.i32(fn)
            body.i32(a)
            body.i32(b)
            body.f64(fval)
            body.i64(int(ival))
            tagged = ival if isinstance(ival, TInt) else (fval if isinstance(fval, TFloat) else None)
            body.i32(body.sym(tagged.sym) if tagged is not None else -1)
        body.i32(len(cols))
        for c in cols:
            kind = c[0]
            body.i32(kind)
            if kind == RCOL_RANGE:
                body.ival(c[1])
                body.ival(c[2])
            elif kind in (RCOL_INLINE_I64, RCOL_INLINE_F64):
                body.arr(c[1], np.int64 if kind == RCOL_INLINE_I64 else np.float64)
            elif kind == RCOL_FIELD:
                body.i32(c[1])
            elif kind == RCOL_COL:
                body.i32(c[1])
                body.i32(c[2])
            elif kind == RCOL_AXIS_RANGE:          # start, step, axis length, inner repeat
                for v in c[1:5]:
                    body.ival(v)
            elif kind == RCOL_AXIS_FIELD:          # field, column (-1: array field), axis length, inner repeat
                body.i32(c[1])
                body.i32(c[2])
                body.ival(c[3])
                body.ival(c[4])
            else:                                  # inline axis values, axis length, inner repeat
                body.arr(c[1], np.int64 if kind == RCOL_AXIS_INLINE_I64 else np.float64)
                body.ival(c[2])
                body.ival(c[3])
    if user:                        # trailing section: the registrations, so that the file loads into any process (exahip_recipe.h)
        body.i32(len(user))
        for biv, fn in sorted(user):
            body.i32(biv)
            body.i32(fn)
            for which in range(9):
                body.s(_user_text(biv, fn, which))
    head = _W()
    head.b += MAGIC
    head.i32(1 if core.minimize else 0)
    schema = core.schema
    fields = schema.fields if schema is not None else []
    head.i32(len(fields))
    for f in fields:
        head.s(f["name"])
        head.i32(f["kind"])
        head.i32(f["type"])
        head.i32(len(f["columns"]))
        for cn, ct in f["columns"]:
            head.s(cn)
            head.i32(ct)
    head.i32(len(body.syms))
    for op, a, b in body.syms:
        head.i32(op)
        head.i64(a)
        head.i64(b)
    return bytes(head.b + body.b)
