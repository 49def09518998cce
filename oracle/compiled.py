"""Straight-line C for hess_coord! of a whole model — TEST / BASELINE infrastructure, like everything under oracle/.

Julia compiles every pattern's `shessian!` into straight-line machine code; the tree-walking interpreter in exa_oracle.c
pays per-node overhead that compiled Julia does not, so as a CPU baseline it flatters the GPU.  This module emits, for
every pattern of a model, the C that the reference's recursions unroll to — the forward sweep (register.jl:65-68,
209-266; formula shapes of functionlist.jl) and hrpass0 / hrpass / hdrpass (hessian.jl:16-517) with one
`H[o2 + o2step*I + comp2[cnt] - 1] += value` per contribution on a zero-filled vector (nlp.jl:1913) — compiles it with gcc
-O2 (-ffp-contract=off: Julia does not contract) and times it: the closest thing to `backend = nothing` this container
can produce.  Slot maps (comp2) and offsets come from the interpreter's planner; tests/test_known_answers.py checks the
compiled values against the interpreter.  Only the functions the benchmark models use are covered (+ - * / ^ inv sqrt
exp log sin cos abs2); anything else raises NotImplementedError.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
import tempfile

import numpy as np

OP_CONST_F, OP_CONST_I, OP_DATA, OP_PAR, OP_VAR, OP_UN, OP_BIN, OP_NULLV = range(8)
U_PLUS, U_MINUS, U_INV, U_SQRT, U_ABS2, U_EXP, U_LOG, U_SIN, U_COS = 0, 1, 2, 3, 6, 8, 12, 16, 17
B_ADD, B_SUB, B_MUL, B_DIV, B_POW = range(5)
COL_I64, COL_F64, COL_RANGE = 0, 1, 2


class _Pat:
    def __init__(self, cp):
        self.kind, self.n, self.root, self.target = cp.kind, cp.n, cp.root, cp.target
        self.nodes = [cp.nodes[j] for j in range(cp.n_nodes)]
        self.cols = [cp.cols[c] for c in range(cp.n_cols)]
        self.isconst = {}

    def const(self, k):
        if k not in self.isconst:
            nd = self.nodes[k]
            if nd.op == OP_VAR:
                r = False
            elif nd.op == OP_UN:
                r = self.const(nd.a)
            elif nd.op == OP_BIN:
                r = self.const(nd.a) and self.const(nd.b)
            else:
                r = True
            self.isconst[k] = r
        return self.isconst[k]


class _AD:
    __slots__ = ("kind", "fn", "fixed", "ir", "cir", "l", "r", "id", "key")


class _Emit:
    """One pattern -> body of `static void pK(const double* x, const double* th, double adj0, double* h, long I)`."""

    def __init__(self, pat, pk):
        self.p, self.pk = pat, pk
        self.lines, self.n = [], 0
        self.keys = []
        self.cnt = 0

    def tmp(self, expr, typ="double"):
        name = f"{'k' if typ == 'long' else 't'}{self.n}"
        self.n += 1
        self.lines.append(f"const {typ} {name} = {expr};")
        return name

    # ---- constants / index expressions (graph.jl:305-323): (text, is_int)
    def cexpr(self, k):
        nd = self.p.nodes[k]
        if nd.op == OP_CONST_F or nd.op == OP_NULLV:
            return repr(float(nd.fval)) if np.isfinite(nd.fval) else ("INFINITY" if nd.fval > 0 else "-INFINITY" if nd.fval < 0 else "NAN"), False
        if nd.op == OP_CONST_I:
            return f"{int(nd.ival)}L", True
        if nd.op == OP_DATA:
            col = self.p.cols[nd.a]
            if col.type == COL_RANGE:
                return f"({int(col.start)}L + {int(col.step)}L * I)", True
            return (f"ci{self.pk}_{nd.a}[I]", True) if col.type == COL_I64 else (f"cf{self.pk}_{nd.a}[I]", False)
        if nd.op == OP_PAR:
            i, _ = self.cexpr(nd.a)
            return f"th[{i} - 1]", False
        if nd.op == OP_VAR:
            i, _ = self.cexpr(nd.a)
            return f"x[{i} - 1]", False
        if nd.op == OP_UN:
            a, ai = self.cexpr(nd.a)
            if nd.fn == U_PLUS:
                return a, ai
            if nd.fn == U_MINUS:
                return f"(-{a})", ai
            if nd.fn == U_ABS2:
                return f"({a} * {a})", ai
            return self.unf(nd.fn, a if not ai else f"(double){a}"), False
        if nd.op == OP_BIN:
            a, ai = self.cexpr(nd.a)
            b, bi = self.cexpr(nd.b)
            if ai and bi and nd.fn in (B_ADD, B_SUB, B_MUL):
                return f"({a} {'+-*'[nd.fn]} {b})", True
            fa = a if not ai else f"(double){a}"
            fb = b if not bi else f"(double){b}"
            if nd.fn == B_POW:
                return (f"ipow({fa}, {b})" if bi else f"pow({fa}, {fb})"), False
            if nd.fn in (B_ADD, B_SUB, B_MUL, B_DIV):
                return f"({fa} {'+-*/'[nd.fn]} {fb})", False
        raise NotImplementedError(f"constant node op={nd.op} fn={nd.fn}")

    @staticmethod
    def unf(fn, u):
        t = {U_INV: f"(1.0 / {u})", U_SQRT: f"sqrt({u})", U_EXP: f"exp({u})", U_LOG: f"log({u})", U_SIN: f"sin({u})", U_COS: f"cos({u})"}
        if fn not in t:
            raise NotImplementedError(f"univariate function {fn}")
        return t[fn]

    # ---- AD tree (register.jl:209-266: a constant operand makes the node a unary `Fixed` one)
    def build(self, k):
        p, nd = self.p, self.p.nodes[k]
        a = _AD()
        a.id, a.fn, a.fixed, a.ir, a.cir, a.l, a.r, a.key = self.n, nd.fn, 0, k, -1, None, None, -1
        self.n += 1
        if nd.op == OP_NULLV:
            a.kind = "null"
        elif p.const(k):
            a.kind = "const"
        elif nd.op == OP_VAR:
            a.kind, a.ir = "var", nd.a
            txt = self.keytext(nd.a)
            if txt not in self.keys:
                self.keys.append(txt)
            a.key = self.keys.index(txt)
        elif nd.op == OP_UN:
            a.kind, a.l = "un", self.build(nd.a)
        else:
            ca = p.const(nd.a) and p.nodes[nd.a].op != OP_NULLV
            cb = p.const(nd.b) and p.nodes[nd.b].op != OP_NULLV
            if cb:
                a.kind, a.fixed, a.cir, a.l = "un", 2, nd.b, self.build(nd.a)
            elif ca:
                a.kind, a.fixed, a.cir, a.l = "un", 1, nd.a, self.build(nd.b)
            else:
                a.kind, a.l, a.r = "bin", self.build(nd.a), self.build(nd.b)
        return a

    def keytext(self, k):
        nd = self.p.nodes[k]
        if nd.op == OP_CONST_I:
            return f"i{nd.ival}"
        if nd.op == OP_CONST_F:
            return f"f{nd.fval!r}"
        if nd.op == OP_DATA:
            return f"d{nd.a}"
        if nd.op == OP_UN:
            return f"u{nd.fn}({self.keytext(nd.a)})"
        if nd.op == OP_BIN:
            return f"b{nd.fn}({self.keytext(nd.a)},{self.keytext(nd.b)})"
        return f"o{nd.op}"

    # ---- bivariate table (functionlist.jl:71-81), shapes kept: (f, d1, d2, d11, d12, d22); None = literal zero
    def bin_table(self, fn, x1, x2, x2_int=None):
        if fn == B_ADD:
            return f"{x1} + {x2}", "1.0", "1.0", None, None, None
        if fn == B_SUB:
            return f"{x1} - {x2}", "1.0", "-1.0", None, None, None
        if fn == B_MUL:
            return f"{x1} * {x2}", x2, x1, None, "1.0", None
        if fn == B_DIV:
            return (f"{x1} / {x2}", f"1.0 / {x2}", f"(-{x1}) / ({x2} * {x2})", None, f"-1.0 / ({x2} * {x2})",
                    f"(2.0 * {x1}) / ({x2} * {x2} * {x2})")
        if fn == B_POW:
            if x2_int is not None:      # Base.^(::Float64, ::Int): x^(n-1), x^(n-2) by repeated multiplication
                n = x2_int
                return (f"ipow({x1}, {n}L)", f"{float(n)!r} * ipow({x1}, {n - 1}L)", f"log({x1}) * ipow({x1}, {n}L)",
                        f"{float((n - 1) * n)!r} * ipow({x1}, {n - 2}L)", None, None)
            return (f"pow({x1}, {x2})", f"{x2} * pow({x1}, -1.0 + {x2})", f"log({x1}) * pow({x1}, {x2})",
                    f"(-1.0 + {x2}) * {x2} * pow({x1}, -2.0 + {x2})", f"pow({x1}, -1.0 + {x2}) + {x2} * pow({x1}, -1.0 + {x2}) * log({x1})",
                    f"(log({x1}) * log({x1})) * pow({x1}, {x2})")
        raise NotImplementedError(f"bivariate function {fn}")

    def forward(self, a):
        v = f"n{a.id}"
        if a.kind == "null":
            self.lines.append(f"const double {v}_x = {self.p.nodes[a.ir].fval!r};")
        elif a.kind == "const":
            e, isint = self.cexpr(a.ir)
            self.lines.append(f"const double {v}_x = {'(double)' if isint else ''}{e};")
        elif a.kind == "var":
            i, _ = self.cexpr(a.ir)
            self.lines.append(f"const long {v}_i = {i};")
            self.lines.append(f"const double {v}_x = x[{v}_i - 1];")
        elif a.kind == "un" and a.fixed == 0:
            self.forward(a.l)
            u = f"n{a.l.id}_x"
            fn = a.fn
            if fn == U_PLUS:
                f, d, dd = u, "1.0", "0.0"
            elif fn == U_MINUS:
                f, d, dd = f"-{u}", "-1.0", "0.0"
            elif fn == U_ABS2:
                f, d, dd = f"{u} * {u}", f"2.0 * {u}", "2.0"
            elif fn == U_EXP:
                self.lines.append(f"const double {v}_e = exp({u});")
                f, d, dd = f"{v}_e", f"{v}_e", f"{v}_e"
            elif fn in (U_SIN, U_COS):
                self.lines.append(f"const double {v}_s = sin({u}), {v}_c = cos({u});")
                f, d, dd = (f"{v}_s", f"{v}_c", f"-{v}_s") if fn == U_SIN else (f"{v}_c", f"-{v}_s", f"-{v}_c")
            elif fn == U_INV:
                f, d, dd = f"1.0 / {u}", f"-1.0 / ({u} * {u})", f"2.0 / ({u} * {u} * {u})"
            elif fn == U_SQRT:
                f, d, dd = f"sqrt({u})", f"1.0 / (2.0 * sqrt({u}))", f"-1.0 / (4.0 * (sqrt({u}) * sqrt({u}) * sqrt({u})))"
            elif fn == U_LOG:
                f, d, dd = f"log({u})", f"1.0 / {u}", f"-1.0 / ({u} * {u})"
            else:
                raise NotImplementedError(f"univariate function {fn}")
            self.lines.append(f"const double {v}_x = {f}, {v}_y1 = {d}, {v}_h11 = {dd};")
        elif a.kind == "un":
            self.forward(a.l)
            c, cint = self.cexpr(a.cir)
            lit_int = int(self.p.nodes[a.cir].ival) if self.p.nodes[a.cir].op == OP_CONST_I else None
            cname = self.tmp(f"(double){c}" if cint else c)
            u = f"n{a.l.id}_x"
            if a.fixed == 2:      # v OP c: d1, d11
                f, d1, _, d11, _, _ = self.bin_table(a.fn, u, cname, lit_int if a.fn == B_POW else None)
                d, dd = d1, d11
            else:                 # c OP v: d2, d22
                f, _, d2, _, _, d22 = self.bin_table(a.fn, cname, u)
                d, dd = d2, d22
            self.lines.append(f"const double {v}_x = {f}, {v}_y1 = {d}, {v}_h11 = {dd or '0.0'};")
        else:
            self.forward(a.l)
            self.forward(a.r)
            f, d1, d2, d11, d12, d22 = self.bin_table(a.fn, f"n{a.l.id}_x", f"n{a.r.id}_x")
            self.lines.append(f"const double {v}_x = {f}, {v}_y1 = {d1}, {v}_y2 = {d2}, {v}_h11 = {d11 or '0.0'}, "
                              f"{v}_h12 = {d12 or '0.0'}, {v}_h22 = {d22 or '0.0'};")

    # ---- reverse sweeps: one `+=` per contribution, hessian.jl order
    def put(self, val):
        self.lines.append(f"h[CMP{self.pk}[{self.cnt}]] += {val};")
        self.cnt += 1

    def hdrpass(self, t1, t2, adj):
        k1, k2 = t1.kind, t2.kind
        if k1 in ("null", "const") or k2 in ("null", "const"):
            return
        y = lambda t, w: f"n{t.id}_y{w}"       # noqa: E731
        if k1 == "un" and k2 == "un":
            return self.hdrpass(t1.l, t2.l, self.tmp(f"{adj} * {y(t1, 1)} * {y(t2, 1)}"))
        if k1 == "var" and k2 == "un":
            return self.hdrpass(t1, t2.l, self.tmp(f"{adj} * {y(t2, 1)}"))
        if k1 == "un" and k2 == "var":
            return self.hdrpass(t1.l, t2, self.tmp(f"{adj} * {y(t1, 1)}"))
        if k1 == "bin" and k2 == "bin":
            for a, wa in ((t1.l, 1), (t1.r, 2)):
                for b, wb in ((t2.l, 1), (t2.r, 2)):
                    self.hdrpass(a, b, self.tmp(f"{adj} * {y(t1, wa)} * {y(t2, wb)}"))
            return
        if k1 == "un" and k2 == "bin":
            self.hdrpass(t1.l, t2.l, self.tmp(f"{adj} * {y(t1, 1)} * {y(t2, 1)}"))
            self.hdrpass(t1.l, t2.r, self.tmp(f"{adj} * {y(t1, 1)} * {y(t2, 2)}"))
            return
        if k1 == "bin" and k2 == "un":
            self.hdrpass(t1.l, t2.l, self.tmp(f"{adj} * {y(t1, 1)} * {y(t2, 1)}"))
            self.hdrpass(t1.r, t2.l, self.tmp(f"{adj} * {y(t1, 2)} * {y(t2, 1)}"))
            return
        if k1 == "var" and k2 == "bin":
            self.hdrpass(t1, t2.l, self.tmp(f"{adj} * {y(t2, 1)}"))
            self.hdrpass(t1, t2.r, self.tmp(f"{adj} * {y(t2, 2)}"))
            return
        if k1 == "bin" and k2 == "var":
            self.hdrpass(t1.l, t2, self.tmp(f"{adj} * {y(t1, 1)}"))
            self.hdrpass(t1.r, t2, self.tmp(f"{adj} * {y(t1, 2)}"))
            return
        self.put(f"(n{t1.id}_i == n{t2.id}_i ? 2.0 * {adj} : {adj})")      # hessian.jl:251-268

    def hrpass(self, t, adj, adj2):
        if t.kind in ("null", "const"):
            return
        v = f"n{t.id}"
        if t.kind == "un":
            return self.hrpass(t.l, self.tmp(f"{adj} * {v}_y1"), self.tmp(f"{adj2} * ({v}_y1 * {v}_y1) + {adj} * {v}_h11"))
        if t.kind == "bin":
            cross = self.tmp(f"{adj2} * {v}_y1 * {v}_y2 + {adj} * {v}_h12")
            self.hrpass(t.l, self.tmp(f"{adj} * {v}_y1"), self.tmp(f"{adj2} * ({v}_y1 * {v}_y1) + {adj} * {v}_h11"))
            self.hrpass(t.r, self.tmp(f"{adj} * {v}_y2"), self.tmp(f"{adj2} * ({v}_y2 * {v}_y2) + {adj} * {v}_h22"))
            return self.hdrpass(t.l, t.r, cross)
        self.put(adj2)

    def hrpass0(self, t, adj, adj2):
        v = f"n{t.id}"
        if t.kind == "un" and t.fixed:
            if t.fn == B_MUL:
                return self.hrpass0(t.l, self.tmp(f"{adj} * {v}_y1"), self.tmp(f"{adj2} * ({v}_y1 * {v}_y1)"))
            if t.fn == B_ADD:
                return self.hrpass0(t.l, adj, adj2)
            if t.fn == B_SUB:
                return self.hrpass0(t.l, self.tmp(f"-{adj}") if t.fixed == 1 else adj, adj2)
        elif t.kind == "un":
            if t.fn == U_PLUS:
                return self.hrpass0(t.l, adj, adj2)
            if t.fn == U_MINUS:
                return self.hrpass0(t.l, self.tmp(f"-{adj}"), adj2)
        elif t.kind == "bin":
            if t.fn == B_ADD:
                self.hrpass0(t.l, adj, adj2)
                return self.hrpass0(t.r, adj, adj2)
            if t.fn == B_SUB:
                self.hrpass0(t.l, adj, adj2)
                return self.hrpass0(t.r, self.tmp(f"-{adj}"), adj2)
        elif t.kind == "var":
            return
        self.hrpass(t, adj, adj2)


def emit_source(ir, o):
    """C source of hess_coord! for the whole model `ir` (exahip ModelIR); `o` = oracle.OracleModel(ir) for the layout."""
    src = ["#include <math.h>", "#include <stdint.h>", "#include <string.h>",
           "static inline double ipow(double x, long n) { if (n == 0) return 1.0; if (n < 0) { x = 1.0 / x; n = -n; } double y = 1.0; "
           "while (n > 1) { if (n & 1) y *= x; x *= x; n >>= 1; } return x * y; }"]
    drivers, setters = [], []
    for k in range(o.npatterns):
        info = o.pattern_info(k)
        if info["o2step"] == 0 or info["n"] == 0:
            continue
        pat = _Pat(ir.patterns[k])
        em = _Emit(pat, k)
        root = em.build(pat.root)
        em.forward(root)
        em.hrpass0(root, "adj0", "zero")      # hessian.jl:714-717: the second-order adjoint starts as a RUN-TIME zero
        comp = o.pattern_comp(k, 2)
        assert em.cnt == len(comp), (k, em.cnt, len(comp))
        for c, col in enumerate(pat.cols):
            if col.type == COL_I64:
                src.append(f"static const long *ci{k}_{c};")
                setters.append((k, c, "i"))
            elif col.type == COL_F64:
                src.append(f"static const double *cf{k}_{c};")
                setters.append((k, c, "f"))
        src.append(f"static const int CMP{k}[] = {{{', '.join(str(c - 1) for c in comp)}}};")
        src.append(f"static void p{k}(const double *x, const double *th, double adj0, double zero, double *h, long I) {{")
        src += ["    " + ln for ln in em.lines]
        src.append("}")
        if pat.kind == 0:
            adj = "sigma"
        elif pat.kind == 1:
            adj = f"y[{info['o0']} + I]"
        else:
            t, _ = em.cexpr(pat.target)
            adj = f"y[{info['o0']} + {t} - 1]"
        drivers.append(f"    _Pragma(\"omp parallel for schedule(static) num_threads(nth)\")\n"
                       f"    for (long I = 0; I < {info['n']}L; I++) p{k}(x, th, {adj}, zero, H + {info['o2']}L + {info['o2step']}L * I, I);")
    src.append("void set_col(int k, int c, const void *p) {")
    for k, c, t in setters:
        src.append(f"    if (k == {k} && c == {c}) c{t}{k}_{c} = p;")
    src.append("}")
    src.append(f"void hess_all(const double *x, const double *y, const double *th, double sigma, double *H, int nth, volatile double *zero_src) {{\n"
               f"    const double zero = *zero_src;\n"
               f"    _Pragma(\"omp parallel for schedule(static) num_threads(nth)\")\n    for (long q = 0; q < {o.nnzh}L; q++) H[q] = 0.0;")
    src += drivers
    src.append("}")
    return "\n".join(src) + "\n"


class CompiledHess:
    """gcc-compiled hess_coord! of a model; `__call__(x, y, sigma, out, threads)`."""

    def __init__(self, ir, o):
        self.src = emit_source(ir, o)
        self.nnzh = o.nnzh
        # a PRIVATE directory, verified before anything in it is loaded: $EXAHIP_CACHE_DIR or ~/.cache/exaoracle (0700, owned
        # by this user, not a symlink) — never a predictable path under /tmp
        base = os.environ.get("EXAHIP_CACHE_DIR") or os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "exaoracle")
        d = os.path.join(base, "compiled_oracle")
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.lstat(d)
        import stat
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
            raise RuntimeError(f"{d} is not a private directory of this user: refusing to load shared objects from it")
        # -march=native like Julia's JIT (which targets the host CPU); -ffp-contract=off: no FMA contraction, like the reference
        flags = ["-O2", "-march=native", "-fPIC", "-shared", "-std=gnu11", "-ffp-contract=off", "-fopenmp", "-w"]
        tag = hashlib.sha256((self.src + " ".join(flags)).encode()).hexdigest()[:24]
        so = os.path.join(d, tag + ".so")
        if not os.path.exists(so):
            fd, csrc = tempfile.mkstemp(suffix=".c", dir=d)
            with os.fdopen(fd, "w") as fh:
                fh.write(self.src)
            tmp = csrc[:-2] + ".so"
            subprocess.check_call(["gcc"] + flags + ["-o", tmp, csrc, "-lm"])
            os.replace(tmp, so)
            os.unlink(csrc)
        self.lib = ctypes.CDLL(so)
        vp = ctypes.c_void_p
        self.lib.set_col.argtypes = [ctypes.c_int, ctypes.c_int, vp]
        self.lib.hess_all.argtypes = [vp, vp, vp, ctypes.c_double, vp, ctypes.c_int, vp]
        self._keep = []
        for k in range(ir.desc.n_patterns):
            cp = ir.patterns[k]
            for c in range(cp.n_cols):
                if cp.cols[c].type != COL_RANGE:
                    self.lib.set_col(k, c, cp.cols[c].data)
        self._ir = ir
        self._zero = np.zeros(1)
        self._theta = np.ascontiguousarray(ir.theta0 if getattr(ir, "theta0", None) is not None else np.zeros(1))

    def __call__(self, x, y, sigma, out=None, threads=1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y if len(y) else np.zeros(1), dtype=np.float64)
        if out is None:
            out = np.empty(self.nnzh)
        self.lib.hess_all(x.ctypes.data, y.ctypes.data, self._theta.ctypes.data, float(sigma), out.ctypes.data, int(threads), self._zero.ctypes.data)
        return out
