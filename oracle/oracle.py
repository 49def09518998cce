"""ctypes binding of the TEST ORACLE (oracle/libexaoracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libexaoracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("exa_oracle.c", "exa_special.h", "exa_quad.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        vp, i64, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double
        L.ora_new.restype = vp
        L.ora_new.argtypes = [vp]
        L.ora_free.argtypes = [vp]
        for f in ("ora_nvar", "ora_ncon", "ora_nnzj", "ora_nnzh", "ora_nnzg"):
            getattr(L, f).restype = i64
            getattr(L, f).argtypes = [vp]
        L.ora_npatterns.argtypes = [vp]
        L.ora_set_threads.argtypes = [vp, ctypes.c_int]
        L.ora_set_shard.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        L.ora_set_theta.argtypes = [vp, i64, vp, i64]
        L.ora_pattern_info.argtypes = [vp, ctypes.c_int, vp]
        L.ora_pattern_comp.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
        L.ora_meta.argtypes = [vp] * 6
        L.ora_obj.restype = dbl
        L.ora_obj.argtypes = [vp, vp]
        L.ora_cons.argtypes = [vp, vp, vp]
        L.ora_cons_quad.argtypes = [vp, vp, vp, vp]
        L.ora_grad.argtypes = [vp, vp, vp]
        L.ora_sgrad.argtypes = [vp, vp, vp]
        L.ora_jac.argtypes = [vp, vp, vp]
        L.ora_hess.argtypes = [vp, vp, vp, dbl, vp]
        L.ora_jprod.argtypes = [vp, vp, vp, vp]
        L.ora_jtprod.argtypes = [vp, vp, vp, vp]
        L.ora_hprod.argtypes = [vp, vp, vp, vp, dbl, vp]
        L.ora_jac_structure.argtypes = [vp, vp, vp]
        L.ora_hess_structure.argtypes = [vp, vp, vp]
        L.ora_lv_hess_compiled.argtypes = [i64, vp, vp, dbl, vp, ctypes.c_int]
        L.ora_un_table.argtypes = [ctypes.c_int, dbl, vp]
        L.ora_bin_table.argtypes = [ctypes.c_int, dbl, dbl, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data if a is not None and a.size else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleModel:
    """CPU evaluation of a model given as ModelIR (the same table the product consumes)."""

    def __init__(self, ir, threads=1):
        self._L = lib()
        self._ir = ir
        self._h = self._L.ora_new(ctypes.addressof(ir.desc))
        self.nvar = self._L.ora_nvar(self._h)
        self.ncon = self._L.ora_ncon(self._h)
        self.nnzj = self._L.ora_nnzj(self._h)
        self.nnzh = self._L.ora_nnzh(self._h)
        self.nnzg = self._L.ora_nnzg(self._h)
        self.npatterns = self._L.ora_npatterns(self._h)
        self.set_threads(threads)

    def __del__(self):
        try:
            if self._h:
                self._L.ora_free(self._h)
                self._h = None
        except Exception:
            pass

    def set_threads(self, n):
        self._L.ora_set_threads(self._h, int(n))

    def set_shard(self, rank, world):
        self._L.ora_set_shard(self._h, int(rank), int(world))

    def set_value(self, offset, vals):
        v = _f64(vals)
        self._L.ora_set_theta(self._h, int(offset), _p(v), v.size)

    def pattern_info(self, k):
        out = np.zeros(9, dtype=np.int64)
        self._L.ora_pattern_info(self._h, k, _p(out))
        names = ["kind", "n", "o0", "o1", "o2", "o1step", "o2step", "n1", "n2"]
        return dict(zip(names, out.tolist()))

    def pattern_comp(self, k, order):
        info = self.pattern_info(k)
        out = np.zeros(max(1, info["n1"] if order == 1 else info["n2"]), dtype=np.int32)
        self._L.ora_pattern_comp(self._h, k, order, _p(out))
        return out[: info["n1"] if order == 1 else info["n2"]].tolist()

    def meta(self):
        x0, lv, uv = np.zeros(self.nvar), np.zeros(self.nvar), np.zeros(self.nvar)
        lc, uc = np.zeros(max(1, self.ncon)), np.zeros(max(1, self.ncon))
        self._L.ora_meta(self._h, _p(x0), _p(lv), _p(uv), _p(lc), _p(uc))
        return x0, lv, uv, lc[: self.ncon], uc[: self.ncon]

    def obj(self, x):
        x = _f64(x)
        return self._L.ora_obj(self._h, _p(x))

    def cons(self, x):
        x = _f64(x)
        c = np.empty(self.ncon)
        self._L.ora_cons(self._h, _p(x), _p(c))
        return c

    def cons_quad(self, x):
        """(c, e): cons_nln! evaluated in __float128 (oracle/exa_quad.h) and rounded to double, and per row the first-order running
        error bound of a double-precision evaluation, in units of eps — the arbiter for rows that cancel.  None when a pattern
        uses a function without a quad restatement."""
        x = _f64(x)
        c, mag = np.empty(self.ncon), np.empty(self.ncon)
        ok = self._L.ora_cons_quad(self._h, _p(x), _p(c), _p(mag))
        return (c, mag) if ok else None

    def grad(self, x):
        x = _f64(x)
        g = np.empty(self.nvar)
        self._L.ora_grad(self._h, _p(x), _p(g))
        return g

    def sgrad(self, x):
        x = _f64(x)
        g = np.empty(self.nnzg)
        self._L.ora_sgrad(self._h, _p(x), _p(g))
        return g

    def jac_coord(self, x, out=None):
        x = _f64(x)
        v = np.empty(self.nnzj) if out is None else out
        self._L.ora_jac(self._h, _p(x), _p(v))
        return v

    def hess_coord(self, x, y, obj_weight=1.0, out=None):
        x, y = _f64(x), _f64(y)
        v = np.empty(self.nnzh) if out is None else out
        self._L.ora_hess(self._h, _p(x), _p(y), float(obj_weight), _p(v))
        return v

    def jprod(self, x, v):
        x, v = _f64(x), _f64(v)
        out = np.empty(self.ncon)
        self._L.ora_jprod(self._h, _p(x), _p(v), _p(out))
        return out

    def jtprod(self, x, v):
        x, v = _f64(x), _f64(v)
        out = np.empty(self.nvar)
        self._L.ora_jtprod(self._h, _p(x), _p(v), _p(out))
        return out

    def hprod(self, x, y, v, obj_weight=1.0):
        x, y, v = _f64(x), _f64(y), _f64(v)
        out = np.empty(self.nvar)
        self._L.ora_hprod(self._h, _p(x), _p(y), _p(v), float(obj_weight), _p(out))
        return out

    def jac_structure(self):
        r, c = np.zeros(self.nnzj, dtype=np.int64), np.zeros(self.nnzj, dtype=np.int64)
        self._L.ora_jac_structure(self._h, _p(r), _p(c))
        return r, c

    def hess_structure(self):
        r, c = np.zeros(self.nnzh, dtype=np.int64), np.zeros(self.nnzh, dtype=np.int64)
        self._L.ora_hess_structure(self._h, _p(r), _p(c))
        return r, c


def un_table(fn_id, x):
    out = np.zeros(3)
    lib().ora_un_table(int(fn_id), float(x), _p(out))
    return out


def bin_table(fn_id, x1, x2):
    out = np.zeros(6)
    lib().ora_bin_table(int(fn_id), float(x1), float(x2), _p(out))
    return out


def lv_hess_compiled(N, x, y, sigma, out=None, threads=1):
    """Hand-specialised LV hess_coord! (proxy for Julia-compiled `backend = nothing`), see exa_oracle.c."""
    x, y = _f64(x), _f64(y)
    if out is None:
        out = np.empty(9 * N - 15)
    lib().ora_lv_hess_compiled(int(N), _p(x), _p(y), float(sigma), _p(out), int(threads))
    return out
