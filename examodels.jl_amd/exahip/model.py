"""ExaModel — the NLPModels oracle surface over libexahip.so.

Host-side mirror of the reference's `ExaModel <: AbstractNLPModel` (src/nlp.jl:704-716, 765-798) restricted to the
evaluation surface: obj, cons (cons_nln!), grad (grad!), jac_structure, jac_coord (jac_coord!), hess_structure,
hess_coord (hess_coord!), set_value (set_value!).  Method names and argument meaning follow NLPModels; the `!`
in-place forms are expressed through the `out=` argument.  Inputs may be numpy arrays (staged through the *_host
entry points, the WrapperNLPModel role of src/utils.jl:159-208) or torch tensors already resident in HBM
(device-pointer entry points, asynchronous on torch's current stream).
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace

import numpy as np

from . import capi
from .core import ExaCore, ModelIR

_WHICH = {"obj": 0, "grad": 1, "cons": 2, "jac": 3, "hess": 4, "launch": 5}


class Recipe:
    """A model whose sizes and data arrive at instantiation (exahip/recipe.py; include/exahip_recipe.h).
    Built from a recipe core (`ExaCore(examples=...)`), from an ordinary core (a fixed model: the same bytes are the
    model file format) or from serialized bytes."""

    def __init__(self, src):
        import json
        self._L = capi.lib()
        self.bytes = bytes(src) if isinstance(src, (bytes, bytearray)) else src.to_recipe()
        self.id = self._L.exa_recipe_load(self.bytes, len(self.bytes))
        if self.id <= 0:
            raise capi.ExaHipError("exa_recipe_load: " + self._L.exa_last_error().decode(errors="replace"))
        self.minimize = bool(int.from_bytes(self.bytes[8:12], "little"))
        self.nargs = self._L.exa_recipe_nargs(self.id)
        self.argtype = self._text(self._L.exa_recipe_argtype)
        self.schema_json = self._text(self._L.exa_recipe_schema)
        self.fields = json.loads(self.schema_json)["fields"]

    def __del__(self):
        try:
            if self.id > 0:
                self._L.exa_recipe_free(self.id)
                self.id = 0
        except Exception:
            pass

    def _text(self, fn):
        n = fn(self.id, None, 0)
        buf = ctypes.create_string_buffer(max(1, n))
        fn(self.id, buf, n)
        return buf.raw[:n].decode()

    def save(self, path):
        with open(path, "wb") as fh:
            fh.write(self.bytes)

    @classmethod
    def load(cls, path):
        with open(path, "rb") as fh:
            return cls(fh.read())

    def _flatten(self, values):
        """`ExaModel(core, n, (v0 = .., lo = ..), tab)` or the consumers' flat spelling `CModel(lib, n, v0, lo, tab)`:
        a dict whose keys are the next schema fields is spread over them."""
        flat = []
        for v in values:
            k = len(flat)
            if isinstance(v, dict) and list(v) == [f["name"] for f in self.fields[k:k + len(v)]]:
                flat.extend(v.values())
            else:
                flat.append(v)
        if len(flat) != len(self.fields):
            raise TypeError(f"this recipe instantiates from {len(self.fields)} values ({self.argtype}), got {len(flat)}")
        return flat

    def _instantiate(self, values, device=True):
        from .recipe import _table_columns
        L = self._L
        if not self.fields:
            if values:
                raise TypeError("a fixed model takes no instantiation values")
            mid = (L.exa_recipe_new if device else L.exa_recipe_plan)(self.id, 0)
        else:
            b = L.exa_data_begin(self.id)
            if b <= 0:
                raise capi.ExaHipError("exa_data_begin failed")
            try:
                for f, v in zip(self.fields, self._flatten(values)):
                    name = f["name"].encode()
                    if f["kind"] == "scalar":
                        st = L.exa_set_scalar_i64(b, name, int(v)) if f["type"] == "i64" else L.exa_set_scalar_f64(b, name, float(v))
                    elif f["kind"] == "array":
                        a = np.ascontiguousarray(v, dtype=np.int64 if f["type"] == "i64" else np.float64)
                        fn = L.exa_set_array_i64 if f["type"] == "i64" else L.exa_set_array_f64
                        st = fn(b, name, a.ctypes.data, a.size)
                    else:
                        cols = v if isinstance(v, dict) else _table_columns(v)
                        st = 0
                        for c in f["columns"]:
                            a = np.ascontiguousarray(cols[c["name"]], dtype=np.int64 if c["type"] == "i64" else np.float64)
                            fn = L.exa_set_col_i64 if c["type"] == "i64" else L.exa_set_col_f64
                            st = st or fn(b, name, c["name"].encode(), a.ctypes.data, a.size)
                    if st:
                        raise capi.ExaHipError(f"builder: field {f['name']!r} rejected (status {st})")
                if L.exa_data_ready(b) != 1:
                    raise capi.ExaHipError("builder: data not ready (a table's columns differ in length?)")
                mid = (L.exa_new_from_data if device else L.exa_plan_from_data)(b)
            finally:
                L.exa_data_free(b)
        if mid <= 0:
            raise capi.ExaHipError("recipe instantiation failed: " + L.exa_last_error().decode(errors="replace"))
        return mid

    def instantiate(self, *values, device=True):
        return ExaModel(self, *values, device=device)

    def matches(self, core, *values):
        """True iff this recipe instantiated at `values` is, entry for entry, the pattern table of `core` (the same
        model built directly at those values).  The reference checks a recipe by instantiating its example before
        compiling (ExaModelsCompiler.jl:153-156); here a SECOND, different set of values is the meaningful check:
        Python cannot stop `1.0 / N` or `2.5 * N` from silently baking the example's N into a real constant (float
        arithmetic accepts the tagged int as a plain one), and this is what catches it.  No device needed."""
        a = ExaModel(self, *values, device=False)
        b = ExaModel(core, device=False)
        return tables_equal(a, b)


def _view(ptr, n, ctype):
    if not ptr or n <= 0:
        return np.zeros(0, dtype=ctype)
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(ctype))), shape=(int(n),))


def tables_equal(a, b):
    """Entry-for-entry comparison of the pattern tables of two models planned without a device."""
    from .core import CNode, COL_RANGE, PAT_CON
    da, db = a.describe().desc, b.describe().desc
    if (da.nvar, da.npar, da.n_patterns, da.minimize) != (db.nvar, db.npar, db.n_patterns, db.minimize):
        return False
    ncon = 0
    for k in range(da.n_patterns):
        pa, pb = da.patterns[k], db.patterns[k]
        if (pa.kind, pa.n_nodes, pa.root, pa.target, pa.base, pa.n_cols, pa.n) != \
                (pb.kind, pb.n_nodes, pb.root, pb.target, pb.base, pb.n_cols, pb.n):
            return False
        if ctypes.string_at(pa.nodes, pa.n_nodes * ctypes.sizeof(CNode)) != ctypes.string_at(pb.nodes, pb.n_nodes * ctypes.sizeof(CNode)):
            return False
        for c in range(pa.n_cols):
            ca, cb = pa.cols[c], pb.cols[c]
            if ca.type != cb.type:
                return False
            if ca.type == COL_RANGE:
                if (ca.start, ca.step) != (cb.start, cb.step):
                    return False
            elif ctypes.string_at(ca.data, 8 * pa.n) != ctypes.string_at(cb.data, 8 * pb.n):
                return False
        if pa.kind == PAT_CON:
            ncon += pa.n
    # a NULL vector stands for its default constant (include/exahip_ir.h)
    for name, n, dflt in (("x0", da.nvar, 0.0), ("lvar", da.nvar, -np.inf), ("uvar", da.nvar, np.inf), ("theta0", da.npar, 0.0),
                          ("y0", ncon, 0.0), ("lcon", ncon, 0.0), ("ucon", ncon, 0.0)):
        va = _view(getattr(da, name), n, np.float64) if getattr(da, name) else np.full(int(n), dflt)
        vb = _view(getattr(db, name), n, np.float64) if getattr(db, name) else np.full(int(n), dflt)
        if not np.array_equal(va, vb):
            return False
    return True


def _is_torch(a):
    return type(a).__module__.startswith("torch")


class Meta:
    """Sizes as plain attributes; x0 / lvar / uvar / lcon / ucon are fetched on first use (five vectors of nvar or ncon
    doubles — 4 GB at N = 1e8 — that an evaluation loop never needs)."""
    _VECTORS = ("x0", "lvar", "uvar", "lcon", "ucon")

    def __init__(self, loader, **scalars):
        self.__dict__.update(scalars)
        self.__dict__["_loader"] = loader

    def __getattr__(self, name):
        if name in Meta._VECTORS:
            self.__dict__.update(self._loader(name))
            return self.__dict__[name]
        raise AttributeError(name)

    def copy(self):
        m = Meta(self._loader)
        m.__dict__.update(self.__dict__)
        return m


class ExaModel:
    def __init__(self, core, *args, device=True):
        """device=True: exa_new_from_table (needs an MI355X; raises otherwise — no CPU fallback).
        device=False: exa_plan_only — layout + generated source only, callbacks raise.
        `ExaModel(recipe_core, args...)` instantiates a recipe (nlp.jl:809-863) through the builder ABI."""
        self._L = capi.lib()
        self.id = 0
        if isinstance(core, Recipe) or getattr(core, "schema", None) is not None:
            rec = core if isinstance(core, Recipe) else Recipe(core)
            self.ir = None
            self.id = rec._instantiate(args, device)
            self._minimize = rec.minimize
        else:
            if args:
                raise TypeError("instantiation arguments given, but the core is not a recipe")
            self.ir = core if isinstance(core, ModelIR) else core.to_ir()
            idc = ctypes.c_int(0)
            fn = self._L.exa_new_from_table if device else self._L.exa_plan_only
            capi.check(fn(ctypes.addressof(self.ir.desc), ctypes.byref(idc)), "exa_new_from_table" if device else "exa_plan_only")
            self.id = idc.value
            self._minimize = bool(self.ir.desc.minimize)
        self.device = device
        L = self._L
        nvar, ncon = L.exa_nvar64(self.id), L.exa_ncon64(self.id)

        def load_vectors(name):
            if self.ir is not None and hasattr(self.ir, "_vecs"):
                return {name: getattr(self.ir, name)}        # the array the table was built from (the library holds a copy)
            x0, lv, uv = np.empty(nvar), np.empty(nvar), np.empty(nvar)
            lc, uc = np.empty(max(1, ncon)), np.empty(max(1, ncon))
            capi.check(L.exa_meta(self.id, x0.ctypes.data, lv.ctypes.data, uv.ctypes.data, lc.ctypes.data, uc.ctypes.data), "exa_meta")
            return {"x0": x0, "lvar": lv, "uvar": uv, "lcon": lc[:ncon], "ucon": uc[:ncon]}

        self.meta = Meta(load_vectors, nvar=nvar, ncon=ncon, nnzj=L.exa_nnzj64(self.id), nnzh=L.exa_nnzh64(self.id),
                         nnzg=L.exa_nnzg64(self.id), minimize=self._minimize)
        self._stream = None
        self._coo_local = False

    def describe(self):
        """Pattern-table view (exa_model_desc_t) of a model planned without a device — what a recipe became."""
        from .core import CModelDesc
        d = CModelDesc()
        capi.check(self._L.exa_describe(self.id, ctypes.addressof(d)), "exa_describe")
        return SimpleNamespace(desc=d, owner=self)

    # ---- named blocks (cnlp P_nblocks / P_block_name / P_block, ExaModelsCompiler.jl:1476-1510) -----------------
    def blocks(self):
        out = []
        for k in range(max(0, self._L.exa_nblocks(self.id))):
            n = self._L.exa_block_name(self.id, k, None, 0)
            buf = ctypes.create_string_buffer(max(1, n))
            self._L.exa_block_name(self.id, k, buf, n)
            rec = (ctypes.c_int * 72)()
            capi.check(self._L.exa_block(self.id, k, rec), "exa_block")
            out.append(SimpleNamespace(name=buf.raw[:n].decode(), kind=rec[0], offset=rec[1], length=rec[2],
                                       dims=[rec[4 + j] for j in range(rec[3])]))
        return out

    def get_value_block(self, k):
        b = self.blocks()[k]
        v = np.empty(b.length)
        capi.check(self._L.exa_get_value_block(self.id, k, v.ctypes.data, v.size), "exa_get_value_block")
        return v

    def set_value_block(self, k, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        capi.check(self._L.exa_set_value_block(self.id, k, v.ctypes.data, v.size), "exa_set_value_block")

    def __del__(self):
        try:
            if self.id:
                self._L.exa_free(self.id)
                self.id = 0
        except Exception:
            pass

    # ---- layout introspection -----------------------------------------------------------------------------
    @property
    def npatterns(self):
        return self._L.exa_npatterns(self.id)

    def pattern_info(self, k):
        out = np.zeros(9, dtype=np.int64)
        capi.check(self._L.exa_pattern_info(self.id, k, out.ctypes.data), "exa_pattern_info")
        return dict(zip(["kind", "n", "o0", "o1", "o2", "o1step", "o2step", "n1", "n2"], out.tolist()))

    def pattern_comp(self, k, order):
        info = self.pattern_info(k)
        n = info["n1"] if order == 1 else info["n2"]
        out = np.zeros(max(1, n), dtype=np.int32)
        capi.check(self._L.exa_pattern_comp(self.id, k, order, out.ctypes.data), "exa_pattern_comp")
        return out[:n].tolist()

    def locality_order(self, k):
        """exa_locality_order: a stable order of pattern k's data points by the smallest variable they reach (plan-only handles)"""
        out = np.zeros(max(1, self.pattern_info(k)["n"]), dtype=np.int64)
        capi.check(self._L.exa_locality_order(self.id, k, out.ctypes.data), "exa_locality_order")
        return out[:self.pattern_info(k)["n"]]

    def eval_all_mode(self):
        """exa_eval_all_mode: how exa_eval_all produces grad! (1 / 2: inside the sweep's one launch; 0 / 3: a launch in front; 4: sorted)."""
        return self._L.exa_eval_all_mode(self.id)

    def kernel_source(self):
        return self._L.exa_kernel_source(self.id).decode()

    def module_source(self, k=0):
        """Generated HIP source of module k: 0 the model's, 1 the owner-computes product windows' ("" when it has none)."""
        return self._L.exa_module_source(self.id, k).decode()

    def compile(self):
        capi.check(self._L.exa_compile(self.id), "exa_compile")
        return self._L.exa_code_object_path(self.id).decode()

    def code_objects(self):
        """[(module name, code object bytes)] of a compiled model: its module and, where it has them, the product windows'."""
        out = []
        for k in range(self._L.exa_code_object_count(self.id)):
            name, path = ctypes.create_string_buffer(128), ctypes.create_string_buffer(4096)
            capi.check(self._L.exa_code_object(self.id, k, name, 128, path, 4096), "exa_code_object")
            with open(path.value.decode(), "rb") as fh:
                out.append((name.value.decode(), fh.read()))
        return out

    def build_info(self):
        """(how, build_ms): how the module was obtained — "preloaded" | "disk" | "hiprtc" | "hipcc" — and the compiler's time."""
        buf, ms = ctypes.create_string_buffer(32), ctypes.c_double(0.0)
        capi.check(self._L.exa_build_info(self.id, buf, 32, ctypes.addressof(ms)), "exa_build_info")
        return buf.value.decode(), ms.value

    def build_audit(self):
        """exa_build_audit as a list of dicts, one per compiled kernel: module, object, flags ("default" | "safe"), kernel, vgpr,
        agpr, scratch, vgpr_spill, sgpr_spill, lds, fits (None for a module whose metadata could not be read)."""
        n = self._L.exa_build_audit(self.id, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        self._L.exa_build_audit(self.id, buf, n + 1)
        out = []
        for line in buf.value.decode().splitlines():
            f = line.split()
            if f[3] == "?":
                out.append(dict(module=f[0], object=f[1], flags=f[2], kernel=None, fits=None))
                continue
            out.append(dict(module=f[0], object=f[1], flags=f[2], kernel=f[3], vgpr=int(f[4]), agpr=int(f[5]), scratch=int(f[6]), vgpr_spill=int(f[7]),
                            sgpr_spill=int(f[8]), lds=int(f[9]), fits=f[10] == "fits"))
        return out

    def tune(self, what=7, x=None, y=None):
        """exa_tune: the explicit, blocking measurement of block orders (bit 0), product implementations (bit 1) and the
        grad! implementation (bit 2); the decisions are persisted next to the cached module.  x, y: device tensors or None."""
        if x is not None:
            self._use_torch_stream(x)
        capi.check(self._L.exa_tune(self.id, int(what), x.data_ptr() if x is not None else None,
                                    y.data_ptr() if y is not None else None), "exa_tune")

    # ---- context ---------------------------------------------------------------------------------------------
    def set_shard(self, rank, world):
        capi.check(self._L.exa_set_shard(self.id, int(rank), int(world)), "exa_set_shard")

    # ---- multi-GPU behind the ABI (include/exahip.h "multi-GPU") ---------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(128)
        capi.check(capi.lib().exa_comm_unique_id(buf), "exa_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        """RCCL communicator on the current HIP device + shard; obj/grad/cons/products are complete on every rank afterwards."""
        capi.check(self._L.exa_comm_init(self.id, int(rank), int(world), ctypes.c_char_p(bytes(unique_id))), "exa_comm_init")

    def comm_hook(self, rank, world, fn):
        """fn(device_ptr: int, count: int, stream: int) -> 0 must leave the sum over ranks in the device buffer."""
        proto = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)

        def tramp(_ctx, buf, count, stream):
            try:
                return int(fn(buf or 0, int(count), stream or 0) or 0)
            except Exception:      # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._hook = proto(tramp)                       # keep the trampoline alive as long as the model
        capi.check(self._L.exa_comm_hook(self.id, int(rank), int(world), ctypes.cast(self._hook, ctypes.c_void_p), None), "exa_comm_hook")

    def comm_free(self):
        capi.check(self._L.exa_comm_free(self.id), "exa_comm_free")
        self._hook = None

    def comm_info(self):
        r, w, k = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        capi.check(self._L.exa_comm_info(self.id, ctypes.addressof(r), ctypes.addressof(w), ctypes.addressof(k)), "exa_comm_info")
        return r.value, w.value, ("none", "rccl", "hook")[k.value]

    def set_reduce(self, on):
        capi.check(self._L.exa_set_reduce(self.id, 1 if on else 0), "exa_set_reduce")

    def allreduce(self, t):
        self._use_torch_stream(t)
        capi.check(self._L.exa_allreduce(self.id, t.data_ptr(), t.numel()), "exa_allreduce")
        return t

    def comm_complete(self, which, buf):
        """exa_comm_complete: the collective operations of exa_collective_plan(which) on buf, through the attached RCCL communicator (deferred
        completion after set_reduce(False); with a world-1 communicator a real in-place ncclAllGather / ncclAllReduce of buf onto itself)."""
        self._use_torch_stream(buf)
        capi.check(self._L.exa_comm_complete(self.id, int(which), buf.data_ptr()), "exa_comm_complete")
        return buf

    def allgather_coo(self, local, hess=True, out=None):
        """exa_allgather_coo: the sharded Jacobian / Hessian COO vector whole on every rank (all-gather-v of the slot ranges).
        `local`: this rank's exa_jac / exa_hess output (packed local slice, or global-length with its slots in place)."""
        import torch
        self._use_torch_stream(local)
        n = self.meta.nnzh if hess else self.meta.nnzj
        if out is None:
            out = local if local.numel() == n and not self._coo_local else torch.empty(n, dtype=torch.float64, device=local.device)
        capi.check(self._L.exa_allgather_coo(self.id, 1 if hess else 0, local.data_ptr(), out.data_ptr()), "exa_allgather_coo")
        return out

    def shard_layout(self, which):
        """"pieces" (complete values in disjoint pieces: all-gather completes) or "partial" (partial sums: all-reduce completes)
        — how a rank of a sharded model leaves the output of `which` (obj grad cons jac hess jprod jtprod hprod) on its own."""
        k = {"obj": 0, "grad": 1, "cons": 2, "jac": 3, "hess": 4, "jprod": 5, "jtprod": 6, "hprod": 7, "fused_cons": 8}[which]
        r = self._L.exa_shard_layout(self.id, k)
        if r < 0:
            raise capi.ExaHipError("exa_shard_layout")
        return "pieces" if r == 1 else "partial"

    def set_coo_local(self, on=True):
        self._coo_local = bool(on)
        capi.check(self._L.exa_set_coo_local(self.id, 1 if on else 0), "exa_set_coo_local")

    @property
    def local_nnzj(self):
        return self._L.exa_local_nnzj64(self.id)

    @property
    def local_nnzh(self):
        return self._L.exa_local_nnzh64(self.id)

    def coo_slices(self, hess=True):
        """[(first global slot, first position in the caller's buffer, length)] per pattern (exa_coo_slices)."""
        out = np.zeros(3 * max(1, self.npatterns), dtype=np.int64)
        capi.check(self._L.exa_coo_slices(self.id, 1 if hess else 0, out.ctypes.data), "exa_coo_slices")
        return [tuple(out[3 * k:3 * k + 3].tolist()) for k in range(self.npatterns)]

    def shard_var_range(self):
        lo, hi = ctypes.c_int64(0), ctypes.c_int64(0)
        capi.check(self._L.exa_shard_var_range(self.id, ctypes.addressof(lo), ctypes.addressof(hi)), "exa_shard_var_range")
        return lo.value, hi.value

    def set_value(self, par, values):
        """set_value!(m, θ, vals): update a Parameter block without rebuilding (nlp.jl:1279-1287).  A vector of the
        wrong length is a DimensionMismatch in the reference (GetterSetterTest.jl:33-34); a scalar fills the block.  A device
        tensor is copied device-to-device on the model's stream (exa_set_value_dev): no host hop, no synchronisation."""
        if _is_torch(values) and values.is_cuda:
            if values.numel() != par.length:
                raise ValueError(f"dimension mismatch: parameter block has {par.length} entries, got {values.numel()}")
            self._use_torch_stream(values)
            t = self._tcheck(values, par.length, "values")
            capi.check(self._L.exa_set_value_dev(self.id, par.offset, t.data_ptr(), par.length), "exa_set_value_dev")
            return
        a = np.asarray(values, dtype=np.float64)
        if a.ndim > 0 and a.size != par.length:
            raise ValueError(f"dimension mismatch: parameter block has {par.length} entries, got {a.size}")
        v = np.ascontiguousarray(np.broadcast_to(a.reshape(-1) if a.ndim else a, (par.length,)))
        capi.check(self._L.exa_set_value(self.id, par.offset, v.ctypes.data, v.size), "exa_set_value")

    def theta_view(self):
        """get_value's device view (nlp.jl:1270-1277): the library's parameter vector as a torch tensor sharing its memory (npar
        doubles on the model's device); writes through it are seen by callbacks launched afterwards on the same stream."""
        import torch
        p = self._L.exa_theta_ptr(self.id)
        if not p:
            raise capi.ExaHipError("exa_theta_ptr: no device-resident parameters")
        n = int(self.ir.desc.npar)

        class _Mem:          # __cuda_array_interface__ carrier: torch.as_tensor wraps the memory without copying
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(p), False), "version": 2}
        return torch.as_tensor(_Mem(), device="cuda")

    def get_value(self, par):
        """get_value(m, θ): the model's current values of a Parameter block (a copy: the storage is the library's)."""
        v = np.empty(par.length)
        capi.check(self._L.exa_get_value(self.id, par.offset, v.ctypes.data, v.size), "exa_get_value")
        return v

    def _use_torch_stream(self, t):
        import torch
        s = torch.cuda.current_stream(t.device).cuda_stream
        if s != self._stream:
            capi.check(self._L.exa_set_stream(self.id, ctypes.c_void_p(s)), "exa_set_stream")
            self._stream = s

    def sync(self):
        capi.check(self._L.exa_sync(self.id), "exa_sync")

    @staticmethod
    def _np(a, n, name):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.size != n:
            raise ValueError(f"{name} has {a.size} entries, expected {n}")
        return a

    def _tcheck(self, t, n, name):
        import torch
        if t.dtype != torch.float64 or not t.is_contiguous() or t.numel() < n or not t.is_cuda:
            raise ValueError(f"{name} must be a contiguous float64 CUDA tensor with >= {n} entries")
        return t

    # ---- callbacks ---------------------------------------------------------------------------------------------
    def obj(self, x):
        out = ctypes.c_double(0.0)
        if _is_torch(x):
            self._use_torch_stream(x)
            self._tcheck(x, self.meta.nvar, "x")
            capi.check(self._L.exa_obj(self.id, x.data_ptr(), ctypes.addressof(out)), "exa_obj")
        else:
            x = self._np(x, self.meta.nvar, "x")
            capi.check(self._L.exa_obj_host(self.id, x.ctypes.data, ctypes.addressof(out)), "exa_obj_host")
        return out.value

    def _call(self, name, x, n_out, out, extra=None):
        if _is_torch(x):
            import torch
            self._use_torch_stream(x)
            self._tcheck(x, self.meta.nvar, "x")
            if out is None:
                out = torch.empty(n_out, dtype=torch.float64, device=x.device)
            self._tcheck(out, n_out, "out")
            if extra is None:
                capi.check(getattr(self._L, "exa_" + name)(self.id, x.data_ptr(), out.data_ptr()), name)
            else:
                y, w = extra
                if y is not None:
                    self._tcheck(y, self.meta.ncon, "y")
                capi.check(self._L.exa_hess(self.id, x.data_ptr(), y.data_ptr() if y is not None else None, float(w), out.data_ptr()), name)
            return out
        x = self._np(x, self.meta.nvar, "x")
        if out is None:
            out = np.empty(n_out)
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.size >= n_out
        if extra is None:
            capi.check(getattr(self._L, f"exa_{name}_host")(self.id, x.ctypes.data, out.ctypes.data), name)
        else:
            y, w = extra
            y = self._np(y, self.meta.ncon, "y") if y is not None else np.empty(0)
            capi.check(self._L.exa_hess_host(self.id, x.ctypes.data, y.ctypes.data if y.size else None, float(w), out.ctypes.data), name)
        return out

    def grad(self, x, out=None):
        return self._call("grad", x, self.meta.nvar, out)

    def cons(self, x, out=None):
        return self._call("cons", x, self.meta.ncon, out)

    cons_nln = cons

    def jac_coord(self, x, out=None):
        return self._call("jac", x, self.local_nnzj, out)

    def hess_coord(self, x, y=None, obj_weight=1.0, out=None):
        """hess_coord!(m, x, y, hess; obj_weight); y=None is the objective-only form hess_coord!(m, x, hess; obj_weight)
        (nlp.jl:1906-1915): the constraint slots come back as zeros."""
        return self._call("hess", x, self.local_nnzh, out, extra=(y, obj_weight))

    def eval_fused(self, x, y, obj_weight=1.0, c=None, jac=None, hess=None, obj_out=None):
        """obj + cons + jac_coord + hess_coord at one x in ONE sweep (exa_eval_fused).  Device tensors only.
        Returns (obj as a 1-element device tensor, c, jac, hess).  Nothing here synchronises, so the call can be
        captured into a hipGraph (torch.cuda.graph) once the model has been evaluated once outside the capture."""
        import torch
        self._use_torch_stream(x)
        dev = x.device
        f = torch.empty(1, dtype=torch.float64, device=dev) if obj_out is None else obj_out
        c = torch.empty(self.meta.ncon, dtype=torch.float64, device=dev) if c is None else c
        jac = torch.empty(self.local_nnzj, dtype=torch.float64, device=dev) if jac is None else jac
        hess = torch.empty(self.local_nnzh, dtype=torch.float64, device=dev) if hess is None else hess
        capi.check(self._L.exa_eval_fused(self.id, x.data_ptr(), y.data_ptr(), float(obj_weight), f.data_ptr(), c.data_ptr(),
                                          jac.data_ptr(), hess.data_ptr()), "exa_eval_fused")
        return f, c, jac, hess

    def eval_all(self, x, y, obj_weight=1.0, g=None, c=None, jac=None, hess=None, obj_out=None):
        """obj + grad + cons + jac_coord + hess_coord at one x (exa_eval_all): the evaluation set of a solver iteration.
        Device tensors only; returns (obj as a 1-element device tensor, g, c, jac, hess); nothing synchronises."""
        import torch
        self._use_torch_stream(x)
        dev = x.device
        f = torch.empty(1, dtype=torch.float64, device=dev) if obj_out is None else obj_out
        g = torch.empty(self.meta.nvar, dtype=torch.float64, device=dev) if g is None else g
        c = torch.empty(self.meta.ncon, dtype=torch.float64, device=dev) if c is None else c
        jac = torch.empty(self.local_nnzj, dtype=torch.float64, device=dev) if jac is None else jac
        hess = torch.empty(self.local_nnzh, dtype=torch.float64, device=dev) if hess is None else hess
        capi.check(self._L.exa_eval_all(self.id, x.data_ptr(), y.data_ptr(), float(obj_weight), f.data_ptr(), g.data_ptr(), c.data_ptr(),
                                        jac.data_ptr(), hess.data_ptr()), "exa_eval_all")
        return f, g, c, jac, hess

    # ---- matrix-free products: jprod_nln! / jtprod_nln! / hprod! (nlp.jl:1882-1978) -----------------------------
    def _prod(self, name, x, v, nv, n_out, out, y=None, w=1.0):
        if _is_torch(x):
            import torch
            self._use_torch_stream(x)
            self._tcheck(x, self.meta.nvar, "x")
            self._tcheck(v, nv, "v")
            if out is None:
                out = torch.empty(n_out, dtype=torch.float64, device=x.device)
            if name == "hprod":
                capi.check(self._L.exa_hprod(self.id, x.data_ptr(), y.data_ptr() if y is not None else None, v.data_ptr(), float(w), out.data_ptr()), name)
            else:
                capi.check(getattr(self._L, "exa_" + name)(self.id, x.data_ptr(), v.data_ptr(), out.data_ptr()), name)
            return out
        x = self._np(x, self.meta.nvar, "x")
        v = self._np(v, nv, "v")
        if out is None:
            out = np.empty(n_out)
        if name == "hprod":
            y = self._np(y, self.meta.ncon, "y") if y is not None else np.empty(0)
            capi.check(self._L.exa_hprod_host(self.id, x.ctypes.data, y.ctypes.data if y.size else None, v.ctypes.data, float(w), out.ctypes.data), name)
        else:
            capi.check(getattr(self._L, f"exa_{name}_host")(self.id, x.ctypes.data, v.ctypes.data if v.size else None, out.ctypes.data), name)
        return out

    def set_product_mode(self, jtprod=-1, hprod=-1):
        """0 atomics in the sweep, 1 COO + sorted gather, 2 owner-computes windows (range-affine models), 3 owner pull (data-indexed
        models), -1 undecided (default): what tune() persisted, else the windows where the model has them, else atomics."""
        capi.check(self._L.exa_set_product_mode(self.id, int(jtprod), int(hprod)), "exa_set_product_mode")

    def set_grad_mode(self, mode=-1):
        """grad!: 0 gathered + FP64 atomics, 1 gradient COO + sorted gather (deterministic), -1 whatever tune() persisted"""
        capi.check(self._L.exa_set_grad_mode(self.id, int(mode)), "exa_set_grad_mode")

    def set_deterministic(self, on=True):
        """grad!, jtprod and hprod by sorted gather (bit-reproducible) / back to undecided"""
        capi.check(self._L.exa_set_deterministic(self.id, 1 if on else 0), "exa_set_deterministic")

    def grad_mode(self):
        a = ctypes.c_int(0)
        capi.check(self._L.exa_get_grad_mode(self.id, ctypes.addressof(a)), "exa_get_grad_mode")
        return a.value

    def product_info(self, which):
        """(mode, text) for which = "jtprod" | "hprod": the implementation a call would run now (0 atomics, 1 sorted gather,
        2 owner-computes windows) and the kernel shape of the windows / why the model has none (exa_product_info)."""
        buf = ctypes.create_string_buffer(512)
        mode = self._L.exa_product_info(self.id, 1 if which == "hprod" else 0, buf, 512)
        return mode, buf.value.decode()

    def product_mode(self):
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        capi.check(self._L.exa_get_product_mode(self.id, ctypes.addressof(a), ctypes.addressof(b)), "exa_get_product_mode")
        return a.value, b.value

    def jprod(self, x, v, out=None):
        return self._prod("jprod", x, v, self.meta.nvar, self.meta.ncon, out)

    def jtprod(self, x, v, out=None):
        return self._prod("jtprod", x, v, self.meta.ncon, self.meta.nvar, out)

    def hprod(self, x, y, v, obj_weight=1.0, out=None):
        """hprod!(m, x, y, v, Hv; obj_weight); y=None is the objective-only form hprod!(m, x, v, Hv; obj_weight) (nlp.jl:1942-1952)"""
        return self._prod("hprod", x, v, self.meta.nvar, self.meta.nvar, out, y=y, w=obj_weight)

    def _structure(self, which, rows, cols, nnz, dtype):
        wide = np.dtype(dtype).itemsize == 8
        if rows is not None and _is_torch(rows):
            self._use_torch_stream(rows)
            fn = getattr(self._L, f"exa_{which}_structure" + ("64" if wide else ""))
            capi.check(fn(self.id, rows.data_ptr(), cols.data_ptr()), which + "_structure")
            return rows, cols
        if rows is None:
            rows, cols = np.empty(nnz, dtype=dtype), np.empty(nnz, dtype=dtype)
        fn = getattr(self._L, f"exa_{which}_structure" + ("64" if wide else "") + "_host")
        capi.check(fn(self.id, rows.ctypes.data, cols.ctypes.data), which + "_structure")
        return rows, cols

    def jac_structure(self, rows=None, cols=None, dtype=np.int64):
        if rows is not None:
            dtype = np.int64 if (rows.element_size() if _is_torch(rows) else rows.dtype.itemsize) == 8 else np.int32
        return self._structure("jac", rows, cols, self.local_nnzj, dtype)

    def hess_structure(self, rows=None, cols=None, dtype=np.int64):
        if rows is not None:
            dtype = np.int64 if (rows.element_size() if _is_torch(rows) else rows.dtype.itemsize) == 8 else np.int32
        return self._structure("hess", rows, cols, self.local_nnzh, dtype)

    # ---- measurement (hipEvents on the model's stream, include/exahip.h exa_time_callback) -------------------
    def time_callback(self, which, reps, x, y=None, obj_weight=1.0, out=None):
        ms = ctypes.c_float(0.0)
        self._use_torch_stream(x)
        capi.check(self._L.exa_time_callback(self.id, _WHICH[which], int(reps), x.data_ptr(),
                                             y.data_ptr() if y is not None else None, float(obj_weight),
                                             out.data_ptr() if out is not None else None, ctypes.addressof(ms)),
                   "exa_time_callback")
        return ms.value


class CompressedExaModel:
    """CompressedNLPModel(m) (src/utils.jl:425-579): same model, Jacobian/Hessian COO with duplicate (row, col) entries
    summed.  Device tensors only (the point is to hand a solver a smaller KKT assembly on the GPU)."""

    def __init__(self, m: ExaModel):
        self.inner = m
        self._L = m._L
        capi.check(self._L.exa_compress(m.id), "exa_compress")
        self.meta = m.meta.copy()
        self.meta.nnzj = self._L.exa_cnnzj64(m.id)
        self.meta.nnzh = self._L.exa_cnnzh64(m.id)

    def path(self, which):
        """("windowed" | "scatter" | "gather", reason) for which = "jac" | "hess" (exa_compress_info): the windowed sweep
        (stencil models), the permuted store (the sweep writes every slot at its sorted position; duplicates summed
        sequentially), or the reference's scheme (uncompressed evaluation + sorted gather)."""
        import ctypes
        buf = ctypes.create_string_buffer(512)
        r = self._L.exa_compress_info(self.inner.id, 1 if which == "hess" else 0, buf, 512, None)
        if r < 0:
            raise RuntimeError("exa_compress_info")
        return {1: "windowed", 2: "scatter"}.get(r, "gather"), buf.value.decode()

    def obj(self, x):
        return self.inner.obj(x)

    def grad(self, x, out=None):
        return self.inner.grad(x, out=out)

    def cons(self, x, out=None):
        return self.inner.cons(x, out=out)

    def _structure(self, which, nnz, device):
        import torch
        rows = torch.empty(nnz, dtype=torch.int64, device=device)
        cols = torch.empty(nnz, dtype=torch.int64, device=device)
        capi.check(getattr(self._L, f"exa_c{which}_structure64")(self.inner.id, rows.data_ptr(), cols.data_ptr()), which)
        return rows, cols

    def csc(self, which, device="cuda:0"):
        """(colptr [nvar+1], rowval [nnz]) of the compressed Jacobian ("jac") or lower-triangular Hessian ("hess"),
        1-based: with the values of jac_coord / hess_coord as nzval this is the SparseMatrixCSC of the matrix."""
        import torch
        nnz = self.meta.nnzj if which == "jac" else self.meta.nnzh
        colptr = torch.empty(self.meta.nvar + 1, dtype=torch.int64, device=device)
        rowval = torch.empty(nnz, dtype=torch.int64, device=device)
        capi.check(getattr(self._L, f"exa_c{which}_csc")(self.inner.id, colptr.data_ptr(), rowval.data_ptr()), which)
        return colptr, rowval

    def jac_structure(self, device="cuda:0"):
        return self._structure("jac", self.meta.nnzj, device)

    def hess_structure(self, device="cuda:0"):
        return self._structure("hess", self.meta.nnzh, device)

    def jac_coord(self, x, out=None):
        import torch
        self.inner._use_torch_stream(x)
        if out is None:
            out = torch.empty(self.meta.nnzj, dtype=torch.float64, device=x.device)
        capi.check(self._L.exa_cjac(self.inner.id, x.data_ptr(), out.data_ptr()), "exa_cjac")
        return out

    def hess_coord(self, x, y, obj_weight=1.0, out=None):
        import torch
        self.inner._use_torch_stream(x)
        if out is None:
            out = torch.empty(self.meta.nnzh, dtype=torch.float64, device=x.device)
        capi.check(self._L.exa_chess(self.inner.id, x.data_ptr(), y.data_ptr(), float(obj_weight), out.data_ptr()), "exa_chess")
        return out


class TimedExaModel:
    """TimedNLPModel(m) (src/utils.jl:271-408): counts the calls of every callback and the seconds spent in them.
    Unlike the reference's wrapper (which does not synchronise the device) the model's stream is drained
    before the clock is read, so the seconds are kernel-inclusive."""

    _NAMES = ("obj", "cons", "grad", "jac_coord", "hess_coord", "jprod", "jtprod", "hprod", "jac_structure", "hess_structure")

    def __init__(self, m):
        self.inner = m
        self.meta = m.meta
        self.stats = {n: {"calls": 0, "seconds": 0.0} for n in self._NAMES}

    def __getattr__(self, name):
        if name in TimedExaModel._NAMES:
            import time
            fn = getattr(self.inner, name)

            def timed(*a, **kw):
                t0 = time.perf_counter()
                out = fn(*a, **kw)
                if hasattr(self.inner, "sync"):
                    self.inner.sync()
                s = self.stats[name]
                s["calls"] += 1
                s["seconds"] += time.perf_counter() - t0
                return out

            return timed
        return getattr(self.inner, name)

    def report(self):
        lines = [f"{'callback':16s} {'calls':>8s} {'seconds':>12s}"]
        for n, s in self.stats.items():
            if s["calls"]:
                lines.append(f"{n:16s} {s['calls']:8d} {s['seconds']:12.6f}")
        return "\n".join(lines)
