"""exahip.pack: a shared library with the reference's literal cnlp ABI v0.1 symbol names.

Played from the consumer's side, the way ExaModelsCompiler/test/cnlpmodels_check.py and builder_check.py drive a
compiled library: nothing but the library path, the prefix and ctypes.  CPU half: catalogue, arity/argtype/schema,
disjoint surfaces, loud failure without a device.  `gpu` half: the two consumer scripts' protocols against the
oracle on the directly built model."""
import ctypes
import json

import numpy as np
import pytest

from conftest import has_gpu
from recipezoo import S_ARGS, S_EX, S_N, build_knob, build_layout, build_struct, make

c_int, c_dbl, vp = ctypes.c_int, ctypes.c_double, ctypes.c_void_p


@pytest.fixture(scope="module")
def packed(libs, tmp_path_factory):
    from exahip import models
    from exahip.pack import pack_library
    out = tmp_path_factory.mktemp("pack") / "structs"
    path = pack_library(str(out),
                        ("structm", make(build_struct, S_EX, True)),
                        ("knob", make(build_knob, (4,), True)),
                        ("lay", make(build_layout, (3,), True)),
                        ("lv7", models.luksan_vlcek_model(7)))              # a fixed model: no placeholders
    try:
        import torch  # noqa: F401  (one HIP runtime per process, see exahip/capi.py)
    except ImportError:
        pass
    return ctypes.CDLL(path)


def _text(fn, *a):
    n = fn(*a, None, 0)
    buf = ctypes.create_string_buffer(max(1, n))
    fn(*a, buf, n)
    return buf.raw[:n].decode()


def test_catalogue_and_surfaces(packed):
    lib = packed
    assert lib.cnlp_nmodels() == 4                                           # runtests.jl:588-591
    names = [_text(lib.cnlp_model_name, k) for k in range(4)]
    assert names == ["structm", "knob", "lay", "lv7"] and lib.cnlp_model_name(4, None, 0) == -1
    # the builder model exports no one-integer constructor; the knob model exports no builder (runtests.jl:593-601)
    assert not hasattr(lib, "structm_new") and hasattr(lib, "structm_data_begin")
    assert hasattr(lib, "knob_new") and not hasattr(lib, "knob_data_begin")
    assert lib.structm_nargs() == 4 and lib.knob_nargs() == 1 and lib.lv7_nargs() == 0
    assert _text(lib.knob_argtype) == "int|size" and _text(lib.lay_argtype) == "int|size" and _text(lib.lv7_argtype) == ""
    assert _text(lib.structm_argtype) == "int|arg1,Vector{f64}|v0,Vector{f64}|lo,Table{i::int w::f64 s::f64}|arg3"
    fields = json.loads(_text(lib.structm_schema))["fields"]
    assert [f["name"] for f in fields] == ["arg1", "v0", "lo", "arg3"]
    assert fields[3] == {"name": "arg3", "kind": "table",
                         "columns": [{"name": "i", "type": "i64"}, {"name": "w", "type": "f64"}, {"name": "s", "type": "f64"}]}
    for p in names:
        for f in ("nvar", "ncon", "nnzj", "nnzh", "meta", "obj", "grad", "cons", "jac_structure", "jac", "hess_structure",
                  "hess", "nblocks", "block_name", "block", "get_value", "set_value", "nargs", "argtype"):
            assert hasattr(lib, f"{p}_{f}"), f"{p}_{f}"
    assert lib.knob_nvar(12345) == -1 and lib.knob_obj(12345, None, None) == 1


@pytest.mark.skipif(has_gpu(), reason="CPU-only behaviour")
def test_no_device_no_model(packed):
    """There is no CPU fallback behind the cnlp names either: without an MI355X `P_new` returns the documented
    failure value 0."""
    assert packed.knob_new(5) == 0 and packed.lv7_new(0) == 0


def test_bundle_is_relocatable(libs, tmp_path):
    """bundle=True: the directory carries libexahip.so and finds it through $ORIGIN after being moved."""
    import shutil
    import subprocess
    from exahip.pack import pack_library
    pack_library(str(tmp_path / "a" / "rosen"), make(build_knob, (4,), True), bundle=True)
    shutil.move(str(tmp_path / "a"), str(tmp_path / "moved"))
    out = subprocess.check_output(["ldd", str(tmp_path / "moved" / "librosen.so")]).decode()
    line = [ln for ln in out.splitlines() if "libexahip.so" in ln][0]
    assert str(tmp_path / "moved" / "libexahip.so") in line


def test_prefix_must_be_a_c_identifier(libs, tmp_path):
    from exahip.pack import pack_library
    for bad in ("lib-a", "2fast"):                                           # runtests.jl:88-89
        with pytest.raises(ValueError):
            pack_library(str(tmp_path / "x"), (bad, make(build_knob, (4,), True)))


# ---- GPU: the consumer protocols ---------------------------------------------------------------------------------
class Consumer:
    """What cnlpmodels.CModel does with a prefix and an instance id (cnlpmodels_check.py:20-47)."""

    def __init__(self, lib, prefix, mid):
        assert mid > 0
        self.f = lambda name: getattr(lib, f"{prefix}_{name}")
        self.id = mid
        self.nvar, self.ncon = self.f("nvar")(mid), self.f("ncon")(mid)
        self.nnzj, self.nnzh = self.f("nnzj")(mid), self.f("nnzh")(mid)
        self.x0, self.lvar, self.uvar = np.empty(self.nvar), np.empty(self.nvar), np.empty(self.nvar)
        self.lcon, self.ucon = np.empty(self.ncon), np.empty(self.ncon)
        assert self.f("meta")(mid, *[vp(a.ctypes.data) for a in (self.x0, self.lvar, self.uvar, self.lcon, self.ucon)]) == 0

    def call(self, name, n, *ins, w=None):
        out = np.full(n, np.nan)
        args = [vp(np.ascontiguousarray(a).ctypes.data) for a in ins]
        if w is not None:
            args.append(c_dbl(w))
        assert self.f(name)(self.id, *args, vp(out.ctypes.data)) == 0, name
        return out

    def structure(self, name, n):
        r, c = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        assert self.f(name)(self.id, vp(r.ctypes.data), vp(c.ctypes.data)) == 0
        return r, c


def _against_oracle(m, core):
    import oracle
    ref = oracle.OracleModel(core.to_ir())
    assert (m.nvar, m.ncon, m.nnzj, m.nnzh) == (ref.nvar, ref.ncon, ref.nnzj, ref.nnzh)
    x = np.linspace(0.5, 3.0, m.nvar)
    y = np.linspace(-1.0, 1.0, m.ncon)
    for a, b in zip((m.x0, m.lvar, m.uvar, m.lcon, m.ucon), ref.meta()):
        assert np.array_equal(a, b)
    np.testing.assert_allclose(m.call("obj", 1, x)[0], ref.obj(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.call("grad", m.nvar, x), ref.grad(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.call("cons", m.ncon, x), ref.cons(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.call("jac", m.nnzj, x), ref.jac_coord(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.call("hess", m.nnzh, x, y, w=0.5), ref.hess_coord(x, y, 0.5), rtol=1e-10, atol=1e-12)
    for (a, b), (ra, rb) in ((m.structure("jac_structure", m.nnzj), ref.jac_structure()),
                             (m.structure("hess_structure", m.nnzh), ref.hess_structure())):
        assert np.array_equal(a, ra) and np.array_equal(b, rb)            # 1-based, as the ABI and ExaModels use


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_one_integer_and_fixed_models_through_cnlp_names(packed):
    from exahip import models
    lib = packed
    m6, m11 = Consumer(lib, "knob", lib.knob_new(6)), Consumer(lib, "knob", lib.knob_new(11))
    assert (m6.nvar, m11.nvar) == (6, 11)
    assert m6.call("obj", 1, np.ones(6))[0] == 6.0 and m11.call("obj", 1, np.ones(11))[0] == 11.0     # runtests.jl:299-307
    _against_oracle(m6, make(build_knob, (6,), False))
    _against_oracle(m11, make(build_knob, (11,), False))
    _against_oracle(Consumer(lib, "lv7", lib.lv7_new(0)), models.luksan_vlcek_model(7))
    assert lib.knob_free(m6.id) == 0 and lib.knob_nvar(m6.id) == -1 and m11.call("obj", 1, np.ones(11))[0] == 11.0


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_structured_model_through_the_builder_names(packed):
    """builder_check.py: positional values against the schema's field order, the table as a dict of columns."""
    lib = packed
    n, dat, tab = S_ARGS
    values = [n, dat["v0"], dat["lo"], {k: v for k, v in tab.cols.items()}]
    fields = json.loads(_text(lib.structm_schema))["fields"]
    b = lib.structm_data_begin()
    assert b > 0
    for f, v in zip(fields, values):
        name = f["name"].encode()
        if f["kind"] == "scalar":
            assert lib.structm_set_scalar_i64(b, name, ctypes.c_longlong(v)) == 0
        elif f["kind"] == "array":
            a = np.ascontiguousarray(v, dtype=np.float64)
            assert lib.structm_set_array_f64(b, name, vp(a.ctypes.data), len(a)) == 0
        else:
            for c in f["columns"]:
                a = np.ascontiguousarray(v[c["name"]], dtype=np.int64 if c["type"] == "i64" else np.float64)
                fn = lib.structm_set_col_i64 if c["type"] == "i64" else lib.structm_set_col_f64
                assert fn(b, name, c["name"].encode(), vp(a.ctypes.data), len(a)) == 0
    assert lib.structm_data_ready(b) == 1
    m = Consumer(lib, "structm", lib.structm_new_from_data(b))
    assert (m.nvar, m.ncon) == (S_N, S_N - 1)
    _against_oracle(m, make(build_struct, S_ARGS, False))


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_layout_and_live_parameters_through_cnlp_names(packed):
    """runtests.jl:676-745: named blocks at THIS instance's size; set_value re-evaluates without a rebuild."""
    lib = packed
    lid = lib.lay_new(3)
    assert lid > 0 and lib.lay_nblocks(lid) == 4
    names, recs = [], []
    for k in range(4):
        names.append(_text(lib.lay_block_name, lid, k))
        out = (c_int * 16)()
        assert lib.lay_block(lid, k, out) == 0
        recs.append(list(out))
    assert names == ["y", "w", "link", "dax"] and [r[0] for r in recs] == [0, 2, 1, 1]
    assert recs[0][1:6] == [0, 6, 2, 3, 2]
    m = Consumer(lib, "lay", lid)
    x = np.linspace(0.5, 3.0, m.nvar)
    f1 = m.call("obj", 1, x)[0]
    w = (c_dbl * 3)(2.0, 1.0, 1.0)
    assert lib.lay_set_value(lid, 1, w, 3) == 0 and lib.lay_set_value(lid, 1, w, 2) == 3 and lib.lay_set_value(lid, 0, w, 6) == 1
    got = (c_dbl * 3)()
    assert lib.lay_get_value(lid, 1, got, 3) == 0 and list(got) == [2.0, 1.0, 1.0]
    np.testing.assert_allclose(m.call("obj", 1, x)[0], 2.0 * f1, rtol=1e-13)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_packed_library_is_ahead_of_time(libs, tmp_path, monkeypatch):
    """The packed library carries its gfx950 code object: with an EMPTY kernel cache and NO usable hipcc the consumer
    still instantiates and evaluates (compile_library's ahead-of-time promise, ExaModelsCompiler.jl:108-222)."""
    from exahip.pack import pack_library
    path = pack_library(str(tmp_path / "aot"), ("knob2", make(build_knob, (4,), True)))
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path / "empty_cache"))
    monkeypatch.setenv("EXAHIP_COMPILER", "hipcc")
    monkeypatch.setenv("EXAHIP_HIPCC", "/nonexistent/hipcc")
    lib = ctypes.CDLL(path)
    m = Consumer(lib, "knob2", lib.knob2_new(9))
    assert m.nvar == 9 and m.call("obj", 1, np.ones(9))[0] == 9.0
    _against_oracle(m, make(build_knob, (9,), False))
    assert not list((tmp_path / "empty_cache").glob("*.hsaco"))          # nothing was compiled or cached at run time
