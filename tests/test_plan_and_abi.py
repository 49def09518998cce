"""-m "not gpu": host logic of libexahip.so without a device.

* the library loads and exports every symbol include/exahip.h declares;
* status codes: 0 ok / 1 bad id or argument / 2 internal error, never an exception across the boundary
  (cnlp ABI convention, ExaModelsCompiler/src/ExaModelsCompiler.jl:1560-1732);
* the PLANNER (slot maps comp1/comp2, o1step/o2step, running offsets) agrees with the test oracle's independent
  implementation on every zoo model and every golden expression;
* the generated HIP module of representative models compiles for gfx950 (hipcc cross-compiles without a GPU);
* no compute entry point works without a device: they return status 1 on a plan-only model, and
  exa_new_from_table fails with status 2 ("no HIP device") instead of falling back to the CPU.
"""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

from conftest import has_gpu  # noqa: E402
from zoo import ZOO  # noqa: E402


def header_symbols(name="exahip.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(exa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(libs):
    from exahip import capi
    L = capi.lib()
    declared = header_symbols()
    assert len(declared) >= 40
    for s in declared:
        assert hasattr(L, s), f"libexahip.so does not export {s}"
    assert sorted(capi.SYMBOLS) == declared, "capi.SYMBOLS and include/exahip.h disagree"
    recipe = header_symbols("exahip_recipe.h")
    assert len(recipe) >= 25
    for s in recipe:
        assert hasattr(L, s), f"libexahip.so does not export {s}"
    assert sorted(capi.RECIPE_SYMBOLS) == recipe, "capi.RECIPE_SYMBOLS and include/exahip_recipe.h disagree"
    assert L.exa_abi_version() == 3


def test_bad_ids_and_arguments_return_status_1(libs):
    from exahip import capi
    L = capi.lib()
    assert L.exa_nvar(12345) == -1 and L.exa_nnzh64(0) == -1
    buf = (ctypes.c_double * 4)()
    assert L.exa_obj_host(999, buf, buf) == 1
    assert L.exa_hess(999, buf, buf, 1.0, buf) == 1
    assert L.exa_free(999) == 1
    assert L.exa_set_shard(999, 0, 1) == 1
    idc = ctypes.c_int(0)
    assert L.exa_new_from_table(None, ctypes.byref(idc)) == 1
    assert L.exa_plan_only(None, ctypes.byref(idc)) == 1


def test_malformed_table_is_rejected_not_crashed(libs):
    from exahip import ExaCore, capi, rng
    from exahip.core import CNode
    L = capi.lib()
    c = ExaCore()
    x = c.add_var(3)
    c.add_con(lambda i: x[i] * x[i], rng(1, 3))
    ir = c.to_ir()
    # corrupt: make the root point past the node array
    ir.patterns[0].root = 10_000
    idc = ctypes.c_int(0)
    assert L.exa_plan_only(ctypes.addressof(ir.desc), ctypes.byref(idc)) == 1
    ir = c.to_ir()
    nodes = ctypes.cast(ir.patterns[0].nodes, ctypes.POINTER(CNode))
    nodes[ir.patterns[0].n_nodes - 1].fn = 99      # unknown function id
    assert L.exa_plan_only(ctypes.addressof(ir.desc), ctypes.byref(idc)) == 1


def test_plan_only_model_cannot_compute_and_shards(libs):
    from exahip import ExaModel, capi, models
    m = ExaModel(models.luksan_vlcek_model(10), device=False)
    L = capi.lib()
    buf = (ctypes.c_double * 100)()
    assert L.exa_obj_host(m.id, buf, buf) == 1
    assert L.exa_hess_host(m.id, buf, buf, 1.0, buf) == 1
    assert L.exa_set_shard(m.id, 1, 2) == 0 and L.exa_set_shard(m.id, 2, 2) == 1 and L.exa_set_shard(m.id, 0, 0) == 1
    x0, lv, uv = np.zeros(10), np.zeros(10), np.zeros(10)
    assert L.exa_meta(m.id, x0.ctypes.data, lv.ctypes.data, uv.ctypes.data, None, None) == 0
    assert x0[0] == -1.2 and x0[1] == 1.0 and np.all(np.isinf(lv))


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback_without_device(libs):
    from exahip import ExaModel, models
    from exahip.capi import ExaHipError
    with pytest.raises(ExaHipError, match="status 2"):
        ExaModel(models.luksan_vlcek_model(10), device=True)


@pytest.mark.parametrize("name", list(ZOO))
def test_planner_matches_oracle_on_zoo(libs, name):
    from exahip import ExaModel
    import oracle
    m = ExaModel(ZOO[name](), device=False)
    o = oracle.OracleModel(m.ir)
    assert (m.meta.nvar, m.meta.ncon, m.meta.nnzj, m.meta.nnzh, m.meta.nnzg) == (o.nvar, o.ncon, o.nnzj, o.nnzh, o.nnzg)
    assert m.npatterns == o.npatterns
    for k in range(m.npatterns):
        assert m.pattern_info(k) == o.pattern_info(k), k
        assert m.pattern_comp(k, 1) == o.pattern_comp(k, 1), k
        assert m.pattern_comp(k, 2) == o.pattern_comp(k, 2), k
    mx0, mlv, muv = m.meta.x0, m.meta.lvar, m.meta.uvar
    ox0, olv, ouv, olc, ouc = o.meta()
    assert np.array_equal(mx0, ox0) and np.array_equal(mlv, olv) and np.array_equal(muv, ouv)
    assert np.array_equal(m.meta.lcon, olc) and np.array_equal(m.meta.ucon, ouc)


def test_planner_matches_oracle_on_golden_expressions(libs):
    from exprs import EXPRS
    from test_golden_oracle import build_case
    from exahip import ExaModel
    import oracle
    for name, f in EXPRS:
        for as_con in (False, True):
            m = ExaModel(build_case(f, as_con), device=False)
            o = oracle.OracleModel(m.ir)
            assert m.pattern_info(0) == o.pattern_info(0), name
            assert m.pattern_comp(0, 1) == o.pattern_comp(0, 1), name
            assert m.pattern_comp(0, 2) == o.pattern_comp(0, 2), name


@pytest.mark.parametrize("name", ["lv20", "mixed", "acopf30"])
def test_generated_module_compiles_for_gfx950(libs, name):
    from exahip import ExaModel
    m = ExaModel(ZOO[name](), device=False)
    src = m.kernel_source()
    assert 'extern "C" __global__' in src and "exa_hess" in src
    path = m.compile()
    assert os.path.exists(path) and os.path.getsize(path) > 1000


def test_module_source_is_size_independent(libs):
    """One compiled module serves every N of the same model (sizes/offsets are run-time parameters)."""
    from exahip import ExaModel, models
    a = ExaModel(models.luksan_vlcek_model(10), device=False).kernel_source()
    b = ExaModel(models.luksan_vlcek_model(12345), device=False).kernel_source()
    assert a == b


def test_kernel_build_failure_is_status_2_with_message(libs, tmp_path, monkeypatch):
    """A failing compiler (internal error) must come back as status 2 with the compiler's text, never as a crash or a
    silent fallback."""
    from exahip import ExaModel, models
    from exahip.capi import ExaHipError
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))       # empty cache -> must invoke the compiler
    monkeypatch.setenv("EXAHIP_COMPILER", "hipcc")
    monkeypatch.setenv("EXAHIP_HIPCC", "/bin/false")
    m = ExaModel(models.luksan_vlcek_model(10), device=False)
    with pytest.raises(ExaHipError, match="status 2.*hipcc failed"):
        m.compile()
    monkeypatch.setenv("EXAHIP_HIPCC", "/opt/rocm/bin/hipcc")
    path = m.compile()
    assert path.startswith(str(tmp_path)) and os.path.getsize(path) > 1000
    assert not [f for f in os.listdir(tmp_path) if f.endswith((".tmp", ".log", ".hip"))]      # no build litter
    # the in-process compiler reports the same way: a flag it cannot parse is a status-2 failure with its text
    monkeypatch.setenv("EXAHIP_COMPILER", "hiprtc")
    monkeypatch.setenv("EXAHIP_HIPCC_FLAGS", "--no-such-flag-xyz")
    with pytest.raises(ExaHipError, match="status 2.*hiprtc failed"):
        m.compile()


def test_in_process_compilation_needs_no_hipcc(libs, tmp_path, monkeypatch):
    """The default build path is hiprtc inside the process (the reference specialises in-process too, KA ext :608-653):
    with NO hipcc on the machine and an empty cache a model still compiles, into a gfx950 ELF, and the second request
    is a cache hit; the cache entry's name carries the source hash and the tool hash."""
    from exahip import ExaModel, models
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("EXAHIP_HIPCC", "/nonexistent/hipcc")
    monkeypatch.delenv("EXAHIP_COMPILER", raising=False)
    m = ExaModel(models.luksan_vlcek_model(10), device=False)
    path = m.compile()
    how, ms = m.build_info()
    assert how == "hiprtc" and ms > 0, (how, ms)
    blob = open(path, "rb").read()
    assert blob[:4] == b"\x7fELF" and int.from_bytes(blob[18:20], "little") == 224          # EM_AMDGPU
    name = capi_name(m)
    assert os.path.basename(path).startswith(name + "-") and len(name) == 4 + 32
    m2 = ExaModel(models.luksan_vlcek_model(77), device=False)                                 # same module, another N
    assert m2.compile() == path and m2.build_info()[0] == "disk"


def capi_name(m):
    return m._L.exa_module_name(m.id).decode()


def test_cache_directory_must_be_private(libs, tmp_path, monkeypatch):
    """A cache directory other users can write to is never used (a planted .hsaco would be run on the GPU): the library
    falls back to ~/.cache/exahip (created 0700) — and never to a shared location such as /tmp."""
    import stat
    from exahip import ExaModel, models
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    home = tmp_path / "home"
    home.mkdir()
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(shared))
    monkeypatch.setenv("HOME", str(home))
    monkeypatch.delenv("XDG_CACHE_HOME", raising=False)
    m = ExaModel(models.luksan_vlcek_model(10), device=False)
    path = m.compile()
    assert path.startswith(str(home / ".cache" / "exahip")), path
    assert not list(shared.iterdir())
    assert stat.S_IMODE(os.stat(home / ".cache" / "exahip").st_mode) == 0o700
    # a symlink in place of the directory is refused as well
    link = tmp_path / "link"
    link.symlink_to(home / ".cache" / "exahip")
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(link))
    monkeypatch.setenv("HOME", str(tmp_path / "home2"))
    (tmp_path / "home2").mkdir()
    assert ExaModel(models.luksan_vlcek_model(10), device=False).compile().startswith(str(tmp_path / "home2"))


def test_cache_add_refuses_what_is_not_a_code_object(libs):
    L = libs[0]
    junk = b"not an elf" * 20
    assert L.exa_cache_add(b"exa_0123", junk, len(junk)) == 1
    elf_other_machine = bytearray(b"\x7fELF\x02\x01\x01" + bytes(57))
    elf_other_machine[18] = 62                                                              # EM_X86_64
    assert L.exa_cache_add(b"exa_0123", bytes(elf_other_machine), len(elf_other_machine)) == 1
    assert L.exa_cache_add(b"other_name", junk, len(junk)) == 1


def test_out_of_range_indices_are_refused_at_build(libs):
    """An index that would be an out-of-bounds device read is the caller's error (status 1) when the model is built:
    range-affine expressions are checked at both ends, table-indexed ones point by point."""
    import numpy as np
    from exahip import ExaCore, ExaModel, Table, capi, rng

    def build(kind):
        c = ExaCore()
        x = c.add_var(10)
        th = c.add_par(3, value=1.0)
        if kind == "range_hi":
            c.add_obj(lambda i: x[i + 1] ** 2, rng(1, 10))              # x[11]
        elif kind == "range_lo":
            c.add_con(lambda i: x[i - 2] * x[i], rng(2, 10))            # x[0]
        elif kind == "table":
            c.add_obj(lambda t: t.w * x[t.i], Table(i=np.array([1, 5, 12]), w=np.ones(3)))
        elif kind == "param":
            c.add_obj(lambda i: th[i] * x[i], rng(1, 4))                # theta[4]
        elif kind == "ok":
            c.add_obj(lambda t: t.w * x[t.i] * th[3], Table(i=np.array([1, 5, 10]), w=np.ones(3)))
        return c

    assert ExaModel(build("ok"), device=False).meta.nvar == 10
    for kind, needle in (("range_hi", "variable index 11"), ("range_lo", "variable index 0"), ("table", "variable index 12"),
                         ("param", "parameter index 4")):
        with pytest.raises(capi.ExaHipError) as e:
            ExaModel(build(kind), device=False)
        assert "status 1" in str(e.value)
        assert needle in capi.lib().exa_last_error().decode()


def test_julia_shim_binds_only_declared_symbols():
    """The ccall shim cannot be executed here (no julia); at least every symbol it names must exist in the C ABI."""
    shim = open(os.path.join(ROOT, "examodels.jl_amd", "julia", "ExaModelsHIP.jl")).read()
    used = sorted(set(re.findall(r":(exa_[a-z0-9_]+)", shim)))
    declared = set(header_symbols()) | set(header_symbols("exahip_recipe.h"))
    assert len(used) >= 10 and not [u for u in used if u not in declared]


def test_julia_shim_wire_format_matches_the_header():
    """No julia here, so the shim's `lower!` cannot run — but what it WRITES is checkable statically: the opcode numbers,
    the order of the univariate / bivariate function tables (the codes are positions in those tuples) and the field order
    and types of the four wire structs must be those of include/exahip_ir.h, or every model built from Julia would be
    silently mis-decoded."""
    shim = open(os.path.join(ROOT, "examodels.jl_amd", "julia", "ExaModelsHIP.jl")).read()
    hdr = open(os.path.join(ROOT, "include", "exahip_ir.h")).read()

    def enum_names(prefix, count_name=None):
        body = next(b for b in re.findall(r"enum\s+\w+\s*\{(.*?)\}", hdr, re.S) if prefix in b)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = [t.split("=")[0].strip() for t in body.split(",") if t.strip()]
        return [n[len(prefix):].lower() for n in names if n.startswith(prefix) and n != count_name]

    # opcodes
    ops = re.search(r"const\s+(OP_[A-Z_, ]+?)\s*=\s*Int32\.\(0:(\d+)\)", shim)
    jl_ops = [o.strip()[3:].lower() for o in ops.group(1).split(",")]
    assert jl_ops == enum_names("EXA_OP_")
    assert int(ops.group(2)) == len(jl_ops) - 1
    # function tables: Julia function -> header name
    alias = {"+": "plus", "-": "minus", "*": "mul", "/": "div", "^": "pow"}
    un = re.search(r"const UN = Dict\(.*?enumerate\(\((.*?)\)\)\)", shim, re.S).group(1)
    jl_un = [alias.get(t.strip(), t.strip()) for t in un.replace("\n", " ").split(",")]
    sym = lambda name: [t.strip()[1:] for t in re.search(r"const %s = \((.*?)\)" % name, shim, re.S).group(1).replace("\n", " ").split(",")]
    assert jl_un + sym("UN_SPECIAL") == enum_names("EXA_U_", "EXA_U_COUNT")
    bn = re.search(r"const BIN = Dict\(.*?enumerate\(\((.*?)\)\)\)", shim, re.S).group(1)
    jl_bin = [{"+": "add", "-": "sub", "atan": "atan2"}.get(t.strip(), alias.get(t.strip(), t.strip())) for t in bn.split(",")]
    assert jl_bin + sym("BIN_SPECIAL") == enum_names("EXA_B_", "EXA_B_COUNT")
    # wire structs: (type, name) sequences
    ctype = {"int32_t": "Int32", "int64_t": "Int64", "double": "Float64"}

    def c_fields(name):
        body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + r"_t;", hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = " ".join(decl.replace("const", "").split())
            if not decl:
                continue
            m = re.match(r"([\w ]+?)\s*(\*?)\s*(\w+)$", decl)
            typ, ptr, field = m.group(1).strip(), m.group(2), m.group(3)
            out.append(("Ptr" if ptr else ctype[typ], field.lstrip("_")))
        return out

    def jl_fields(name):
        body = re.search(r"struct " + name + r";(.*?)end", shim, re.S).group(1)
        out = []
        for decl in body.replace("\n", " ").split(";"):
            decl = decl.strip()
            if decl:
                field, typ = decl.split("::")
                out.append(("Ptr" if typ.startswith("Ptr") else typ, field))
        return out

    for c, j in (("exa_node", "CNode"), ("exa_column", "CColumn"), ("exa_pattern", "CPattern"), ("exa_model_desc", "CModelDesc")):
        assert c_fields(c) == jl_fields(j), (c, c_fields(c), jl_fields(j))


def test_julia_shim_ccall_signatures_match_the_header():
    """Every `ccall((:exa_x, LIB), Ret, (ArgTypes...), ...)` of the shim against the prototype of exa_x in the headers:
    return type, number of arguments, and for each argument scalar-vs-pointer and the scalar's width (a Cint passed where
    the ABI takes int64_t corrupts the following arguments without any error)."""
    shim = open(os.path.join(ROOT, "examodels.jl_amd", "julia", "ExaModelsHIP.jl")).read()
    hdr = "".join(open(os.path.join(ROOT, "include", f)).read() for f in ("exahip.h", "exahip_recipe.h"))
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"\n\s*((?:const\s+)?\w+\s*\*?)\s*(exa_\w+)\s*\(([^;{]*?)\)\s*;", hdr):
        protos[name] = (ret.strip(), [] if args.strip() in ("", "void") else [a.strip() for a in args.split(",")])

    def c_class(decl):
        decl = decl.replace("const", " ")
        if "*" in decl or "[" in decl or re.search(r"\bexa_allreduce_fn\b", decl):
            return "ptr"
        typ = decl.split()[0] if len(decl.split()) > 1 else decl.strip()
        return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "double": "f64", "float": "f32", "char": "i8"}[typ]

    def jl_class(t):
        t = t.strip()
        if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring",):
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Cdouble": "f64", "Float64": "f64", "Cfloat": "f32"}[t]

    calls = re.findall(r"ccall\(\(:(exa_\w+), LIB\),\s*(\w+(?:\{\w+\})?),\s*\(([^)]*)\)", shim)
    assert len(calls) >= 25
    # the Julia snippets of INTEGRATION.md are held to the same prototypes
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    doc_calls = re.findall(r"ccall\(\(:(exa_\w+), \"libexahip\.so\"\),\s*(\w+(?:\{\w+\})?),\s*\(([^)]*)\)", doc)
    assert len(doc_calls) >= 8
    calls += doc_calls
    for name, ret, argt in calls:
        assert name in protos, name
        cret, cargs = protos[name]
        jargs = [a for a in argt.split(",") if a.strip()]
        assert len(jargs) == len(cargs), (name, jargs, cargs)
        assert jl_class(ret) == c_class(cret + " x"), (name, ret, cret)
        for ja, ca in zip(jargs, cargs):
            assert jl_class(ja) == c_class(ca), (name, ja, ca)


def test_loopfree_note_makes_every_build_arrive_at_the_same_module(libs):
    """the decision "this module's scatter kernels spill: generate them without loops" is taken where the module
    is compiled (from the code object's metadata, exa_build.cpp kernel_resources) and recorded as a NOTE of the first
    module's name; a build that finds the note — plan-only or device, a packed library's consumer (exa_cache_note) —
    generates the final module at once, under the same name, without a second compile."""
    from exahip import ExaModel, capi, models
    L = capi.lib()
    core = models.rocket_model(77)
    a = ExaModel(core, device=False)
    first = L.exa_module_name(a.id).decode()
    assert L.exa_module_alias(a.id).decode() == "" and "scatter kernels without loops" not in a.kernel_source()
    assert L.exa_cache_note(first.encode(), b"loopfree") == 0          # what a packed library does at load
    b = ExaModel(models.rocket_model(77), device=False)
    final = L.exa_module_name(b.id).decode()
    assert final != first and L.exa_module_alias(b.id).decode() == first
    assert "scatter kernels without loops" in b.kernel_source()
    c = ExaModel(models.rocket_model(77), device=False)
    assert L.exa_module_name(c.id).decode() == final                    # deterministic: the same final module every time
    assert L.exa_cache_note(first.encode(), b"") == 0                   # (leave no note behind for the other tests)


def test_kernel_resources_are_read_from_the_code_object(libs):
    """exa_compile of a small model: its module and the product-window module are both reported (exa_code_object), and a
    model whose scatter kernels fit the register file keeps its first module (no alias)."""
    from exahip import ExaModel, capi, models
    m = ExaModel(models.luksan_vlcek_model(50), device=False)
    m.compile()
    objs = m.code_objects()
    assert len(objs) == 2 and all(name.startswith("exa_") and blob[:4] == b"\x7fELF" or blob[:24] == b"__CLANG_OFFLOAD_BUNDLE__" for name, blob in objs)
    assert objs[0][0] == capi.lib().exa_module_name(m.id).decode() and capi.lib().exa_module_alias(m.id).decode() == ""
    assert b"exa_hprodw" in objs[1][1] and b"exa_jtprodw" in objs[1][1]


def test_staged_hessian_kernel_is_generated_where_the_stencil_allows(libs, tmp_path, monkeypatch):
    """exa_hesscl (x staged through LDS per wavefront): for groups whose x indices are all (unit-step range) + literal WITHIN ONE variable array
    — Luksan-Vlcek.  Models reading several arrays (cops_chain: u, x1, x2, x3; the rocket: h, v, m, tau) would need one stretch per array: built
    in round 4, slower in every A/B (profiles/r4_staging_ab.txt, r5_rocket_staging_ab.txt), not generated since round 6.  ACOPF (data-indexed)
    and a stepped range have nothing to stage.  The plain chained kernel exists for all of them."""
    from exahip import ExaModel, models
    from zoo import ZOO
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))
    lv = ExaModel(models.luksan_vlcek_model(5000), device=False).kernel_source()
    assert "exa_hesscl(" in lv and "exa_hessc(" in lv and "xs[0 + xd[0] + " in lv and "double g0_[1]" in lv
    for mk in (ZOO["cops_chain"], ZOO["acopf30"], ZOO["stepped"], lambda: models.rocket_model(500)):
        src = ExaModel(mk(), device=False).kernel_source()
        assert "exa_hessc(" in src and "exa_hesscl(" not in src
    r = ExaModel(models.rocket_model(500), device=False)
    r.compile()                                                   # ... asks the compiled kernels: everything fits, nothing is regenerated
    assert r._L.exa_module_alias_note(r.id) == b"" and all(a["fits"] and a["flags"] == "default" for a in r.build_audit())


def test_generated_module_has_the_zero_fill_and_the_folding_objective(libs):
    """exa_zero (the zero-fill under the atomics: a launch of the module instead of hipMemsetAsync) and the arrival counters of
    exa_obj (the last workgroup folds the partial sums — ONE launch at any size since round 6: sharded counters beyond 512 workgroups,
    arrivals as sc1 store + vmcnt(0) + relaxed add instead of a device-scope release) are part of every module."""
    from exahip import ExaModel, models
    src = ExaModel(models.luksan_vlcek_model(100), device=False).kernel_source()
    assert "void __launch_bounds__(EXA_BLOCK) exa_zero(double* __restrict__ out, long n)" in src
    assert "exa_obj(const long* __restrict__ P, const double* __restrict__ x, const double* __restrict__ th, double* part, unsigned* done, double* __restrict__ out)" in src
    assert "exa_publish(); last_ = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)" in src
    assert "exa_reduce_partials" not in src and "__ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT" not in src
    # Horner steps of exa_sincos take their coefficients from SGPRs
    assert 'asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k))' in src


def test_locality_order_of_a_branch_table_is_by_bus(libs):
    """exa_locality_order: the data points of a pattern in the order of the smallest variable they reach — a branch table: by its lower
    bus, the order of a case file.  Host logic (plan-only handle); the library itself never re-orders a table."""
    import numpy as np
    from exahip import ExaModel, models
    data = models.synthetic_power_data(300, 480, 40, seed=4, topology="random")
    m = ExaModel(models.ac_power_model(data), device=False)
    f, t = data["branch"].cols["f_bus"], data["branch"].cols["t_bus"]
    k = next(k for k in range(m.npatterns) if m.pattern_info(k)["n"] == 480 and m.pattern_info(k)["kind"] == 1)
    perm = m.locality_order(k)
    assert sorted(perm.tolist()) == list(range(480))
    key = np.minimum(f, t)[perm]
    assert np.all(np.diff(key) >= 0)
    assert np.array_equal(perm, np.argsort(np.minimum(f, t), kind="stable"))
    # a range pattern is in order already
    lv = ExaModel(models.luksan_vlcek_model(50), device=False)
    assert np.array_equal(lv.locality_order(0), np.arange(lv.pattern_info(0)["n"]))


def test_owner_pull_module_is_planned_for_data_indexed_models_only(libs, monkeypatch):
    """The second module of a model: product WINDOWS where every scatter target is range-affine (LV), the owner-PULL kernels where a target
    comes from a data column (ACOPF), nothing with EXAHIP_PRODUCT_PULL=0.  Host logic: plan-only handles."""
    from exahip import ExaModel, models
    from zoo import ZOO
    a = ExaModel(ZOO["acopf30"](), device=False)
    assert "owner pull available" in a.product_info("jtprod")[1] and a._L.exa_code_object_count(a.id) == 2
    src = a.module_source(1)
    assert "exa_jtpull(" in src and "exa_hppull(" in src and "exa_jtkeys(" in src and "exa_jtprodw" not in src
    assert a.product_info("jtprod")[0] == 0                       # undecided and untuned: the atomics (the pull is slower, profiles/r4_pull_ab.txt)
    lv = ExaModel(models.luksan_vlcek_model(100), device=False)
    assert "owner pull" not in lv.product_info("jtprod")[1] and "exa_jtprodw" in lv.module_source(1) and "exa_jtpull" not in lv.module_source(1)
    monkeypatch.setenv("EXAHIP_PRODUCT_PULL", "0")
    b = ExaModel(ZOO["acopf30"](), device=False)
    assert "owner pull" not in b.product_info("jtprod")[1] and b._L.exa_code_object_count(b.id) == 1 and b.module_source(1) == ""


@pytest.mark.parametrize("seed", [3021, 3003])
def test_pull_key_kernels_walk_the_block_map_like_the_scatter_kernels(libs, seed):
    """The owner-pull lists are keyed by kernels that mirror the block map of exa_jtprod / exa_hprod.  When a target is shared by
    every data point those kernels take EXA_BLOCK * 16 points per map entry; the key kernels took EXA_BLOCK (15 keys out of 16
    uninitialised: wrong J'v / Hv and memory faults on deep random models, round 4 closing sweep).  Same factor in both."""
    import randexpr
    from exahip import ExaModel
    randexpr.NPTS = 300
    m = ExaModel(randexpr.build_model(seed, 8, 4), device=False)
    main, pull = m.kernel_source(), m.module_source(1)
    assert "exa_jtkeys(" in pull and "exa_hpkeys(" in pull

    def factor(src, kernel):
        body = src[src.index("void __launch_bounds__(EXA_BLOCK) " + kernel + "("):]
        body = body[:body.index("\n}\n")]
        mm = re.search(r"tid0 = \(e_ & \(\(1L << 40\) - 1\)\) \* \(?EXA_BLOCK(?: \* (\d+)\))? \+ threadIdx\.x", body)
        assert mm, (kernel, body[:400])
        return int(mm.group(1) or 1), body
    for scatter, keys in (("exa_jtprod", "exa_jtkeys"), ("exa_hprod", "exa_hpkeys")):
        f_main, _ = factor(main, scatter)
        f_keys, body = factor(pull, keys)
        assert f_main == f_keys, (scatter, f_main, f_keys)
        if f_keys > 1:
            assert f"u < {f_keys}" in body
    assert any(factor(main, k)[0] == 16 for k in ("exa_jtprod", "exa_hprod"))          # these seeds DO have a shared target


def test_generated_module_has_the_tile_loop_kernels(libs):
    """exa_jacl / exa_consl (cons_nln! / jac_coord! as a loop over `ppt` consecutive block-map entries, both launch arguments): part of every
    module; the loop is not unrolled and the next entry is fetched through the constant address space (a scalar load) ahead of its tile."""
    from exahip import ExaModel, models
    for core in (models.luksan_vlcek_model(100), models.rocket_model(20)):
        src = ExaModel(core, device=False).kernel_source()
        for k in ("exa_jacl", "exa_consl"):
            body = src.split("__launch_bounds__(EXA_BLOCK) " + k + "(")[1].split('extern "C"')[0]
            assert "long nent, int ppt)" in body.split("{")[0]
            assert "#pragma unroll 1\n    for (int u_ = 0; u_ < ppt; u_++)" in body and "if (b >= nent) break;" in body
            assert "en_ = ((const __attribute__((address_space(4))) long*)P[" in body
