"""Recipes, the schema/builder ABI and the model file format (include/exahip_recipe.h).

The checks restate ExaModelsCompiler/test/runtests.jl: a recipe instantiated at a size and data it was never
traced on must equal the model built directly at that size and data (:611-629), instances are independent
(:299-307), the schema / argtype strings are the published ones (:603-609, :651-657), the one-integer and the
builder surfaces are disjoint (:593-601), named blocks report this instance's layout (:676-700).
CPU half: libexahip plans the instance (no device), `exa_describe` hands the resulting pattern table to the test
oracle, which is compared with the oracle on the directly built model.  `gpu` half: same through the HIP path."""
import ctypes

import numpy as np
import pytest

from conftest import has_gpu
from recipezoo import S_ARGS, S_EX, S_N, build_knob, build_layout, build_struct, lv_recipe_builder, make
from zoo import ZOO

needs_gpu = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _oracle_of(m):
    import oracle
    return oracle.OracleModel(m.describe() if m.ir is None else m.ir)


def _same(o, ref, nvar, ncon):
    """runtests.jl:280-297: x = linspace(0.5, 3, n), y = linspace(-1, 1, ncon), obj_weight = 0.5"""
    assert (o.nvar, o.ncon, o.nnzj, o.nnzh) == (ref.nvar, ref.ncon, ref.nnzj, ref.nnzh) and (o.nvar, o.ncon) == (nvar, ncon)
    x, y = np.linspace(0.5, 3.0, nvar), np.linspace(-1.0, 1.0, ncon)
    for a, b in zip(o.meta(), ref.meta()):
        assert np.array_equal(a, b)
    # (equal_nan: this x leaves the domain of some functions of the `specialfn` zoo model — both sides then hold the same NaNs)
    assert np.array_equal(o.obj(x), ref.obj(x), equal_nan=True)
    for f in ("grad", "cons", "jac_coord"):
        assert np.array_equal(getattr(o, f)(x), getattr(ref, f)(x), equal_nan=True), f
    assert np.array_equal(o.hess_coord(x, y, 0.5), ref.hess_coord(x, y, 0.5), equal_nan=True)
    for a, b in zip(o.jac_structure() + o.hess_structure(), ref.jac_structure() + ref.hess_structure()):
        assert np.array_equal(a, b)


# ---- CPU -------------------------------------------------------------------------------------------------------
def test_structured_recipe_equals_direct_build(libs):
    from exahip import ExaModel, Recipe
    rec = Recipe(make(build_struct, S_EX, True))
    assert rec.nargs == 4
    assert rec.argtype == "int|arg1,Vector{f64}|v0,Vector{f64}|lo,Table{i::int w::f64 s::f64}|arg3"     # runtests.jl:656-657
    for needle in ('"arg1"', '"v0"', '"lo"',
                   '{"name":"arg3","kind":"table","columns":[{"name":"i","type":"i64"},{"name":"w","type":"f64"},'
                   '{"name":"s","type":"f64"}]}'):                                                          # runtests.jl:603-609
        assert needle in rec.schema_json
    m = ExaModel(rec, *S_ARGS, device=False)                       # NamedTuple spelling: ExaModel(core, n, (v0=, lo=), tab)
    ref = _oracle_of(ExaModel(make(build_struct, S_ARGS, False), device=False))
    _same(_oracle_of(m), ref, S_N, S_N - 1)
    flat = (S_ARGS[0], S_ARGS[1]["v0"], S_ARGS[1]["lo"], S_ARGS[2])
    _same(_oracle_of(rec.instantiate(*flat, device=False)), ref, S_N, S_N - 1)   # the consumers' flat spelling
    # a second instance at the example's own size; the first one undisturbed (runtests.jl:631-638)
    m2 = ExaModel(rec, *S_EX, device=False)
    assert m2.meta.nvar == 4 and m.meta.nvar == S_N
    _same(_oracle_of(m2), _oracle_of(ExaModel(make(build_struct, S_EX, False), device=False)), 4, 3)


def test_one_integer_recipe_and_independent_instances(libs):
    from exahip import ExaModel, Recipe
    rec = Recipe(make(build_knob, (4,), True))
    assert (rec.nargs, rec.argtype) == (1, "int|size")
    m1, m2 = ExaModel(rec, 6, device=False), ExaModel(rec, 11, device=False)
    assert (m1.meta.nvar, m2.meta.nvar) == (6, 11)
    o1, o2 = _oracle_of(m1), _oracle_of(m2)
    assert o1.obj(np.ones(6)) == 6.0 and o2.obj(np.ones(11)) == 11.0              # runtests.jl:303-306
    for n, o in ((6, o1), (11, o2)):
        _same(o, _oracle_of(ExaModel(make(build_knob, (n,), False), device=False)), n, n - 1)
    assert np.all(m1.meta.lcon == 3.0) and np.all(np.isinf(m1.meta.ucon))


def test_named_blocks_report_this_instance(libs):
    """runtests.jl:676-700 (`pbuild` at 3, not at the example's size)"""
    from exahip import ExaModel, Recipe
    rec = Recipe(make(build_layout, (5,), True))
    m = ExaModel(rec, 3, device=False)
    bs = m.blocks()
    assert [b.name for b in bs] == ["y", "w", "link", "dax"]
    assert [b.kind for b in bs] == [0, 2, 1, 1]
    assert bs[0].dims == [3, 2] and bs[0].length == 6 and bs[0].offset == 0
    assert bs[3].length == 4 and bs[3].offset == 3
    _same(_oracle_of(m), _oracle_of(ExaModel(make(build_layout, (3,), False), device=False)), 6, 7)
    # parameters are live instance state (ExaModelsCompiler.jl:1529-1535): get / set by block, status 3 on a wrong length
    assert np.array_equal(m.get_value_block(1), np.ones(3))
    m.set_value_block(1, [2.0, 3.0, 4.0])
    assert np.array_equal(m.get_value_block(1), [2.0, 3.0, 4.0])
    L = m._L
    buf = (ctypes.c_double * 2)()
    assert L.exa_get_value_block(m.id, 1, buf, 2) == 3
    assert L.exa_get_value_block(m.id, 0, buf, 6) == 1          # block 0 is a variable, not a parameter


@pytest.mark.parametrize("name", list(ZOO))
def test_model_file_round_trip(libs, name, tmp_path):
    """A concrete core's recipe bytes are the model file format: save -> load -> plan == the direct build."""
    from exahip import ExaModel, Recipe
    core = ZOO[name]()
    Recipe(core).save(tmp_path / "m.exarcp")
    rec = Recipe.load(tmp_path / "m.exarcp")
    assert rec.nargs == 0 and rec.argtype == ""
    m = rec.instantiate(device=False)
    direct = ExaModel(core, device=False)
    assert m.kernel_source() == direct.kernel_source()
    import oracle
    _same(_oracle_of(m), oracle.OracleModel(direct.ir), direct.meta.nvar, direct.meta.ncon)


def test_headline_model_from_one_recipe(libs):
    from exahip import ExaModel, Recipe, models
    rec = Recipe(make(lv_recipe_builder, (7,), True))
    for n in (3, 20, 1000):
        o = _oracle_of(ExaModel(rec, n, device=False))
        ref = _oracle_of(ExaModel(models.luksan_vlcek_model(n), device=False))
        x, y = np.linspace(0.5, 3.0, n), np.linspace(-1.0, 1.0, n - 2)
        assert (o.nnzj, o.nnzh) == (ref.nnzj, ref.nnzh) == (3 * (n - 2), 9 * n - 15)
        assert np.array_equal(o.hess_coord(x, y, 0.5), ref.hess_coord(x, y, 0.5))
        assert np.array_equal(o.jac_coord(x), ref.jac_coord(x)) and o.obj(x) == ref.obj(x)


def test_argument_test_models(libs):
    """test/ArgumentTest/ArgumentTest.jl:185-229 — sizes, starts and bounds supplied at instantiation; one core
    instantiated twice without being consumed."""
    from exahip import ExaCore, ExaModel, Recipe, rng

    # "a model built against two argument objects" (:185-200)
    c = ExaCore(examples=(dict(N=2), dict(lo=np.zeros(2), v=np.ones(2))))
    sz, dat = c.args
    y = c.add_var(sz.N, lvar=dat.lo, start=dat.v)
    c.add_obj(lambda i: y[i] ** 2, rng(1, sz.N))
    rec = Recipe(c)
    m = ExaModel(rec, dict(N=3), dict(lo=[-1.0, -2.0, -3.0], v=[4.0, 5.0, 6.0]), device=False)
    assert m.meta.nvar == 3 and list(m.meta.lvar) == [-1.0, -2.0, -3.0] and list(m.meta.x0) == [4.0, 5.0, 6.0]
    assert _oracle_of(m).obj(m.meta.x0) == 77.0
    m2 = ExaModel(rec, dict(N=2), dict(lo=[0.0, 0.0], v=[1.0, 2.0]), device=False)
    assert m2.meta.nvar == 2 and list(m2.meta.x0) == [1.0, 2.0]

    # "building a model against arg" (:202-229)
    c = ExaCore(examples=(dict(N=2, v=np.ones(2)),))
    (arg,) = c.args
    y = c.add_var(arg.N, start=arg.v)
    c.add_obj(lambda i: y[i] ** 2, rng(1, arg.N))
    c.add_con(lambda i: y[i] - 2.0, rng(1, arg.N), lcon=0.0, ucon=0.0)
    rec = Recipe(c)
    m = ExaModel(rec, dict(N=3, v=[1.0, 2.0, 3.0]), device=False)
    assert (m.meta.nvar, m.meta.ncon) == (3, 3) and _oracle_of(m).obj(m.meta.x0) == 14.0
    m2 = ExaModel(rec, dict(N=2, v=[4.0, 5.0]), device=False)
    assert m2.meta.nvar == 2 and _oracle_of(m2).obj(m2.meta.x0) == 41.0 and list(m.meta.x0) == [1.0, 2.0, 3.0]
    cb = ExaCore(examples=(dict(N=3, lo=np.zeros(3)),))
    (arg,) = cb.args
    cb.add_var(arg.N, lvar=arg.lo, uvar=0.0)
    mb = ExaModel(cb, dict(N=2, lo=[-1.0, -2.0]), device=False)
    assert list(mb.meta.lvar) == [-1.0, -2.0] and list(mb.meta.uvar) == [0.0, 0.0]


def test_deferred_real_coefficient_and_second_block_offset(libs):
    """ArgumentTest.jl:231-307 — h = 1 / (N + 1) is a loop-invariant REAL computed from the size and used as a
    coefficient; the offset of a block behind a block of deferred length is deferred too."""
    from exahip import ExaCore, ExaModel, Recipe, rng

    def build_h(c, N):
        h = 1 / (N + 1)
        x = c.add_var(N, start=1.0)
        c.add_obj(lambda i: h * (x[i] - 2.0) ** 2, rng(1, N))
        c.add_con(lambda i: h * x[i] + x[i], rng(1, N), lcon=0.0, ucon=1.0)
        return c

    c = ExaCore(examples=(4,))
    rec = Recipe(build_h(c, c.args[0]))
    for n, pt in ((6, np.arange(0.1, 0.65, 0.1)), (9, np.arange(0.1, 0.95, 0.1))):
        ref = _oracle_of(ExaModel(build_h(ExaCore(), n), device=False))
        o = _oracle_of(ExaModel(rec, n, device=False))
        mult = pt[::-1].copy()
        assert (o.nvar, o.nnzj, o.nnzh) == (ref.nvar, ref.nnzj, ref.nnzh)
        assert o.obj(pt) == ref.obj(pt) and np.array_equal(o.grad(pt), ref.grad(pt)) and np.array_equal(o.cons(pt), ref.cons(pt))
        assert np.array_equal(o.jac_coord(pt), ref.jac_coord(pt)) and np.array_equal(o.hess_coord(pt, mult, 1.0), ref.hess_coord(pt, mult, 1.0))
        assert rec.matches(build_h(ExaCore(), n), n)
    # the coefficient really is deferred: a baked-in h would make the two sizes agree on a common point
    o6, o9 = _oracle_of(ExaModel(rec, 6, device=False)), _oracle_of(ExaModel(rec, 9, device=False))
    assert o6.obj(np.full(6, 0.5)) / 6 != o9.obj(np.full(9, 0.5)) / 9

    def build_z(c, N):
        x = c.add_var(N, start=1.0)
        z = c.add_var(3, start=2.0)
        c.add_obj(lambda j: z[j] ** 2, rng(1, 3))
        c.add_con(lambda i: x[i] + z[1], rng(1, N), lcon=0.0, ucon=np.inf)
        return c

    c = ExaCore(examples=(5,))
    rec = Recipe(build_z(c, c.args[0]))
    for n in (4, 7):
        assert rec.matches(build_z(ExaCore(), n), n)
        o, ref = _oracle_of(ExaModel(rec, n, device=False)), _oracle_of(ExaModel(build_z(ExaCore(), n), device=False))
        pt = np.arange(1.0, n + 4) / 10
        assert o.nvar == n + 3 and o.obj(pt) == ref.obj(pt) and np.array_equal(o.cons(pt), ref.cons(pt))
        assert all(np.array_equal(a, b) for a, b in zip(o.jac_structure(), ref.jac_structure()))


def _acopf_recipe():
    from exahip import ExaCore, Recipe, models
    ex = models.synthetic_power_data(nbus=5, nbr=6, ngen=2, seed=1)
    c = ExaCore(examples=(ex,))
    models.ac_power_model(c.args[0], core=c)
    return Recipe(c)


def test_data_defined_acopf_recipe(libs):
    """The shape the builder ABI exists for (ExaModelsCompiler.jl:58-66, test/NLPTest/power.jl:112-213): every size a
    table length, every bound a data field; traced on a 5-bus example, instantiated on a 30-bus network."""
    from exahip import ExaModel, models
    rec = _acopf_recipe()
    assert rec.nargs == 15 and rec.argtype.startswith("Table{i::int pd::f64 gs::f64 qd::f64 bs::f64}|bus,")
    assert "Vector{int}|ref_buses" in rec.argtype
    d = models.synthetic_power_data(nbus=30, nbr=41, ngen=6, seed=3)
    assert rec.matches(models.ac_power_model(d), d)
    m = ExaModel(rec, d, device=False)
    assert [(b.name, b.offset, b.length) for b in m.blocks()] == \
        [("va", 0, 30), ("vm", 30, 30), ("pg", 60, 6), ("qg", 66, 6), ("p", 72, 82), ("q", 154, 82)]
    direct = ExaModel(models.ac_power_model(d), device=False)
    assert m.kernel_source() == direct.kernel_source()
    for k in range(m.npatterns):
        assert m.pattern_info(k) == direct.pattern_info(k)


def test_builder_error_paths(libs):
    from exahip import Recipe, capi
    L = capi.lib()
    rec = Recipe(make(build_struct, S_EX, True))
    assert L.exa_recipe_new(rec.id, 6) == 0                      # structured: `P_new` is not its entry point
    b = L.exa_data_begin(rec.id)
    assert b > 0 and L.exa_data_ready(b) == 0
    one = (ctypes.c_double * 1)(1.0)
    assert L.exa_set_scalar_i64(b, b"nope", 3) == 1              # unknown field
    assert L.exa_set_scalar_f64(b, b"arg1", 3.0) == 1            # right name, wrong type
    assert L.exa_set_array_f64(b, b"arg1", one, 1) == 1          # right name, wrong kind
    assert L.exa_set_col_f64(b, b"arg3", b"i", one, 1) == 1      # column i is i64
    assert L.exa_set_scalar_i64(9999, b"arg1", 3) == 1           # bad builder id
    assert L.exa_new_from_data(b) == 0 and L.exa_plan_from_data(b) == 0     # incomplete
    # complete but inconsistent: v0 has 5 entries for a size of 6
    v5, v6 = (ctypes.c_double * 5)(), (ctypes.c_double * 6)()
    i3 = (ctypes.c_int64 * 3)(2, 5, 6)
    w3 = (ctypes.c_double * 3)(1.5, 3.0, 0.5)
    assert L.exa_set_scalar_i64(b, b"arg1", 6) == 0
    assert L.exa_set_array_f64(b, b"v0", v5, 5) == 0 and L.exa_set_array_f64(b, b"lo", v6, 6) == 0
    assert L.exa_set_col_i64(b, b"arg3", b"i", i3, 3) == 0 and L.exa_set_col_f64(b, b"arg3", b"w", w3, 3) == 0
    assert L.exa_data_ready(b) == 0                              # column s still missing
    assert L.exa_set_col_f64(b, b"arg3", b"s", w3, 2) == 0
    assert L.exa_data_ready(b) == 0                              # columns disagree in length
    assert L.exa_set_col_f64(b, b"arg3", b"s", w3, 3) == 0
    assert L.exa_data_ready(b) == 1
    assert L.exa_plan_from_data(b) == 0
    assert b"x0 block expects 6 values" in L.exa_last_error()
    assert L.exa_set_array_f64(b, b"v0", v6, 6) == 0
    mid = L.exa_plan_from_data(b)
    assert mid > 0 and L.exa_nvar(mid) == 6 and L.exa_ncon(mid) == 5
    L.exa_free(mid)
    assert L.exa_data_free(b) == 0 and L.exa_data_free(b) == 1
    # malformed bytes are refused, not crashed on
    good = rec.bytes
    assert L.exa_recipe_load(good[:40], 40) == 0 and L.exa_recipe_load(b"NOTARCP0" + good[8:], len(good)) == 0
    assert L.exa_recipe_load(good + b"x", len(good) + 1) == 0
    for cut in range(8, len(good), max(1, len(good) // 97)):
        assert L.exa_recipe_load(good[:cut], cut) == 0


def test_mutated_recipes_never_crash_the_library(libs):
    """Recipe bytes are external input: random byte mutations must end in a refusal (0) or in a valid model, never in
    a crash.  (A recipe that still parses is instantiated on the planner only.)"""
    from exahip import Recipe, capi
    L = capi.lib()
    good = Recipe(make(build_struct, S_EX, True)).bytes
    n, dat, tab = S_ARGS
    bound = [(b"v0", np.ascontiguousarray(dat["v0"])), (b"lo", np.ascontiguousarray(dat["lo"]))]
    cols = [(k.encode(), np.ascontiguousarray(v)) for k, v in tab.cols.items()]
    r = np.random.default_rng(0)
    loaded = built = 0
    for _ in range(600):
        b = bytearray(good)
        for _ in range(int(r.integers(1, 4))):
            b[int(r.integers(8, len(b)))] = int(r.integers(0, 256))
        rid = L.exa_recipe_load(bytes(b), len(b))
        if rid <= 0:
            continue
        loaded += 1
        bid = L.exa_data_begin(rid)
        L.exa_set_scalar_i64(bid, b"arg1", n)
        for name, a in bound:
            L.exa_set_array_f64(bid, name, a.ctypes.data, a.size)
        for name, a in cols:
            (L.exa_set_col_i64 if a.dtype.kind == "i" else L.exa_set_col_f64)(bid, b"arg3", name, a.ctypes.data, a.size)
        mid = L.exa_plan_from_data(bid)
        if mid > 0:
            built += 1
            assert L.exa_nvar(mid) >= 0
            L.exa_free(mid)
        L.exa_data_free(bid)
        L.exa_recipe_free(rid)
    assert loaded > 100 and built > 50          # the mutations do reach the planner, they are not all refused up front


def test_recipe_refuses_what_it_cannot_defer(libs):
    from exahip import ExaCore
    from exahip.recipe import RecipeError
    c = ExaCore(examples=(5,))
    (N,) = c.args
    x = c.add_var(N)
    with pytest.raises(RecipeError):
        c.add_obj(lambda i: (N % 2) * x[i], range(1, 6))         # an operation with no deferred form
    with pytest.raises(TypeError):
        c.add_obj(lambda i: x[i] ** N, range(1, 6))               # a size as a literal exponent
    with pytest.raises(RecipeError):
        c.add_var(N, start=np.ones(5))                            # inline data for a block of deferred length


def test_matches_catches_a_baked_in_example(libs):
    """`1.0 / N` cannot be intercepted in Python (float arithmetic takes the tagged int as a plain one); the recipe
    then carries the EXAMPLE's 1/N.  Recipe.matches at a second size is the check that exposes it."""
    from exahip import ExaCore, Recipe

    def build(c, N, leak):
        x = c.add_var(N, start=1.0)
        c.add_obj(lambda i: ((1.0 / N) if leak else 0.25) * (x[i] - 2.0) ** 2, rng_(1, N))
        return c

    from exahip import rng as rng_
    for leak in (False, True):
        c = ExaCore(examples=(4,))
        rec = Recipe(build(c, c.args[0], leak))
        assert rec.matches(build(ExaCore(), 4, leak), 4)                     # at the example itself: always equal
        assert rec.matches(build(ExaCore(), 9, leak), 9) == (not leak)       # elsewhere: only the sound recipe
    assert Recipe(make(build_struct, S_EX, True)).matches(make(build_struct, S_ARGS, False), *S_ARGS)
    assert not Recipe(make(build_struct, S_EX, True)).matches(make(build_struct, S_EX, False), *S_ARGS)


# ---- GPU -------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_hip_structured_recipe_instance(libs):
    from exahip import ExaModel, Recipe
    import oracle
    rec = Recipe(make(build_struct, S_EX, True))
    m = ExaModel(rec, *S_ARGS)
    ref = oracle.OracleModel(make(build_struct, S_ARGS, False).to_ir())
    x, y = np.linspace(0.5, 3.0, S_N), np.linspace(-1.0, 1.0, S_N - 1)
    assert (m.meta.nvar, m.meta.ncon) == (S_N, S_N - 1)
    np.testing.assert_allclose(m.obj(x), ref.obj(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.grad(x), ref.grad(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.cons(x), ref.cons(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.jac_coord(x), ref.jac_coord(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.hess_coord(x, y, 0.5), ref.hess_coord(x, y, 0.5), rtol=1e-10, atol=1e-12)
    for a, b in zip(m.jac_structure() + m.hess_structure(), ref.jac_structure() + ref.hess_structure()):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_hip_acopf_recipe_at_config4_scale(libs):
    """BASELINE configs[3] instantiated through the builder ABI from the 5-bus recipe: same module, same numbers."""
    import torch
    from exahip import ExaModel, models
    rec = _acopf_recipe()
    d = models.synthetic_power_data(nbus=78484, nbr=126015, ngen=6800, seed=0)
    m = ExaModel(rec, d)
    core = models.ac_power_model(d)
    direct = ExaModel(core)
    assert m._L.exa_code_object_path(m.id) == direct._L.exa_code_object_path(direct.id)
    assert (m.meta.nvar, m.meta.ncon, m.meta.nnzj, m.meta.nnzh) == (direct.meta.nvar, direct.meta.ncon, direct.meta.nnzj, direct.meta.nnzh)
    x = torch.from_numpy(models.acopf_start(core)).cuda()
    y = torch.from_numpy(np.random.default_rng(1).standard_normal(m.meta.ncon)).cuda()
    assert torch.equal(m.hess_coord(x, y, 0.5), direct.hess_coord(x, y, 0.5))
    assert torch.equal(m.jac_coord(x), direct.jac_coord(x)) and torch.equal(m.cons(x), direct.cons(x))
    hr, hc = m.hess_structure()
    dr, dc = direct.hess_structure()
    assert np.array_equal(hr, dr) and np.array_equal(hc, dc)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_hip_recipe_instances_share_one_kernel_module(libs):
    """The generated module is size-independent, so every instance of a recipe reuses one cached code object."""
    from exahip import ExaModel, Recipe, models
    import oracle
    rec = Recipe(make(lv_recipe_builder, (7,), True))
    paths = set()
    for n in (50, 1000, 100000):
        m = ExaModel(rec, n)
        paths.add(m._L.exa_code_object_path(m.id))
        ref = oracle.OracleModel(models.luksan_vlcek_model(n).to_ir())
        x, y = np.linspace(0.5, 1.5, n), np.linspace(-1.0, 1.0, n - 2)
        np.testing.assert_allclose(m.hess_coord(x, y, 0.5), ref.hess_coord(x, y, 0.5), rtol=1e-10, atol=1e-9)
    assert len(paths) == 1


def test_two_dimensional_recipe_with_products_and_tuple_targets(libs):
    """The split Luksan-Vlcek model (test/NLPTest/luksan.jl:17-26) as a recipe in TWO sizes: 2-D variable strides,
    product iterators and the column-major row of an augmentation target are all deferred."""
    from exahip import ExaCore, Recipe, product, rng
    from exahip.graph import exp, sin

    def build(c, N, M):
        x = c.add_var(N, M, start=0.7, name="x")

        def con1(p):
            i, j = p
            return 3 * x[i + 1, j] ** 3 + 2 * x[i + 2, j] - 5

        def con2(p):
            i, j = p
            return ((i, j), sin(x[i + 1, j] - x[i + 2, j]) * sin(x[i + 1, j] + x[i + 2, j]) + 4 * x[i + 1, j]
                    - x[i, j] * exp(x[i, j] - x[i + 1, j]) - 3)

        def obj(p):
            i, j = p
            return 100 * (x[i - 1, j] ** 2 - x[i, j]) ** 2 + (x[i - 1, j] - 1) ** 2

        s = c.add_con(con1, product(rng(1, N - 2), rng(1, M)), name="s")
        c.add_con_aug(s, con2, product(rng(1, N - 2), rng(1, M)))
        c.add_obj(obj, product(rng(2, N), rng(1, M)))
        e = c.add_con(N, M, lcon=-1.0, ucon=1.0)                                   # an empty 2-D constraint block ...
        c.add_con_aug(e, lambda p: ((p[0], p[1]), x[p[0], p[1]] ** 2), product(rng(1, N, 2), rng(M, 1, -1)))   # stepped / reversed axes
        return c

    c = ExaCore(examples=(5, 2))
    rec = Recipe(build(c, *c.args))
    assert rec.argtype == "int|arg1,int|arg2"
    for n, mm in ((5, 2), (9, 1), (4, 7), (20, 3)):
        assert rec.matches(build(ExaCore(), n, mm), n, mm), (n, mm)
    from exahip import ExaModel
    m = ExaModel(rec, 6, 4, device=False)
    b = {blk.name: blk for blk in m.blocks()}
    assert b["x"].dims == [6, 4] and b["s"].dims == [4, 4] and b["s"].length == 16
