"""Property-style parity over random expression trees (tests/randexpr.py).

CPU (-m "not gpu"): the product's planner agrees with the oracle's independent implementation on the COO layout of
every random pattern.  GPU (-m gpu): all callbacks and products agree with the oracle to 1e-10."""
import numpy as np
import pytest

import randexpr
from conftest import has_gpu


@pytest.mark.parametrize("seed", range(24))
def test_random_patterns_layout(libs, seed):
    from exahip import ExaModel
    import oracle
    m = ExaModel(randexpr.build_model(seed), device=False)
    o = oracle.OracleModel(m.ir)
    assert (m.meta.nnzj, m.meta.nnzh, m.meta.nnzg) == (o.nnzj, o.nnzh, o.nnzg)
    for k in range(m.npatterns):
        assert m.pattern_info(k) == o.pattern_info(k)
        assert m.pattern_comp(k, 1) == o.pattern_comp(k, 1)
        assert m.pattern_comp(k, 2) == o.pattern_comp(k, 2)


@pytest.mark.parametrize("flavour", ["mixed", "unit", "blocks"])
@pytest.mark.parametrize("seed", range(8))
def test_random_range_models_layout_and_plans(libs, seed, flavour):
    """Range-iterated random models (the ones the windows, the gathered gradient and the staged kernels live on), plan only:
    sizes, per-pattern slot maps and first/second-order component lists equal the oracle's; the product-window planner
    returns a shape or a reason for both products; a 3-way shard's variable footprint lies inside the model."""
    from exahip import ExaModel
    import oracle
    npts = (300, 1000, 1037)[seed % 3]
    m = ExaModel(randexpr.build_range_model(seed, npts=npts, unit=flavour in ("unit", "blocks"), blocks=flavour == "blocks"), device=False)
    o = oracle.OracleModel(m.ir)
    assert (m.meta.nvar, m.meta.ncon, m.meta.nnzj, m.meta.nnzh, m.meta.nnzg) == (o.nvar, o.ncon, o.nnzj, o.nnzh, o.nnzg)
    for k in range(m.npatterns):
        assert m.pattern_info(k) == o.pattern_info(k)
        assert m.pattern_comp(k, 1) == o.pattern_comp(k, 1) and m.pattern_comp(k, 2) == o.pattern_comp(k, 2)
    for which in ("jtprod", "hprod"):
        mode, text = m.product_info(which)
        assert mode in (0, 1, 2) and text
    for r in range(3):
        m.set_shard(r, 3)
        lo, hi = m.shard_var_range()
        assert 0 <= lo <= hi <= m.meta.nvar
    m.set_shard(0, 1)


def relerr(a, ref):
    a, ref = np.asarray(a), np.asarray(ref)
    if ref.size == 0:
        return 0.0
    scale = np.maximum(np.abs(ref), 1e-3 * max(1.0, np.max(np.abs(ref))))
    return float(np.max(np.abs(a - ref) / scale))


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("seed", range(10))
def test_random_patterns_values_on_hip(libs, seed):
    from exahip import ExaModel
    import oracle
    m = ExaModel(randexpr.build_model(seed))
    o = oracle.OracleModel(m.ir)
    x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(seed + 2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(seed + 3).standard_normal(m.meta.ncon)
    tol = 1e-10
    assert abs(m.obj(x) - o.obj(x)) <= tol * max(1.0, abs(o.obj(x)))
    assert relerr(m.cons(x), o.cons(x)) <= tol
    assert relerr(m.grad(x), o.grad(x)) <= tol
    assert relerr(m.jac_coord(x), o.jac_coord(x)) <= tol
    assert relerr(m.hess_coord(x, y, 0.7), o.hess_coord(x, y, 0.7)) <= tol
    jr, jc = m.jac_structure()
    orr, oc = o.jac_structure()
    assert np.array_equal(jr, orr) and np.array_equal(jc, oc)
    hr, hc = m.hess_structure()
    orr, oc = o.hess_structure()
    assert np.array_equal(hr, orr) and np.array_equal(hc, oc)
    assert relerr(m.jprod(x, v), o.jprod(x, v)) <= tol
    assert relerr(m.jtprod(x, w), o.jtprod(x, w)) <= tol
    assert relerr(m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7)) <= tol


def _special_case(seed):
    """Random trees in which a third of the function picks come from the SpecialFunctions extension (ext/functionlist.jl), composed with the
    Base entries, fixed operands, parameters and shared augmentation rows (6 patterns of depth 3: the special routines are inlined per use,
    these are the largest kernels of the suite)."""
    core = randexpr.build_model(7000 + seed, npat=6, depth=3, special=0.35)
    x0 = np.asarray(core.to_ir().x0)
    return core, x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, len(x0))


@pytest.mark.parametrize("seed", range(6))
def test_special_function_trees_oracle_against_central_differences(libs, seed):
    """-m "not gpu": the oracle's gradient of such a model against central differences of its own objective — an AD-independent check of
    the restated table inside compositions (the single entries are pinned by tests/golden/special_golden.json)."""
    import oracle
    core, x = _special_case(seed)
    o = oracle.OracleModel(core.to_ir())
    g = o.grad(x)
    h = 1e-6
    fd = np.array([(o.obj(x + h * e) - o.obj(x - h * e)) / (2 * h) for e in np.eye(len(x))])
    assert np.all(np.isfinite(g)) and np.max(np.abs(fd - g)) <= 1e-5 * max(1.0, float(np.max(np.abs(g))))


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("seed", range(6))
def test_special_function_trees_on_hip(libs, seed):
    """Every callback of the same models on the HIP path against the oracle."""
    import oracle
    from exahip import ExaModel
    core, x = _special_case(seed)
    m = ExaModel(core)
    o = oracle.OracleModel(m.ir)
    y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(seed + 2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(seed + 3).standard_normal(m.meta.ncon)
    tol = 1e-9          # (Bessel second derivatives: differences of ocml / glibc values near their zeros)
    assert abs(m.obj(x) - o.obj(x)) <= tol * max(1.0, abs(o.obj(x)))
    for name, a, b in (("cons", m.cons(x), o.cons(x)), ("grad", m.grad(x), o.grad(x)), ("jac", m.jac_coord(x), o.jac_coord(x)),
                       ("hess", m.hess_coord(x, y, 0.7), o.hess_coord(x, y, 0.7)), ("jprod", m.jprod(x, v), o.jprod(x, v)),
                       ("jtprod", m.jtprod(x, w), o.jtprod(x, w)), ("hprod", m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7))):
        assert np.all(np.isfinite(b)), name
        assert relerr(a, b) <= tol, (name, relerr(a, b))


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("seed", [523, 541, 1011, 1013, 1029])
def test_deep_random_patterns_on_hip(libs, seed, monkeypatch):
    """Depth-6 trees over 300 data points: Hessian bodies of hundreds to thousands of SSA values per pattern, kernels at
    the 512-register limit with AGPR / scratch spills.  Regression cases of the scatter kernels (J'v, H*v by atomics),
    found by sweeps over a few hundred such models (tests/sweeps/random_model_check.py):
      523          wrong H*v and an occasional memory fault when the 16-tile loop was unrolled twice (round 1);
      541          memory fault with a lane-group peeling loop inside a 5 000-line body;
      1011, 1029   wrong / NaN entries of H*v once patterns were fused into groups — per-lane sums of shared targets
                   carried in registers across a spilling body, lanes of skipped wavefronts masked instead of branched;
      1013         J'v garbage after another kernel had run (stale spill space), NaN on some boxes.
    What the generator does about it: the whole-wavefront skip is a scalar branch (readfirstlane), fused groups of the
    products are capped by raw contributions, bodies past 1000 lines (kHugeBody) get no loops, and a module whose
    compiled scatter kernels spill (scratch, or more than 256 registers) is generated again without loops."""
    from exahip import ExaModel
    import oracle
    monkeypatch.setattr(randexpr, "NPTS", 300)
    m = ExaModel(randexpr.build_model(seed, 12, 6))
    o = oracle.OracleModel(m.ir)
    m.set_product_mode(0, 0)
    x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(seed + 2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(seed + 3).standard_normal(m.meta.ncon)
    tol = 1e-10
    for _ in range(3):                                   # the failure was intermittent
        assert relerr(m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7)) <= tol
        assert relerr(m.jtprod(x, w), o.jtprod(x, w)) <= tol
    assert relerr(m.grad(x), o.grad(x)) <= tol
    assert relerr(m.hess_coord(x, y, 0.7), o.hess_coord(x, y, 0.7)) <= tol
    assert relerr(m.jac_coord(x), o.jac_coord(x)) <= tol
    assert relerr(m.cons(x), o.cons(x)) <= tol
    # all five from one sweep: the objective patterns add their first partials inside the (spilling) fused kernel with
    # wavefront-level adds — the same kind of code as the scatter kernels above
    import torch
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    for _ in range(3):
        f, g, c, j, h = m.eval_all(xd, yd, 0.7)
        torch.cuda.synchronize()
        assert abs(f.item() - o.obj(x)) <= tol * max(1.0, abs(o.obj(x)))
        assert relerr(g.cpu().numpy(), o.grad(x)) <= tol and relerr(h.cpu().numpy(), o.hess_coord(x, y, 0.7)) <= tol
        assert relerr(c.cpu().numpy(), o.cons(x)) <= tol and relerr(j.cpu().numpy(), o.jac_coord(x)) <= tol


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("seed", range(6))
def test_random_range_patterns_on_hip(libs, seed):
    """Random trees over RANGE iterators (unit and stepped, ~1000 points, `range + c` indices): the gathered gradient,
    the LDS-window scatter of J'v / Hv, multi-tile flushes, partial wavefronts, the fused sweep and a 3-way shard."""
    import torch
    from exahip import ExaModel
    import oracle
    m = ExaModel(randexpr.build_range_model(seed))
    o = oracle.OracleModel(m.ir)
    x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(seed + 2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(seed + 3).standard_normal(m.meta.ncon)
    tol = 1e-10
    H = o.hess_coord(x, y, 0.7)
    assert abs(m.obj(x) - o.obj(x)) <= tol * max(1.0, abs(o.obj(x)))
    assert relerr(m.cons(x), o.cons(x)) <= tol and relerr(m.grad(x), o.grad(x)) <= tol
    assert relerr(m.jac_coord(x), o.jac_coord(x)) <= tol and relerr(m.hess_coord(x, y, 0.7), H) <= tol
    assert relerr(m.jprod(x, v), o.jprod(x, v)) <= tol and relerr(m.jtprod(x, w), o.jtprod(x, w)) <= tol
    assert relerr(m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7)) <= tol
    for a, b in zip(m.jac_structure() + m.hess_structure(), o.jac_structure() + o.hess_structure()):
        assert np.array_equal(a, b)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    f, c, j, h = m.eval_fused(xd, yd, 0.7)
    assert relerr(h.cpu().numpy(), H) <= tol and relerr(j.cpu().numpy(), o.jac_coord(x)) <= tol
    from conftest import RankReplay
    acc = torch.zeros(m.meta.nnzh, dtype=torch.float64, device="cuda")
    rr = RankReplay(m, [("grad", m.meta.nvar)], "cuda")       # grad!: owner pieces (gathered objective) or partial sums
    try:
        for r in range(3):
            m.set_shard(r, 3)
            part = torch.zeros_like(acc)
            acc += m.hess_coord(xd, yd, 0.7, out=part)
            rr.add("grad", lambda out: m.grad(xd, out=out))
    finally:
        m.set_shard(0, 1)
    assert relerr(acc.cpu().numpy(), H) <= tol and relerr(rr.result("grad"), o.grad(x)) <= tol
