"""No kernel of the library may read a register lane it did not write.

tests/sweeps/poison_registers.py found the mechanism behind "wrong, run-to-run different" window sums of an over-sized kernel
(profiles/NOTES.md): the compiled code read stale register lanes — contents of whatever kernel ran before.  Such a read is
invisible as long as the stale contents happen to be harmless; here every architectural VGPR and every AGPR of the chip is
filled with a NaN pattern first (a 512-register asm kernel over 4096 workgroups), then every callback runs and must still
equal the oracle.  Small models, deep random models (kernels with AGPR / scratch spills), random range models (windows, chunk
loops) and the benchmark models are covered; tests/sweeps/range_model_check.py --poison sweeps more seeds the same way."""
import numpy as np
import pytest

import randexpr
from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


@pytest.fixture(scope="module")
def poison(tmp_path_factory):
    from conftest import PREBUILD
    if PREBUILD:                 # (kernel prebuild on the CPU, tests/conftest.py: nothing runs, the poison is never called)
        return lambda: None
    from poison import make_poison
    return make_poison(str(tmp_path_factory.mktemp("poison")))


def relerr(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if ref.size == 0:
        return 0.0
    fin = np.isfinite(ref)
    if not fin.any():
        return 0.0
    if not np.all(np.isfinite(a[fin])):
        return float("inf")
    scale = np.maximum(np.abs(ref[fin]), 1e-3 * max(1.0, float(np.max(np.abs(ref[fin])))))
    return float(np.max(np.abs(a[fin] - ref[fin]) / scale))


def check_all(m, o, x, y, v, w, poison, tol):
    """every callback, each one right after a poisoning of the register files"""
    import torch
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    calls = [("obj", lambda: np.array([m.obj(x)]), lambda: np.array([o.obj(x)])), ("cons", lambda: m.cons(x), lambda: o.cons(x)),
             ("grad", lambda: m.grad(x), lambda: o.grad(x)), ("jac", lambda: m.jac_coord(x), lambda: o.jac_coord(x)),
             ("hess", lambda: m.hess_coord(x, y, 0.7), lambda: o.hess_coord(x, y, 0.7)), ("jprod", lambda: m.jprod(x, v), lambda: o.jprod(x, v)),
             ("jtprod", lambda: m.jtprod(x, w), lambda: o.jtprod(x, w)), ("hprod", lambda: m.hprod(x, y, v, 0.7), lambda: o.hprod(x, y, v, 0.7))]
    for name, run, ref in calls:
        poison()
        assert relerr(run(), ref()) <= tol, name
    poison()
    f, g, c, j, h = m.eval_all(xd, yd, 0.7)
    torch.cuda.synchronize()
    assert relerr(g.cpu().numpy(), o.grad(x)) <= tol and relerr(h.cpu().numpy()[:m.meta.nnzh], o.hess_coord(x, y, 0.7)) <= tol
    assert relerr(c.cpu().numpy()[:m.meta.ncon], o.cons(x)) <= tol and relerr(j.cpu().numpy()[:m.meta.nnzj], o.jac_coord(x)) <= tol
    assert relerr(np.array([f.item()]), np.array([o.obj(x)])) <= tol


@pytest.mark.parametrize("name", list(ZOO))
def test_zoo_after_poison(libs, poison, name):
    from exahip import ExaModel
    import oracle
    m = ExaModel(ZOO[name]())
    o = oracle.OracleModel(m.ir)
    x, y, _ = point(m.meta.x0, m.meta.ncon, seed=31)
    v = np.random.default_rng(32).standard_normal(m.meta.nvar)
    w = np.random.default_rng(33).standard_normal(m.meta.ncon)
    check_all(m, o, x, y, v, w, poison, 1e-10)


@pytest.mark.parametrize("seed", [523, 1011])
def test_deep_random_models_after_poison(libs, poison, seed, monkeypatch):
    from exahip import ExaModel
    import oracle
    monkeypatch.setattr(randexpr, "NPTS", 300)
    m = ExaModel(randexpr.build_model(seed, 12, 6))
    o = oracle.OracleModel(m.ir)
    x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(seed + 2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(seed + 3).standard_normal(m.meta.ncon)
    check_all(m, o, x, y, v, w, poison, 1e-9)


@pytest.mark.parametrize("seed,flavour", [(8, ""), (1, "blocks"), (227, "blocks")])
def test_random_range_models_after_poison(libs, poison, seed, flavour):
    """(1, blocks) is the model whose Hv window kernel read stale lanes; the library now refuses that kernel, and what runs
    instead must be clean as well"""
    from exahip import CompressedExaModel, ExaModel
    import oracle
    import torch
    m = ExaModel(randexpr.build_range_model(seed, npts=1000, unit=flavour == "blocks", blocks=flavour == "blocks"))
    o = oracle.OracleModel(m.ir)
    x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(seed + 2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(seed + 3).standard_normal(m.meta.ncon)
    check_all(m, o, x, y, v, w, poison, 1e-9)
    # the compressed COO (windowed sweep on these models), densified
    cm = CompressedExaModel(m)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    for kind, nrow in (("jac", max(m.meta.ncon, 1)), ("hess", m.meta.nvar)):
        r_, c_ = o.jac_structure() if kind == "jac" else o.hess_structure()
        vals = o.jac_coord(x) if kind == "jac" else o.hess_coord(x, y, 0.7)
        want = np.zeros(nrow * m.meta.nvar)
        np.add.at(want, (np.asarray(c_) - 1) * nrow + (np.asarray(r_) - 1), np.where(np.isfinite(vals), vals, 0.0))
        poison()
        cr, cc = cm.jac_structure() if kind == "jac" else cm.hess_structure()
        poison()
        cv = (cm.jac_coord(xd) if kind == "jac" else cm.hess_coord(xd, yd, 0.7)).cpu().numpy()
        n = cm.meta.nnzj if kind == "jac" else cm.meta.nnzh
        cr = np.asarray(cr.cpu() if hasattr(cr, "cpu") else cr)[:n]
        cc = np.asarray(cc.cpu() if hasattr(cc, "cpu") else cc)[:n]
        got = np.zeros_like(want)
        np.add.at(got, (cc - 1) * nrow + (cr - 1), np.where(np.isfinite(cv[:n]), cv[:n], 0.0))
        assert relerr(got, want) <= 1e-9, kind


@pytest.mark.parametrize("which", ["lv", "lv_chained", "rocket", "acopf"])
def test_benchmark_models_after_poison(libs, poison, which, monkeypatch):
    """the benchmark models at sizes that fill the chip several times over (many wavefronts per SIMD, every kernel variant of
    hess_coord!: plain, and the chained / staged one forced)"""
    from exahip import ExaModel, models
    import oracle
    if which == "lv_chained":
        monkeypatch.setenv("EXAHIP_HESS_VARIANT", "1")
    core = {"lv": lambda: models.luksan_vlcek_model(1_000_000), "lv_chained": lambda: models.luksan_vlcek_model(1_000_000),
            "rocket": lambda: models.rocket_model(100_000),
            "acopf": lambda: models.ac_power_model(models.synthetic_power_data(20_000, 32_000, 1_700, seed=0))}[which]()
    m = ExaModel(core)
    o = oracle.OracleModel(m.ir)
    x = (models.acopf_start(core) if which == "acopf" else m.meta.x0 + 0.1 * np.random.default_rng(41).uniform(-1, 1, m.meta.nvar))
    y = np.random.default_rng(42).standard_normal(m.meta.ncon)
    v = np.random.default_rng(43).standard_normal(m.meta.nvar)
    w = np.random.default_rng(44).standard_normal(m.meta.ncon)
    check_all(m, o, x, y, v, w, poison, 1e-10)


@pytest.mark.parametrize("variant", [0, 1])
def test_headline_size_after_poison(libs, poison, variant, monkeypatch):
    """BASELINE config 2 itself (Luksan-Vlcek N = 1e7, the 9e7-entry Hessian), by the plain kernel and by the chained / staged
    one, each right after the poisoning, against the oracle's full vector"""
    import torch
    from exahip import ExaModel, models
    import oracle
    monkeypatch.setenv("EXAHIP_HESS_VARIANT", str(variant))
    N = 10_000_000
    m = ExaModel(models.luksan_vlcek_model(N))
    assert m._L.exa_hess_variant(m.id) == variant
    o = oracle.OracleModel(m.ir, threads=16)
    x = m.meta.x0 + 0.1 * np.random.default_rng(51).uniform(-1, 1, N)
    y = np.random.default_rng(52).standard_normal(m.meta.ncon)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = torch.full((m.meta.nnzh,), float("nan"), dtype=torch.float64, device="cuda")
    poison()
    m.hess_coord(xd, yd, 0.5, out=out)
    torch.cuda.synchronize()
    assert relerr(out.cpu().numpy(), o.hess_coord(x, y, 0.5)) <= 1e-10
