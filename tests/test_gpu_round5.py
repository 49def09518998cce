"""-m gpu, round 5: the library's own use of locality for ORDER-FREE outputs (exa_set_locality: grad!, J'v, Hv by atomics run on a
locality-ordered copy of every table-driven pattern's columns; VERDICT r4 item 4 — the reference sorts its scatter lists at build,
ext/ExaModelsKernelAbstractions.jl:44-53, 79-101), everything with a slot / row order untouched.  Built, correct — and measured SLOWER
on the random ACOPF graph (profiles/r5_locality_ab.txt), so it is opt-in: these tests switch it on."""
import numpy as np
import pytest

from conftest import has_gpu
from zoo import point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]
RTOL = 1e-10


def relerr(a, ref):
    a, ref = np.asarray(a), np.asarray(ref)
    scale = np.maximum(np.abs(ref), 1e-3 * max(1.0, np.max(np.abs(ref))))
    return float(np.max(np.abs(a - ref) / scale))


@pytest.fixture(scope="module")
def acopf(libs):
    """6 000 buses / 9 000 branches with random ends in random order: the worst case for line-granular gathers, and above the 4 096 rows a
    table needs before the library builds a permutation for it."""
    from exahip import ExaModel, models
    import oracle
    core = models.ac_power_model(models.synthetic_power_data(6_000, 9_000, 600, seed=3))
    m = ExaModel(core)
    return m, oracle.OracleModel(m.ir, threads=8), models.acopf_start(core)


def test_locality_copies_are_built_for_table_driven_patterns_only(acopf, libs):
    from exahip import ExaModel, models
    m, _, _ = acopf
    assert m.set_locality(-1) == 0                       # off by default: measured slower where the patterns also read by row (exahip.h)
    assert m.set_locality(1) >= 1                        # the branch table (9 000 rows, random order) has a permutation to install
    src = m.kernel_source()
    assert "? ((const long*)P[" in src                   # the order-free kernels read the original row through the permutation column
    lv = ExaModel(models.luksan_vlcek_model(10_000))
    assert lv.set_locality(-1) == 0 and "? ((const long*)P[" not in lv.kernel_source()       # range-iterated patterns: nothing to permute


def test_order_free_callbacks_on_the_locality_ordered_copies(acopf):
    """grad!, J'v, Hv with the copies in: against the oracle (1e-10) and against the same kernels on the caller's order (1e-12: the
    same terms, added in another order by atomics either way); with the copies out the results are the round-4 ones."""
    m, o, x0 = acopf
    x, y, sigma = point(x0, m.meta.ncon, seed=9)
    v = np.random.default_rng(1).standard_normal(m.meta.nvar)
    w = np.random.default_rng(2).standard_normal(m.meta.ncon)
    m.set_product_mode(0, 0)                # atomics: the implementation that uses the copies
    ref = {"grad": o.grad(x), "jtprod": o.jtprod(x, w), "hprod": o.hprod(x, y, v, sigma)}
    got = {}
    for on in (1, 0, 1):
        assert (m.set_locality(on) >= 1) == bool(on)
        got[on] = {"grad": m.grad(x), "jtprod": m.jtprod(x, w), "hprod": m.hprod(x, y, v, sigma)}
        for k in ref:
            assert relerr(got[on][k], ref[k]) <= RTOL, (on, k)
    for k in ref:
        assert relerr(got[1][k], got[0][k]) <= 1e-12, k
    # what has a slot / row order does not move at all
    m.set_locality(1)
    a = (m.cons(x), m.jac_coord(x), m.hess_coord(x, y, sigma), m.jprod(x, v))
    m.set_locality(0)
    b = (m.cons(x), m.jac_coord(x), m.hess_coord(x, y, sigma), m.jprod(x, v))
    m.set_locality(1)
    for p, q in zip(a, b):
        assert np.array_equal(p, q)


def test_sharded_partial_sums_on_the_locality_ordered_copies(acopf):
    """Three ranks replayed on one GPU: each adds the contributions of ITS stretch of the permuted table; the partial sums add up to
    the whole (which rows a rank holds differs from the COO's shard of the caller's order — nothing depends on that)."""
    import torch
    m, o, x0 = acopf
    x, y, sigma = point(x0, m.meta.ncon, seed=10)
    v = np.random.default_rng(3).standard_normal(m.meta.nvar)
    w = np.random.default_rng(4).standard_normal(m.meta.ncon)
    dev = torch.device("cuda:0")
    xd, yd, vd, wd = (torch.from_numpy(a).to(dev) for a in (x, y, v, w))
    m.set_product_mode(0, 0)
    m.set_locality(1)
    G = 3
    jtv, hv, g = np.zeros(m.meta.nvar), np.zeros(m.meta.nvar), np.zeros(m.meta.nvar)
    try:
        for r in range(G):
            m.set_shard(r, G)
            assert m.set_locality(-1) >= 1
            jtv += m.jtprod(xd, wd).cpu().numpy()
            hv += m.hprod(xd, yd, vd, sigma).cpu().numpy()
            g += m.grad(xd).cpu().numpy()
    finally:
        m.set_shard(0, 1)
    assert relerr(jtv, o.jtprod(x, w)) <= RTOL
    assert relerr(hv, o.hprod(x, y, v, sigma)) <= RTOL
    assert relerr(g, o.grad(x)) <= RTOL


def test_locality_copies_under_register_poison(acopf, tmp_path):
    from poison import make_poison
    m, o, x0 = acopf
    x, y, sigma = point(x0, m.meta.ncon, seed=11)
    v = np.random.default_rng(5).standard_normal(m.meta.nvar)
    w = np.random.default_rng(6).standard_normal(m.meta.ncon)
    poison = make_poison(str(tmp_path))
    m.set_product_mode(0, 0)
    m.set_locality(1)
    for _ in range(2):
        poison()
        assert relerr(m.jtprod(x, w), o.jtprod(x, w)) <= RTOL
        poison()
        assert relerr(m.hprod(x, y, v, sigma), o.hprod(x, y, v, sigma)) <= RTOL
        poison()
        assert relerr(m.grad(x), o.grad(x)) <= RTOL


@pytest.mark.parametrize("seed", [0, 3])
def test_locality_copies_on_random_table_models_with_content_aliased_columns(libs, seed):
    """Twelve random patterns over ONE table of 5 000 rows (tests/randexpr.py): columns are aliased BY CONTENT, so a pattern may alias
    columns of several other patterns — the case that crashed the first version (a permuted word taken from a pattern that had none).
    Copies in: grad!, J'v, Hv against the oracle; copies out: the same numbers to 1e-12."""
    import randexpr
    from exahip import ExaModel
    import oracle
    saved = randexpr.NPTS
    randexpr.NPTS = 5000
    try:
        m = ExaModel(randexpr.build_model(seed, npat=12, depth=3).to_ir())
    finally:
        randexpr.NPTS = saved
    o = oracle.OracleModel(m.ir, threads=8)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=seed)
    v = np.random.default_rng(7).standard_normal(m.meta.nvar)
    w = np.random.default_rng(8).standard_normal(m.meta.ncon)
    m.set_product_mode(0, 0)
    out = {}
    for on in (1, 0):
        m.set_locality(on)
        out[on] = (m.grad(x), m.jtprod(x, w), m.hprod(x, y, v, sigma))
        for got, ref in zip(out[on], (o.grad(x), o.jtprod(x, w), o.hprod(x, y, v, sigma))):
            assert relerr(got, ref) <= 1e-9          # (deep random trees: the sweeps' tolerance)
    for a, b in zip(out[1], out[0]):
        assert relerr(a, b) <= 1e-11


# ---- exa_eval_all in one launch for an injective data-indexed objective (the ACOPF half) --------------------------------
def test_eval_all_stores_an_injective_objective_gradient_inside_the_sweep(acopf):
    """ACOPF's generator costs reach pg through the generator table: every variable at most once (proven on the data at model build) — the
    sweep's objective tiles store their first partials and zero tiles of the same launch cover the rest: g fully overwritten (it starts as NaN),
    bitwise equal to grad! (each entry is ONE term either way), no zero-fill launch.  Sharded: back to atomics on a zeroed vector."""
    import torch
    m, o, x0 = acopf
    x, y, sigma = point(x0, m.meta.ncon, seed=12)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    assert m.eval_all_mode() == 2
    g = torch.full((m.meta.nvar,), float("nan"), dtype=torch.float64, device=dev)
    f, g, c, j, h = m.eval_all(xd, yd, sigma, g=g)
    torch.cuda.synchronize()
    assert relerr(g.cpu().numpy(), o.grad(x)) <= RTOL and torch.equal(g, m.grad(xd))
    # grad! itself takes the same one-launch form: into a NaN-filled vector (every entry written: stores + zero tiles), twice
    for _ in range(2):
        g2 = torch.full((m.meta.nvar,), float("nan"), dtype=torch.float64, device=dev)
        m.grad(xd, out=g2)
        torch.cuda.synchronize()
        assert torch.equal(g2, g)
    assert relerr(h.cpu().numpy(), o.hess_coord(x, y, sigma)) <= RTOL and relerr(c.cpu().numpy(), o.cons(x)) <= RTOL
    acc = np.zeros(m.meta.nvar)
    try:
        for r in range(3):
            m.set_shard(r, 3)
            assert m.eval_all_mode() == 0
            gr = torch.full((m.meta.nvar,), float("nan"), dtype=torch.float64, device=dev)
            acc += m.eval_all(xd, yd, sigma, g=gr)[1].cpu().numpy()
    finally:
        m.set_shard(0, 1)
    assert relerr(acc, o.grad(x)) <= RTOL


def test_eval_all_keeps_the_atomics_where_the_objective_scatter_is_not_injective(libs):
    """Random table models: several objective patterns name the same variables through data columns — the build finds the collision and
    exa_eval_all keeps the zero-fill + atomics; LV: the gathered gradient's tiles (mode 1)."""
    import torch
    import randexpr
    from exahip import ExaModel, models
    import oracle
    m = ExaModel(randexpr.build_model(5, npat=8, depth=3).to_ir())
    assert m.eval_all_mode() in (0, 3)
    o = oracle.OracleModel(m.ir)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=5)
    dev = torch.device("cuda:0")
    g = torch.full((m.meta.nvar,), float("nan"), dtype=torch.float64, device=dev)
    g = m.eval_all(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), sigma, g=g)[1]
    assert relerr(g.cpu().numpy(), o.grad(x)) <= 1e-9
    assert ExaModel(models.luksan_vlcek_model(5000)).eval_all_mode() == 1


def test_cons_of_a_model_without_augmentation_by_fused_groups(libs, monkeypatch):
    """cons_nln! of the rocket (no augmentation terms) runs exa_cons1 — its three equally long dynamics patterns evaluated by one thread — by
    default; EXAHIP_CONS_FUSED=0 gives the per-pattern exa_cons.  Same rows, same bits, both equal to the oracle; sharded rows still complete."""
    import torch
    from exahip import ExaModel, models
    import oracle
    core = models.rocket_model(3000)
    m = ExaModel(core)
    o = oracle.OracleModel(m.ir)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=13)
    c1 = m.cons(x)
    monkeypatch.setenv("EXAHIP_CONS_FUSED", "0")
    m0 = ExaModel(models.rocket_model(3000))
    c0 = m0.cons(x)
    assert np.array_equal(c0, c1) and relerr(c1, o.cons(x)) <= RTOL
    dev = torch.device("cuda:0")
    xd = torch.from_numpy(x).to(dev)
    buf = torch.full((m.meta.ncon,), float("nan"), dtype=torch.float64, device=dev)
    try:
        for r in range(3):
            m.set_shard(r, 3)
            m.cons(xd, out=buf)
    finally:
        m.set_shard(0, 1)
    assert np.array_equal(buf.cpu().numpy(), c1)
