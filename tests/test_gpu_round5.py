"""-m gpu, round 5: exa_eval_all in one launch for injective data-indexed objectives, cons_nln! by fused groups.  (The locality-ordered table
copies this file used to test — built in round 5, measured slower on the random ACOPF graph, profiles/r5_locality_ab.txt — were removed in round 6.)"""
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


def test_the_locality_copies_are_gone(acopf, libs):
    """Round 5's locality-ordered table copies (exa_set_locality) lost their A/B (profiles/r5_locality_ab.txt) and were removed in round 6: no
    permutation word, no row indirection in any kernel, no second copy of a table's columns."""
    m, _, _ = acopf
    assert "? ((const long*)P[" not in m.kernel_source() and not hasattr(m._L, "exa_set_locality_checked")
    import ctypes
    with pytest.raises(AttributeError):
        ctypes.CDLL(m._L._name).exa_set_locality


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
