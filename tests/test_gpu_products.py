"""-m gpu: matrix-free products jprod_nln! / jtprod_nln! / hprod! against the oracle, which applies
the reference's leaf actions literally (src/jacobian.jl:41-68, src/hessian.jl:291-315, 566-579), and against the
assembled COO matrices (NLPTest.jl:78-80 checks the same three products between backends)."""
import numpy as np
import pytest

from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]
RTOL = 1e-10


def relerr(a, ref):
    a, ref = np.asarray(a), np.asarray(ref)
    if ref.size == 0:
        return 0.0
    scale = np.maximum(np.abs(ref), 1e-3 * max(1.0, np.max(np.abs(ref))))
    return float(np.max(np.abs(a - ref) / scale))


@pytest.mark.parametrize("name", list(ZOO))
def test_products_match_oracle_and_coo(libs, name):
    from exahip import ExaModel
    import oracle
    m = ExaModel(ZOO[name]())
    o = oracle.OracleModel(m.ir)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=4)
    v = np.random.default_rng(5).standard_normal(m.meta.nvar)
    w = np.random.default_rng(6).standard_normal(m.meta.ncon)
    assert relerr(m.jprod(x, v), o.jprod(x, v)) <= RTOL
    assert relerr(m.jtprod(x, w), o.jtprod(x, w)) <= RTOL
    assert relerr(m.hprod(x, y, v, sigma), o.hprod(x, y, v, sigma)) <= RTOL
    # consistency with the COO the same library returns
    jr, jc = m.jac_structure()
    J = np.zeros((m.meta.ncon, m.meta.nvar))
    np.add.at(J, (jr - 1, jc - 1), m.jac_coord(x))
    hr, hc = m.hess_structure()
    H = np.zeros((m.meta.nvar, m.meta.nvar))
    np.add.at(H, (hr - 1, hc - 1), m.hess_coord(x, y, sigma))
    H = H + np.tril(H, -1).T
    assert relerr(m.jprod(x, v), J @ v) <= 1e-9
    assert relerr(m.jtprod(x, w), J.T @ w) <= 1e-9
    assert relerr(m.hprod(x, y, v, sigma), H @ v) <= 1e-9


def test_products_device_pointers_and_sharding(libs):
    import torch
    from exahip import ExaModel
    import oracle
    m = ExaModel(ZOO["acopf30"]())
    o = oracle.OracleModel(m.ir)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=8)
    v = np.random.default_rng(1).standard_normal(m.meta.nvar)
    w = np.random.default_rng(2).standard_normal(m.meta.ncon)
    dev = torch.device("cuda:0")
    xd, yd, vd, wd = (torch.from_numpy(a).to(dev) for a in (x, y, v, w))
    G = 3
    jtv, hv = np.zeros(m.meta.nvar), np.zeros(m.meta.nvar)
    # J v: a rank writes the rows of its own data points, complete (the row's owner adds the augmentation terms itself),
    # and nothing else — all ranks into one buffer; J'v / Hv of this data-indexed model: partial sums, added up
    assert m.shard_layout("jprod") == "pieces" and m.shard_layout("jtprod") == "partial" and m.shard_layout("hprod") == "partial"
    jvd = torch.full((m.meta.ncon,), float("nan"), dtype=torch.float64, device=dev)
    try:
        for r in range(G):
            m.set_shard(r, G)
            m.jprod(xd, vd, out=jvd)
            jtv += m.jtprod(xd, wd).cpu().numpy()
            hv += m.hprod(xd, yd, vd, sigma).cpu().numpy()
    finally:
        m.set_shard(0, 1)
    jv = jvd.cpu().numpy()
    assert relerr(jv, o.jprod(x, v)) <= RTOL
    assert relerr(jtv, o.jtprod(x, w)) <= RTOL
    assert relerr(hv, o.hprod(x, y, v, sigma)) <= RTOL


def test_fused_sweep_many_wavefronts(libs):
    """Regression: the fused function flushes TWO tiles of different shape (J then H) per wavefront; their LDS regions
    must be private to the wavefront.  Needs many concurrently resident wavefronts to show."""
    import torch
    from exahip import ExaModel, models
    N = 300_000
    m = ExaModel(models.luksan_vlcek_model(N))
    dev = torch.device("cuda:0")
    x = torch.from_numpy(m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)).to(dev)
    y = torch.from_numpy(np.random.default_rng(1).standard_normal(N - 2)).to(dev)
    c, j, h = m.cons(x), m.jac_coord(x), m.hess_coord(x, y, 0.5)
    for _ in range(3):
        f, cf, jf, hf = m.eval_fused(x, y, 0.5)
        torch.cuda.synchronize()
        assert torch.equal(cf, c) and torch.equal(jf, j) and torch.equal(hf, h)


@pytest.mark.parametrize("name", list(ZOO))
def test_fused_sweep_equals_separate_callbacks(libs, name):
    """exa_eval_fused: obj, cons, jac_coord, hess_coord from one sweep == the oracle's separate results;
    outputs are poisoned first (fully overwritten)."""
    import torch
    from exahip import ExaModel
    import oracle
    m = ExaModel(ZOO[name]())
    o = oracle.OracleModel(m.ir)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=21)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    nan = float("nan")
    c = torch.full((m.meta.ncon,), nan, dtype=torch.float64, device=dev)
    j = torch.full((m.meta.nnzj,), nan, dtype=torch.float64, device=dev)
    h = torch.full((m.meta.nnzh,), nan, dtype=torch.float64, device=dev)
    f, c, j, h = m.eval_fused(xd, yd, sigma, c=c, jac=j, hess=h)
    torch.cuda.synchronize()
    fo = o.obj(x)
    assert abs(f.item() - fo) <= RTOL * max(1.0, abs(fo))
    assert relerr(c.cpu().numpy(), o.cons(x)) <= RTOL
    assert relerr(j.cpu().numpy(), o.jac_coord(x)) <= RTOL
    assert relerr(h.cpu().numpy(), o.hess_coord(x, y, sigma)) <= RTOL


@pytest.mark.parametrize("name", list(ZOO))
def test_eval_all_equals_the_five_separate_callbacks(libs, name):
    """exa_eval_all (the evaluation set of NLPModelsIpoptLite.jl:28-40): obj, grad!, cons, jac_coord, hess_coord
    from one sweep (+ the gathered gradient kernel) == the separate callbacks — bitwise, except the part of g that objective
    patterns add atomically inside the sweep — and == the oracle.  Outputs are poisoned first: fully overwritten."""
    import torch
    from exahip import ExaModel
    import oracle
    m = ExaModel(ZOO[name]())
    o = oracle.OracleModel(m.ir)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=23)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    nan = float("nan")
    bufs = [torch.full((max(1, k),), nan, dtype=torch.float64, device=dev) for k in (m.meta.nvar, m.meta.ncon, m.meta.nnzj, m.meta.nnzh)]
    f, g, c, j, h = m.eval_all(xd, yd, sigma, g=bufs[0], c=bufs[1], jac=bufs[2], hess=bufs[3])
    torch.cuda.synchronize()
    fo = o.obj(x)
    assert abs(f.item() - fo) <= RTOL * max(1.0, abs(fo))
    for got, ref, sep in ((g, o.grad(x), m.grad(xd)), (c, o.cons(x), m.cons(xd)), (j, o.jac_coord(x), m.jac_coord(xd)),
                          (h, o.hess_coord(x, y, sigma), m.hess_coord(xd, yd, sigma))):
        n = ref.size
        assert relerr(got.cpu().numpy()[:n], ref) <= RTOL
        if got is g and m.shard_layout("grad") == "pieces":           # gathered gradient: the same function, the same bits
            assert torch.equal(got[:n], sep[:n])
        else:       # (atomic adds in any order; the sweep and the separate kernels contract their FMAs differently)
            assert relerr(got.cpu().numpy()[:n], sep.cpu().numpy()[:n]) <= 1e-12
    # deterministic gradient requested: eval_all takes the sorted gather for it
    m.set_grad_mode(1)
    f2, g2, *_ = m.eval_all(xd, yd, sigma)
    assert relerr(g2.cpu().numpy(), o.grad(x)) <= RTOL and torch.equal(g2, m.grad(xd))
    m.set_grad_mode(-1)


@pytest.mark.parametrize("name", ["lv20", "acopf30", "rocket50", "mixed", "cops_elec"])
@pytest.mark.parametrize("mode", [0, 1])
def test_both_product_implementations(libs, name, mode):
    """jtprod/hprod: atomics-in-the-sweep (0) and COO + sorted gather (1, the reference's prod helper) agree with the
    oracle; the default picks one by measurement."""
    from exahip import ExaModel
    import oracle
    m = ExaModel(ZOO[name]())
    o = oracle.OracleModel(m.ir)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=31)
    v = np.random.default_rng(5).standard_normal(m.meta.nvar)
    w = np.random.default_rng(6).standard_normal(m.meta.ncon)
    m.set_product_mode(mode, mode)
    assert relerr(m.jtprod(x, w), o.jtprod(x, w)) <= RTOL
    assert relerr(m.hprod(x, y, v, sigma), o.hprod(x, y, v, sigma)) <= RTOL
    assert m.product_mode() == (mode, mode)
    m.set_product_mode(-1, -1)
    assert relerr(m.jtprod(x, w), o.jtprod(x, w)) <= RTOL
    assert relerr(m.hprod(x, y, v, sigma), o.hprod(x, y, v, sigma)) <= RTOL
    assert all(k in (0, 1, 2) for k in m.product_mode())


@pytest.mark.parametrize("name", sorted(ZOO))
def test_grad_by_sorted_gather_is_the_reference_scheme_and_deterministic(libs, name):
    """exa_set_grad_mode(1): gradient COO (ExaCore.nnzg slots) + (variable, slot) lists sorted once, every variable's slots
    added in slot order (KA ext :310-336).  Same values as the oracle and as the default path, bit-identical run to run."""
    import torch
    import oracle
    from exahip import ExaModel
    m = ExaModel(ZOO[name]())
    o = oracle.OracleModel(m.ir)
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=23)
    ref = o.grad(x)
    g0 = m.grad(x)
    m.set_grad_mode(1)
    assert m.grad_mode() == 1
    g1 = m.grad(x)
    scale = 1.0 + np.abs(ref).max()
    assert np.all(np.abs(g1 - ref) <= 1e-10 * np.abs(ref) + 1e-13 * scale)
    assert np.all(np.abs(g1 - g0) <= 1e-12 * np.abs(ref) + 1e-14 * scale)
    xd = torch.from_numpy(x).cuda()
    a, b = m.grad(xd).clone(), m.grad(xd).clone()
    assert torch.equal(a, b)
    m.set_grad_mode(0)


def test_grad_tuning_picks_the_sorted_gather_for_hot_targets(libs, tmp_path, monkeypatch):
    """Sixteen shared variables behind a table column in random order: the atomics of 64 lanes land on one cache line
    (17 ms at 1e7 points).  exa_tune (bit 2) measures both implementations, installs and persists the sorted gather; a
    second process-lifetime model of the same module starts with it."""
    import time
    import torch
    from exahip import ExaCore, ExaModel, Table, rng
    from exahip.graph import sin
    import oracle
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))
    N, K = 400_000, 16

    def build():
        c = ExaCore()
        x = c.add_var(N + K, start=np.linspace(0.1, 1.0, N + K))
        tab = Table(i=np.arange(1, N + 1), k=N + 1 + np.random.default_rng(0).integers(0, K, N), w=np.linspace(0.5, 1.5, N))
        c.add_obj(lambda t: t.w * x[t.i] * sin(x[t.k]), tab)
        c.add_obj(lambda i: (x[i] - 1.0) ** 2, rng(1, N))
        return c
    m = ExaModel(build())
    xs = np.asarray(m.meta.x0) + 0.01
    xd = torch.from_numpy(xs).cuda()
    ref = oracle.OracleModel(m.ir).grad(xs)

    def timed(mm):
        mm.grad(xd); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            mm.grad(xd)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10
    assert m.grad_mode() == -1
    t_atomics = timed(m)
    m.tune(4, xd)
    assert m.grad_mode() == 1
    t_sorted = timed(m)
    np.testing.assert_allclose(m.grad(xd).cpu().numpy(), ref, rtol=1e-10, atol=1e-12)
    assert t_sorted < t_atomics
    # every shared variable collects 25 000 slots: summed chunk by chunk, the chunk sums folded in chunk order — same bits
    a, b = m.grad(xd).clone(), m.grad(xd).clone()
    assert torch.equal(a, b)
    m2 = ExaModel(build())
    m2.grad(xd)
    assert m2.grad_mode() == 1                       # the persisted decision
    np.testing.assert_allclose(m2.grad(xd).cpu().numpy(), ref, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("name", ["acopf30", "rocket50", "mixed", "lv1000"])
def test_deterministic_switch_makes_every_callback_bit_reproducible(libs, name):
    """exa_set_deterministic: grad! by sorted gather, jtprod and hprod by owner-computes windows where the model has them
    (fixed order of additions, no atomics), else by owner pull (data-indexed targets), else by sorted gather.  Ten evaluations of everything, interleaved: one bit
    pattern per output — and still the oracle's values."""
    import torch
    import oracle
    from exahip import ExaModel
    m = ExaModel(ZOO[name]())
    o = oracle.OracleModel(m.ir)
    m.set_deterministic(True)
    # windows where the model has them, else the owner pull (data-indexed targets, round 4), else the sorted gather
    pull = "owner pull available" in m.product_info("jtprod")[1]
    assert m.grad_mode() == 1 and m.product_mode() == ((2, 2) if name in ("rocket50", "lv1000") else (3, 3) if pull else (1, 1))
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=29)
    v = np.random.default_rng(5).standard_normal(m.meta.nvar)
    w = np.random.default_rng(6).standard_normal(max(m.meta.ncon, 1))[:m.meta.ncon]
    dev = torch.device("cuda:0")
    xd, yd, vd, wd = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, y, v, w))
    first = None
    for _ in range(10):
        f, c, j, h = m.eval_fused(xd, yd, s)
        outs = [m.grad(xd), m.jtprod(xd, wd), m.hprod(xd, yd, vd, s), m.cons(xd), m.jac_coord(xd), m.hess_coord(xd, yd, s), c, j, h, f,
                m.jprod(xd, vd)]
        torch.cuda.synchronize()
        outs = [t.clone() for t in outs]
        if first is None:
            first = outs
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, outs))
    np.testing.assert_allclose(first[0].cpu().numpy(), o.grad(x), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(first[1].cpu().numpy(), o.jtprod(x, w), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(first[2].cpu().numpy(), o.hprod(x, y, v, s), rtol=1e-10, atol=1e-11)
    m.set_deterministic(False)
    assert m.grad_mode() == -1 and m.product_mode() == (-1, -1)
