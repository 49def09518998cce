"""-m gpu: parity at BASELINE.json's full sizes (configs 2-4).

The oracle is an interpreter, but with OpenMP over data points it still finishes each of these in seconds on the GPU
box's host cores, so the whole COO vector is compared — plus the size-independent properties the domain offers:
linearity of the Lagrangian Hessian in (obj_weight, y), shard tiling, and window locality of Luksan-Vlcek.
"""
import os

import numpy as np
import pytest

from conftest import coo_slot, has_gpu, parity, parity_cons
from zoo import point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]
RTOL = 1e-10
THREADS = max(1, min(32, (os.cpu_count() or 2) // 2))


def maxrel(a, ref):
    scale = np.maximum(np.abs(ref), 1e-3 * max(1.0, float(np.max(np.abs(ref)))))
    return float(np.max(np.abs(a - ref) / scale))


def full_compare(core, seed, check_products=False, label="model", strict=RTOL):
    import torch
    from exahip import ExaModel
    import oracle
    m = ExaModel(core)
    o = oracle.OracleModel(m.ir, threads=THREADS)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=seed)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    h = m.hess_coord(xd, yd, sigma)
    j = m.jac_coord(xd)
    c = m.cons(xd)
    g = m.grad(xd)
    f = m.obj(xd)
    torch.cuda.synchronize()
    # floored and strict (component-wise) relative error, conftest.parity: `strict` = the bound asserted on the latter
    parity(label, "hess", h.cpu().numpy(), o.hess_coord(x, y, sigma), RTOL, strict, where=lambda k: coo_slot(m, True, k))
    parity(label, "jac", j.cpu().numpy(), o.jac_coord(x), RTOL, strict, where=lambda k: coo_slot(m, False, k))
    parity_cons(label, c.cpu().numpy(), o, x, RTOL)
    parity(label, "grad", g.cpu().numpy(), o.grad(x), RTOL, strict)
    fo = o.obj(x)
    assert abs(f - fo) <= RTOL * max(1.0, abs(fo))
    # linearity in (sigma, y): H(x, a*y, a*sigma) = a*H and additivity
    y2 = torch.from_numpy(np.random.default_rng(11).standard_normal(m.meta.ncon)).to(dev)
    h2 = m.hess_coord(xd, y2, 1.25)
    hs = m.hess_coord(xd, yd + y2, sigma + 1.25)
    torch.cuda.synchronize()
    num = (hs - (h + h2)).abs().max().item()
    den = max(1.0, hs.abs().max().item())
    assert num / den <= 1e-12
    # fused sweep == the separate callbacks (different kernels, same arithmetic up to FMA contraction)
    ff, cf, jf, hf = m.eval_fused(xd, yd, sigma)
    torch.cuda.synchronize()
    for a, b in ((cf, c), (jf, j), (hf, h)):
        d = (a - b).abs().max().item() if a.numel() else 0.0
        assert d <= 1e-11 * max(1.0, b.abs().max().item() if b.numel() else 1.0)
    assert abs(ff.item() - f) <= 1e-11 * max(1.0, abs(f))
    if not check_products:
        return m, o, (x, y, sigma), (xd, yd)
    # products at full size: H v and J' w against the COO of the same library, assembled on the device
    v = torch.from_numpy(np.random.default_rng(3).standard_normal(m.meta.nvar)).to(dev)
    w = torch.from_numpy(np.random.default_rng(4).standard_normal(m.meta.ncon)).to(dev)
    rows = torch.empty(m.meta.nnzh, dtype=torch.int64, device=dev)
    cols = torch.empty(m.meta.nnzh, dtype=torch.int64, device=dev)
    m.hess_structure(rows, cols)
    hv = m.hprod(xd, yd, v, sigma)
    ref = torch.zeros(m.meta.nvar, dtype=torch.float64, device=dev)
    ref.index_add_(0, rows - 1, h * v[cols - 1])
    off = rows != cols
    ref.index_add_(0, cols[off] - 1, h[off] * v[rows[off] - 1])
    assert (hv - ref).abs().max().item() <= 1e-9 * max(1.0, ref.abs().max().item())
    del rows, cols, ref
    jr = torch.empty(m.meta.nnzj, dtype=torch.int64, device=dev)
    jc = torch.empty(m.meta.nnzj, dtype=torch.int64, device=dev)
    m.jac_structure(jr, jc)
    jtv = m.jtprod(xd, w)
    ref = torch.zeros(m.meta.nvar, dtype=torch.float64, device=dev)
    ref.index_add_(0, jc - 1, j * w[jr - 1])
    assert (jtv - ref).abs().max().item() <= 1e-9 * max(1.0, ref.abs().max().item())
    jv = m.jprod(xd, v)
    ref = torch.zeros(m.meta.ncon, dtype=torch.float64, device=dev)
    ref.index_add_(0, jr - 1, j * v[jc - 1])
    assert (jv - ref).abs().max().item() <= 1e-9 * max(1.0, ref.abs().max().item())
    return m, o, (x, y, sigma), (xd, yd)


def test_config2_lv_1e7_full_vector(libs):
    import torch
    from exahip import ExaModel, models
    import oracle
    N = 10_000_000
    # component-wise: every one of the 9e7 Hessian entries within 1e-12 of the oracle's (a few 1e-16 on LV)
    m, o, (x, y, sigma), (xd, yd) = full_compare(models.luksan_vlcek_model(N), seed=0, label="config2 LV 1e7", strict=1e-12)
    assert m.meta.nnzh == 9 * N - 15
    # structure at full size: int32 rows/cols, lower triangle, equal to the oracle's
    rows = torch.empty(m.meta.nnzh, dtype=torch.int32, device=xd.device)
    cols = torch.empty(m.meta.nnzh, dtype=torch.int32, device=xd.device)
    m.hess_structure(rows, cols)
    torch.cuda.synchronize()
    orr, oc = o.hess_structure()
    assert torch.all(rows >= cols).item()
    assert np.array_equal(rows.cpu().numpy(), orr.astype(np.int32)) and np.array_equal(cols.cpu().numpy(), oc.astype(np.int32))
    # window locality: the slots of data points [a, a+n) equal those of a small model built on x[a : a+n+2]
    a, n = 4_321_000, 500
    small = ExaModel(models.luksan_vlcek_model(n + 2))
    hw = small.hess_coord(x[a:a + n + 2], y[a:a + n], sigma)
    big = m.hess_coord(xd, yd, sigma).cpu().numpy()
    np.testing.assert_allclose(big[6 * a:6 * (a + n)], hw[:6 * n], rtol=1e-13, atol=0)


def test_config3_rocket_1e6(libs):
    from exahip import models
    m, o, _, _ = full_compare(models.rocket_model(1_000_000), seed=1, label="config3 rocket 1e6", strict=RTOL)
    assert m.meta.nvar == 4 * 1_000_001 + 1 and m.meta.ncon == 3 * 1_000_000 + 4


def test_config4_acopf_78k_synthetic(libs):
    from exahip import models
    nbus, nbr, ngen = 78_484, 126_015, 6_800
    data = models.synthetic_power_data(nbus, nbr, ngen, seed=0)
    m, o, _, _ = full_compare(models.ac_power_model(data), seed=2, check_products=True, label="config4 ACOPF 78k")
    assert m.meta.nnzh == ngen + 44 * nbr + 2 * nbus


def test_config5_size_lv_1e8_on_one_gpu(libs):
    """BASELINE.json configs[4] size on ONE GPU: LV N=1e8 (nnzh = 9e8 < 2^31, 7.2 GB of COO).  64-bit indexing, int32
    structure, and window parity against small models (window locality of Luksan-Vlcek: the slots of points [a, a+n)
    depend only on x[a : a+n+2], y[a : a+n])."""
    import torch
    import oracle
    from exahip import ExaModel, models
    N = 100_000_000
    m = ExaModel(models.luksan_vlcek_model(N))
    dev = torch.device("cuda:0")
    x = m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)
    y = np.random.default_rng(1).standard_normal(N - 2)
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    h = torch.empty(m.meta.nnzh, dtype=torch.float64, device=dev)
    m.hess_coord(xd, yd, 0.5, out=h)
    torch.cuda.synchronize()
    assert m.meta.nnzh == 9 * N - 15
    ob = 6 * (N - 2)
    for a in (0, 12_345_678, 49_999_000, N - 1000 - 2):
        n = 1000
        o = oracle.OracleModel(models.luksan_vlcek_model(n + 2).to_ir())
        ref = o.hess_coord(x[a:a + n + 2], y[a:a + n], 0.5)
        got = h[6 * a:6 * (a + n)].cpu().numpy()
        assert maxrel(got, ref[:6 * n]) <= RTOL
        got_o = h[ob + 3 * a: ob + 3 * (a + n)].cpu().numpy()     # objective block: points I = a .. a+n (i = I+2)
        assert maxrel(got_o, ref[6 * n: 6 * n + 3 * n]) <= RTOL
    rows = torch.empty(m.meta.nnzh, dtype=torch.int32, device=dev)
    cols = torch.empty(m.meta.nnzh, dtype=torch.int32, device=dev)
    m.hess_structure(rows, cols)
    torch.cuda.synchronize()
    assert bool(torch.all(rows >= cols)) and int(rows.max()) == N and int(cols.min()) == 1


@pytest.mark.parametrize("which", ["lv", "rocket", "acopf"])
def test_compressed_full_size_equals_scattered_uncompressed(libs, which):
    """Compressed COO (CompressedNLPModel) at full size, all on the device: every compressed entry equals the sum of the
    uncompressed slots with its coordinates (torch index_add_ through a searchsorted on the (col,row) key), the entries
    are strictly (col,row)-ascending, and the CSC colptr brackets them.  LV and the rocket take the windowed sweep, the
    ACOPF the permuted store (merged slots for the Hessian)."""
    import torch
    from exahip import CompressedExaModel, ExaModel, models
    core = {"lv": lambda: models.luksan_vlcek_model(10_000_000), "rocket": lambda: models.rocket_model(1_000_000),
            "acopf": lambda: models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0))}[which]()
    m = ExaModel(core)
    cm = CompressedExaModel(m)
    # (ACOPF: bus variables are reached through table columns — no windows; the sweep stores the merged slots of every
    # fused group at their sorted positions instead of gathering 5.7 M values at random)
    assert cm.path("hess")[0] == ("scatter" if which == "acopf" else "windowed"), cm.path("hess")
    if which == "acopf":
        assert "merged slots" in cm.path("hess")[1] and cm.path("jac")[0] == "scatter"
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=5)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    for kind in ("jac", "hess"):
        nrow = max(m.meta.ncon if kind == "jac" else m.meta.nvar, 1)
        nnz = m.meta.nnzj if kind == "jac" else m.meta.nnzh
        r = torch.empty(nnz, dtype=torch.int64, device=dev)
        c = torch.empty(nnz, dtype=torch.int64, device=dev)
        (m.jac_structure if kind == "jac" else m.hess_structure)(r, c)
        v = m.jac_coord(xd) if kind == "jac" else m.hess_coord(xd, yd, sigma)
        cr, cc = cm.jac_structure() if kind == "jac" else cm.hess_structure()
        cv = cm.jac_coord(xd) if kind == "jac" else cm.hess_coord(xd, yd, sigma)
        ckey = (cc - 1) * nrow + (cr - 1)
        assert bool(torch.all(ckey[1:] > ckey[:-1]))                       # sorted, no duplicates
        pos = torch.searchsorted(ckey, (c - 1) * nrow + (r - 1))
        assert bool(torch.all(ckey[pos] == (c - 1) * nrow + (r - 1)))      # every uncompressed slot has its entry
        # entries collecting a million slots (the rocket's step length) are summed separately: FP64 index_add_ on one
        # address is a CAS storm (175 s for this test)
        cnt = torch.bincount(pos, minlength=cv.numel()).to(torch.float64)
        heavy = torch.nonzero(cnt > 10_000).flatten()
        light = ~(cnt > 10_000)[pos]
        ref = torch.zeros_like(cv).index_add_(0, pos[light], v[light])
        mag = torch.zeros_like(cv).index_add_(0, pos[light], v[light].abs())
        assert heavy.numel() <= 8
        for e in heavy.tolist():
            sel = pos == e
            ref[e] = v[sel].sum()
            mag[e] = v[sel].abs().sum()
        # the compressed sweep is its own generated kernel (other common subexpressions, other FMA contraction): a slot
        # whose value is what cancellation inside its expression leaves (5e-5 from terms of order 1 in the ACOPF flow
        # rows) differs by rounding of the TERMS, hence the absolute part scaled by the largest value
        bound = (4e-16 * cnt.sqrt() + 1e-13) * mag + 1e-15 * float(v.abs().max()) + 1e-300
        assert bool(torch.all((cv - ref).abs() <= bound)), float(((cv - ref).abs() / bound).max())
        colptr, rowval = cm.csc(kind)
        assert int(colptr[0]) == 1 and int(colptr[-1]) == cv.numel() + 1 and bool(torch.all(colptr[1:] >= colptr[:-1]))
        assert torch.equal(rowval, cr)
        # column of entry k is the j with colptr[j] <= k+1 < colptr[j+1]
        k = torch.arange(0, cv.numel(), max(1, cv.numel() // 100_000), device=dev)
        assert bool(torch.all(torch.searchsorted(colptr, k + 1, right=True) == cc[k]))
        del r, c, v, pos, ref, mag, cnt, light
