"""-m gpu: compressed (duplicate-summed) COO — the CompressedNLPModel wrapper of the reference (src/utils.jl:425-579;
test/UtilsTest/UtilsTest.jl uses it end to end).  Properties checked: no duplicate coordinates, (col,row)-sorted,
same matrix as the partially compressed COO, duplicates added in original slot order (bit-exact vs numpy's ordered
accumulation)."""
import numpy as np
import pytest

from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def reference_compress(rows, cols, vals, nrowdim):
    """utils.jl:476-478, 519-562 restated with numpy: stable sort on (col,row), runs summed in slot order."""
    key = (cols - 1) * nrowdim + (rows - 1)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.concatenate([[0], np.nonzero(ks[1:] != ks[:-1])[0] + 1, [len(ks)]]) if len(ks) else np.array([0])
    out_r, out_c, out_v = [], [], []
    for a, b in zip(starts[:-1], starts[1:]):
        s = 0.0
        for j in order[a:b]:
            s += vals[j]
        out_v.append(s)
        out_r.append(rows[order[a]])
        out_c.append(cols[order[a]])
    return np.array(out_r), np.array(out_c), np.array(out_v)


@pytest.mark.parametrize("window", [0, 1])
@pytest.mark.parametrize("name", ["lv20", "lv_split_20x2", "acopf30", "mixed", "conaug2d", "rocket50"])
def test_compressed_equals_reference_compression(libs, name, window, monkeypatch):
    """window = 0: the reference's scheme (uncompressed evaluation + sorted gather), bit-exact against its summation
    order.  window = 1: whatever exa_compress picks — for stencil models the windowed sweep, which adds the same terms
    in a different (fixed) order: equal within rounding of the terms' magnitudes, and bit-identical run to run."""
    import torch
    from exahip import CompressedExaModel, ExaModel
    monkeypatch.setenv("EXAHIP_CWINDOW", str(window))
    m = ExaModel(ZOO[name]())
    cm = CompressedExaModel(m)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=12)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    for which, nrowdim in (("jac", max(m.meta.ncon, 1)), ("hess", max(m.meta.nvar, 1))):
        if which == "jac":
            r, c = m.jac_structure()
            v = m.jac_coord(x)
            cr, cc = cm.jac_structure()
            cv = cm.jac_coord(xd)
        else:
            r, c = m.hess_structure()
            v = m.hess_coord(x, y, sigma)
            cr, cc = cm.hess_structure()
            cv = cm.hess_coord(xd, yd, sigma)
        torch.cuda.synchronize()
        cr, cc, cv = cr.cpu().numpy(), cc.cpu().numpy(), cv.cpu().numpy()
        er, ec, ev = reference_compress(r, c, v, nrowdim)
        assert np.array_equal(cr, er) and np.array_equal(cc, ec)
        assert len(set(zip(cr.tolist(), cc.tolist()))) == len(cr), "duplicates left"
        kind, why = cm.path(which)
        if window == 0:
            assert kind == "gather" and "disabled" in why
        if kind == "gather":
            np.testing.assert_array_equal(cv, ev)     # same additions in the same order -> bit-exact
        else:
            _, _, mag = reference_compress(r, c, np.abs(v), nrowdim)
            assert np.all(np.abs(cv - ev) <= 2e-14 * mag), (np.abs(cv - ev) / np.maximum(mag, 1e-300)).max()
            again = (cm.jac_coord(xd) if which == "jac" else cm.hess_coord(xd, yd, sigma)).cpu().numpy()
            np.testing.assert_array_equal(again, cv)
    assert cm.meta.nnzh <= m.meta.nnzh and cm.meta.nnzj <= m.meta.nnzj


def test_compressed_lv_counts(libs):
    """LV(N): the 9N-15 partially compressed Hessian slots collapse onto the 2N-1 entries of a symmetric tridiagonal
    lower triangle plus the (i+2,i+1)... pattern: distinct pairs are (i,i) N, (i+1,i) N-1 -> 2N-1."""
    from exahip import CompressedExaModel, ExaModel, models
    N = 1000
    m = ExaModel(models.luksan_vlcek_model(N))
    cm = CompressedExaModel(m)
    assert m.meta.nnzh == 9 * N - 15
    assert cm.meta.nnzh == 2 * N - 1
    assert cm.meta.nnzj == m.meta.nnzj == 3 * (N - 2)      # the LV Jacobian has no duplicates
    # a stencil model: both matrices take the windowed sweep
    assert cm.path("hess")[0] == "windowed" and cm.path("jac")[0] == "windowed", (cm.path("hess"), cm.path("jac"))


def test_compressed_mid_size_many_blocks(libs):
    """Radix sort / run-length / scan path with millions of entries (multi-block): LV N = 200 000."""
    import torch
    from exahip import CompressedExaModel, ExaModel, models
    N = 200_000
    m = ExaModel(models.luksan_vlcek_model(N))
    cm = CompressedExaModel(m)
    assert cm.meta.nnzh == 2 * N - 1
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=3)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    r, c = m.hess_structure()
    v = m.hess_coord(x, y, sigma)
    key = (c - 1) * N + (r - 1)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.concatenate([[0], np.nonzero(ks[1:] != ks[:-1])[0] + 1])
    ev = np.add.reduceat(v[order], starts)            # pairwise order may differ from sequential: compare to 1e-13
    cr, cc = cm.hess_structure()
    cv = cm.hess_coord(xd, yd, sigma)
    torch.cuda.synchronize()
    assert np.array_equal(cr.cpu().numpy(), r[order][starts]) and np.array_equal(cc.cpu().numpy(), c[order][starts])
    np.testing.assert_allclose(cv.cpu().numpy(), ev, rtol=1e-13, atol=1e-13)


def test_heavily_duplicated_entry(libs):
    """The rocket's step length is shared by every data point: ONE compressed Hessian entry collects thousands of
    contributions (was summed by a single thread: 0.9 s at nh = 1e6; now per-chunk partial sums + an ordered fold).
    Those entries are summed as a tree, so they match the slot-ordered sum to rounding, the rest bit for bit."""
    import time
    import torch
    from exahip import CompressedExaModel, ExaModel, models
    m = ExaModel(models.rocket_model(20000))
    cm = CompressedExaModel(m)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=4)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    r, c = m.hess_structure()
    v = m.hess_coord(x, y, sigma)
    cr, cc = (t.cpu().numpy() for t in cm.hess_structure())
    cv = cm.hess_coord(xd, yd, sigma).cpu().numpy()
    key = (c - 1) * m.meta.nvar + (r - 1)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.concatenate([[0], np.nonzero(ks[1:] != ks[:-1])[0] + 1, [len(ks)]])
    counts = np.diff(starts)
    assert counts.max() > 10000 and len(counts) == len(cv)
    want = np.add.reduceat(v[order], starts[:-1])
    assert np.array_equal(cr, r[order][starts[:-1]]) and np.array_equal(cc, c[order][starts[:-1]])
    # numpy's reduceat sums pairwise, the library sequentially (or as a tree for the long entries): compare relative to
    # the magnitude of what was added, entry by entry (bit-exactness of the sequential sums is test 1's business)
    mag = np.add.reduceat(np.abs(v[order]), starts[:-1])
    assert np.all(np.abs(cv - want) <= 1e-13 * np.maximum(mag, 1e-300) * np.sqrt(counts))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        cm.hess_coord(xd, yd, sigma)
    torch.cuda.synchronize()
    assert (time.perf_counter() - t0) / 5 < 0.05          # seconds; the single-thread sum took 18 ms per 20000 points


def _check_against_uncompressed(m, cm, seed, tol=1e-14):
    """compressed values == duplicate-summed uncompressed values, within rounding of the summed magnitudes; the path
    taken is returned"""
    import torch
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=seed)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    kinds = []
    for which, nrowdim in (("jac", max(m.meta.ncon, 1)), ("hess", max(m.meta.nvar, 1))):
        if which == "jac":
            r, c = m.jac_structure(); v = m.jac_coord(x); cv = cm.jac_coord(xd); cr, cc = cm.jac_structure()
        else:
            r, c = m.hess_structure(); v = m.hess_coord(x, y, sigma); cv = cm.hess_coord(xd, yd, sigma); cr, cc = cm.hess_structure()
        if len(v) == 0:
            continue
        key = (c - 1) * nrowdim + (r - 1)
        order = np.argsort(key, kind="stable")
        ks = key[order]
        starts = np.concatenate([[0], np.nonzero(ks[1:] != ks[:-1])[0] + 1])
        ev = np.add.reduceat(v[order], starts)
        mag = np.add.reduceat(np.abs(v[order]), starts)
        cnt = np.diff(np.concatenate([starts, [len(ks)]]))
        cv = cv.cpu().numpy()
        assert np.array_equal(cr.cpu().numpy(), r[order][starts]) and np.array_equal(cc.cpu().numpy(), c[order][starts])
        fin = np.isfinite(ev)
        assert np.array_equal(np.isfinite(cv), fin)
        bound = (2e-15 * np.sqrt(cnt) + tol) * mag
        err = np.abs(cv[fin] - ev[fin])
        assert np.all(err <= bound[fin]), (which, (err / np.maximum(bound[fin], 1e-300)).max())
        kinds.append(cm.path(which)[0])
    return kinds


@pytest.mark.parametrize("unit", [False, True])
@pytest.mark.parametrize("seed", range(8))
def test_windowed_random_range_models(libs, seed, unit):
    """Random range-iterated models (tests/randexpr.build_range_model): stepped ranges, index offsets, constant-index
    leaves (entries shared by every point), several patterns meeting in the same compressed entries — the shapes the
    windowed sweep classifies into passes.  Whatever path exa_compress picks must reproduce the uncompressed sums."""
    import randexpr
    from exahip import CompressedExaModel, ExaModel
    m = ExaModel(randexpr.build_range_model(seed, npts=1500, npat=6, depth=3, unit=unit).to_ir())
    cm = CompressedExaModel(m)
    kinds = _check_against_uncompressed(m, cm, seed)
    print(seed, unit, kinds, cm.path("jac"), cm.path("hess"))


def test_windowed_reversed_and_tiny_patterns(libs):
    """Index decreasing with the data point (negative stride in the compressed array), one- and two-point patterns,
    and a pattern whose every point is 'irregular' (all of it goes through exa_c*x)."""
    from exahip import CompressedExaModel, ExaCore, ExaModel
    from exahip.core import rng
    N = 700
    c = ExaCore()
    x = c.add_var(N + 2, start=np.linspace(0.5, 1.5, N + 2))
    c.add_obj(lambda i: (x[N + 1 - i] - x[N - i] ** 2) ** 2 * x[N + 2 - i], rng(1, N - 1))
    c.add_obj(lambda i: x[i] ** 3, rng(5, 5))
    c.add_obj(lambda i: x[i] * x[i + 1], rng(9, 10))
    g = c.add_con(lambda i: x[N + 1 - i] ** 2 - 1.0, rng(1, N))
    c.add_con_aug(g, lambda i: (i, x[N + 2 - i] ** 3), rng(1, N))
    m = ExaModel(c.to_ir())
    cm = CompressedExaModel(m)
    kinds = _check_against_uncompressed(m, cm, 5)
    assert kinds == ["windowed", "windowed"], (cm.path("jac"), cm.path("hess"))


@pytest.mark.parametrize("name,size", [("lv", 300_000), ("rocket", 60_000)])
def test_windowed_equals_gather_at_scale(libs, name, size, monkeypatch):
    """Same model compressed twice — windowed sweep and the reference's gather — many windows, multi-chunk passes,
    shared entries with tens of thousands of terms."""
    import torch
    from exahip import CompressedExaModel, ExaModel, models
    build = (lambda: models.luksan_vlcek_model(size)) if name == "lv" else (lambda: models.rocket_model(size))
    m1 = ExaModel(build()); c1 = CompressedExaModel(m1)
    monkeypatch.setenv("EXAHIP_CWINDOW", "0")
    m0 = ExaModel(build()); c0 = CompressedExaModel(m0)
    assert c1.path("hess")[0] == "windowed" and c1.path("jac")[0] == "windowed" and c0.path("hess")[0] == "gather"
    x, y, sigma = point(m1.meta.x0, m1.meta.ncon, seed=8)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    for a, b in ((c1.jac_coord(xd), c0.jac_coord(xd)), (c1.hess_coord(xd, yd, sigma), c0.hess_coord(xd, yd, sigma))):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        scale = np.abs(b).max()
        assert np.all(np.abs(a - b) <= 1e-12 * np.abs(b) + 1e-13 * scale), np.abs(a - b).max()
    for p, q in zip(c1.hess_structure(), c0.hess_structure()):
        assert torch.equal(p, q)


@pytest.mark.parametrize("name", ["lv20", "acopf30", "rocket50", "mixed"])
def test_compressed_csc_view(libs, name):
    """colptr / rowval of the compressed entries + the values of jac_coord / hess_coord = the CSC matrix (scipy
    csc_matrix built from them equals the duplicate-summed COO); empty leading / trailing columns get equal pointers."""
    import scipy.sparse as sp
    import torch
    from exahip import CompressedExaModel, ExaModel
    m = ExaModel(ZOO[name]())
    cm = CompressedExaModel(m)
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=6)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    for which, nrow in (("jac", m.meta.ncon), ("hess", m.meta.nvar)):
        colptr, rowval = (t.cpu().numpy() for t in cm.csc(which))
        vals = (cm.jac_coord(xd) if which == "jac" else cm.hess_coord(xd, yd, sigma)).cpu().numpy()
        assert colptr[0] == 1 and colptr[-1] == len(vals) + 1 and np.all(np.diff(colptr) >= 0)
        A = sp.csc_matrix((vals, rowval - 1, colptr - 1), shape=(max(nrow, 1), m.meta.nvar))
        if which == "jac":
            r, c = m.jac_structure(); v = m.jac_coord(x)
        else:
            r, c = m.hess_structure(); v = m.hess_coord(x, y, sigma)
        B = sp.coo_matrix((v, (r - 1, c - 1)), shape=A.shape).tocsc()
        B.sum_duplicates()
        assert abs(A - B).max() <= 1e-12 * max(1.0, abs(B).max())
        assert A.has_sorted_indices or A.nnz == 0


def test_block_owned_windows_for_separate_variable_arrays(libs):
    """A discretised control problem laid out as separate arrays x[0..N], u[0..N], z[0..N]: the Hessian / Jacobian entries
    of one time step land in three far-apart column blocks.  exa_compress must pick the block-owned windows (every
    pattern evaluated once per point, one LDS window per column block) and reproduce the uncompressed sums; the step
    length `h` shared by all steps goes through the shared-entry kernel, the boundary conditions through the tail."""
    from exahip import CompressedExaModel, ExaCore, ExaModel
    from exahip.core import rng
    from exahip.graph import cos, exp, sin
    N = 5000
    c = ExaCore()
    x = c.add_var(N + 1, start=np.linspace(0.2, 1.0, N + 1))
    u = c.add_var(N + 1, start=np.linspace(-0.5, 0.5, N + 1))
    z = c.add_var(N + 1, start=0.3)
    h = c.add_var(1, start=0.01, lvar=1e-4)
    c.add_obj(lambda i: h[1] * (u[i] ** 2 + x[i] ** 2 * z[i]), rng(1, N + 1))
    c.add_con(lambda i: x[i + 1] - x[i] - h[1] * (sin(x[i]) * u[i] + z[i + 1] * x[i + 1]), rng(1, N))
    c.add_con(lambda i: z[i + 1] - z[i] - 0.5 * h[1] * (exp(-x[i]) * z[i] + cos(u[i + 1]) * z[i + 1]), rng(1, N))
    c.add_con(lambda i: x[i] - 0.2, rng(1, 1))
    c.add_con(lambda i: z[i] ** 2 - 0.09, rng(1, 1))
    m = ExaModel(c.to_ir())
    cm = CompressedExaModel(m)
    kinds = _check_against_uncompressed(m, cm, 3)
    assert kinds == ["windowed", "windowed"]
    assert "block-owned" in cm.path("hess")[1] and "block-owned" in cm.path("jac")[1], (cm.path("jac"), cm.path("hess"))


@pytest.mark.parametrize("seed", range(6))
def test_windowed_random_models_over_three_variable_blocks(libs, seed):
    """Random unit-range models whose symbolic indices fall into three far-apart variable blocks (x[i+c], x[n+i+c],
    x[2n+i+c]): the compressed entries of one point land in up to three column blocks — the shape the block-owned windows
    are for (whichever shape exa_compress picks must reproduce the uncompressed sums)."""
    import randexpr
    from exahip import CompressedExaModel, ExaModel
    m = ExaModel(randexpr.build_range_model(seed, npts=2000, npat=5, depth=3, unit=True, blocks=True).to_ir())
    cm = CompressedExaModel(m)
    # slot values that are themselves differences of large terms differ between two compilations of the same expression
    # (FMA contraction) by more than the summation bound: 1e-11 of the summed magnitudes still exposes any wrong window
    kinds = _check_against_uncompressed(m, cm, seed, tol=1e-11)
    print(seed, kinds, cm.path("jac"), cm.path("hess"))


def test_no_compiler_for_the_window_module_falls_back_to_the_gather(libs, monkeypatch, tmp_path):
    """exa_compress generates a second module for the windowed sweep; where it cannot be compiled (a packed library's
    consumer without hipcc) the call must still succeed and evaluate through the gather."""
    import shutil
    import torch
    from exahip import CompressedExaModel, ExaModel, models
    m = ExaModel(models.luksan_vlcek_model(777))                  # base module compiled (or cached) with the real compiler
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))          # empty cache: the window module is not there
    monkeypatch.setenv("EXAHIP_COMPILER", "hipcc")
    monkeypatch.setenv("EXAHIP_HIPCC", "/nonexistent/hipcc")
    cm = CompressedExaModel(m)
    kind, why = cm.path("hess")
    assert kind == "gather" and "could not be built" in why, (kind, why)
    assert _check_against_uncompressed(m, cm, 1) == ["gather", "gather"]


@pytest.mark.parametrize("seed", range(5))          # (its second module is compiled on the GPU box at exa_compress: 8 s a seed)
def test_random_table_models_take_the_merged_permuted_store(libs, seed):
    """Random models whose 12 patterns all iterate ONE table (tests/randexpr.py build_model): identical columns are
    aliased, the patterns form fused groups, and the compressed Hessian goes through merged slots + permuted store
    wherever the groups collapse enough slots (else the plain permuted store).  Whatever exa_compress picks must give the
    duplicate-summed uncompressed matrix — also from a larger table (several workgroups, partial tiles)."""
    import randexpr
    from exahip import CompressedExaModel, ExaModel
    for npts in (randexpr.NPTS, 700):
        saved = randexpr.NPTS
        randexpr.NPTS = npts
        try:
            m = ExaModel(randexpr.build_model(seed, npat=12, depth=3).to_ir())
        finally:
            randexpr.NPTS = saved
        cm = CompressedExaModel(m)
        kinds = _check_against_uncompressed(m, cm, seed, tol=1e-11)
        assert all(k in ("scatter", "gather", "windowed") for k in kinds), kinds
        print(seed, npts, kinds, cm.path("hess")[1])
        if npts == 7:           # (the larger table piles > 512 duplicates on some entries: those keep the cooperative gather)
            assert "scatter" in kinds, (kinds, cm.path("jac"), cm.path("hess"))


@pytest.mark.parametrize("name,size", [("lv", 200_000), ("rocket", 40_000), ("lv", 1000)])
def test_a_shard_compresses_its_local_slice_through_the_windows(libs, name, size, monkeypatch):
    """exa_compress of a sharded model (local-slice COO): every rank plans the windowed sweep over ITS data points
    (absolute point indices, the packed slice's offsets) — same kernels as the unsharded model, no gather.  Each rank's
    compressed matrix must equal the reference compression (gather) of the same slice entry for entry, and the ranks'
    matrices must add up to the unsharded compressed matrix."""
    import torch
    from exahip import CompressedExaModel, ExaModel, models
    build = (lambda: models.luksan_vlcek_model(size)) if name == "lv" else (lambda: models.rocket_model(size))
    world = 3
    dev = torch.device("cuda:0")
    whole = ExaModel(build())
    cw = CompressedExaModel(whole)
    x, y, sigma = point(whole.meta.x0, whole.meta.ncon, seed=12)
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    nvar, ncon = whole.meta.nvar, max(whole.meta.ncon, 1)
    total = {"jac": {}, "hess": {}}
    for rank in range(world):
        both = []
        for window in ("1", "0"):
            monkeypatch.setenv("EXAHIP_CWINDOW", window)
            monkeypatch.setenv("EXAHIP_CSCATTER", window)
            m = ExaModel(build())
            m.set_shard(rank, world)
            m.set_coo_local(True)
            c = CompressedExaModel(m)
            if window == "1" and size > 1000:
                assert c.path("hess")[0] == "windowed" and c.path("jac")[0] == "windowed", (c.path("hess"), c.path("jac"))
            if window == "0":
                assert c.path("hess")[0] == "gather" and c.path("jac")[0] == "gather"
            both.append((c.jac_structure(), c.jac_coord(xd), c.hess_structure(), c.hess_coord(xd, yd, sigma)))
        (js1, jv1, hs1, hv1), (js0, jv0, hs0, hv0) = both
        for s1, s0 in ((js1, js0), (hs1, hs0)):
            assert torch.equal(s1[0], s0[0]) and torch.equal(s1[1], s0[1])
        for a, b in ((jv1, jv0), (hv1, hv0)):
            a, b = a.cpu().numpy(), b.cpu().numpy()
            assert np.all(np.abs(a - b) <= 1e-12 * np.abs(b) + 1e-13 * np.abs(b).max()), np.abs(a - b).max()
        for kind, (rows, cols), vals, nrow in (("jac", js1, jv1, ncon), ("hess", hs1, hv1, nvar)):
            key = ((cols - 1) * nrow + (rows - 1)).cpu().numpy()
            for k, v in zip(key.tolist(), vals.cpu().numpy().tolist()):
                total[kind][k] = total[kind].get(k, 0.0) + v
    for kind, (rows, cols), vals, nrow in (("jac", cw.jac_structure(), cw.jac_coord(xd), ncon), ("hess", cw.hess_structure(), cw.hess_coord(xd, yd, sigma), nvar)):
        key = ((cols - 1) * nrow + (rows - 1)).cpu().numpy()
        ref = vals.cpu().numpy()
        assert set(key.tolist()) == set(total[kind])                   # the union of the ranks' structures is the model's
        got = np.array([total[kind][k] for k in key.tolist()])
        assert np.all(np.abs(got - ref) <= 1e-11 * np.abs(ref) + 1e-12 * np.abs(ref).max()), np.abs(got - ref).max()
