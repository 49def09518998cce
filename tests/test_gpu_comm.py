"""-m gpu: the multi-GPU layer BEHIND THE C ABI (include/exahip.h exa_comm_*) and the explicit tuner.

The test box has one MI355X, so:
  * the RCCL code path runs with a world-1 communicator (ncclCommInitRank + ncclAllReduce execute on the hardware);
  * two ranks share cuda:0 and reduce through exa_comm_hook over gloo (tests/test_gpu_dist.py) — the same ABI hook an
    MPI.jl host would use.
What needs reducing is what the reference accumulates over data points on one device: obj (KA ext :253-271),
grad! (:310-336), cons_nln! (:273-308) and the products."""
import os
import time

import numpy as np
import pytest

from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _all_outputs(m, xd, yd, vd, wd, s):
    import torch
    f, c, j, h = m.eval_fused(xd, yd, s)
    out = {"obj": m.obj(xd), "grad": m.grad(xd), "cons": m.cons(xd), "jprod": m.jprod(xd, vd), "jtprod": m.jtprod(xd, wd),
           "hprod": m.hprod(xd, yd, vd, s), "fused_obj": f, "fused_c": c, "jac": m.jac_coord(xd), "hess": m.hess_coord(xd, yd, s)}
    torch.cuda.synchronize()
    return {k: (v.cpu().numpy().copy() if hasattr(v, "cpu") else v) for k, v in out.items()}


@pytest.mark.parametrize("name", ["acopf30", "rocket50", "lv1000"])
def test_rccl_allreduce_runs_behind_the_abi_world1(libs, name):
    """exa_comm_unique_id + exa_comm_init create a real RCCL communicator; every reducing callback then enqueues
    ncclAllReduce on the model's stream.  With one rank the sum is the identity, so the results must not move."""
    import torch
    from exahip import ExaModel
    m = ExaModel(ZOO[name]())
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=21)
    dev = torch.device("cuda:0")
    v = np.random.default_rng(1).standard_normal(m.meta.nvar)
    w = np.random.default_rng(2).standard_normal(m.meta.ncon)
    xd, yd, vd, wd = (torch.from_numpy(a).to(dev) for a in (x, y, v, w))
    m.set_product_mode(1, 1)                         # deterministic products: bitwise comparison below
    before = _all_outputs(m, xd, yd, vd, wd, s)
    uid = m.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    m.comm_init(0, 1, uid)
    assert m.comm_info() == (0, 1, "rccl")
    after = _all_outputs(m, xd, yd, vd, wd, s)
    for k in before:
        if k == "grad" and name == "acopf30":       # FP64 atomics on a data-indexed objective: order varies
            np.testing.assert_allclose(after[k], before[k], rtol=1e-13, atol=1e-13)
        else:
            assert np.array_equal(np.asarray(after[k]), np.asarray(before[k])), k
    t = torch.arange(5, dtype=torch.float64, device=dev)
    assert torch.equal(m.allreduce(t.clone()), t)
    m.set_reduce(False)
    assert m.obj(xd) == before["obj"]
    m.comm_free()
    assert m.comm_info() == (0, 1, "none")


@pytest.mark.parametrize("name", ["lv1000", "rocket50", "acopf30"])
def test_rccl_allgather_branch_runs_on_one_device(libs, name):
    """everything the first multi-GPU run will execute for the first time, on ONE device.  A real RCCL communicator of
    world 1; exa_allgather_coo (forced gather) and exa_comm_complete issue the ONE-PIECE plans of the COO vectors, grad!, cons_nln! and the
    products through rccl_run_plan_f64 — in-place ncclAllGather where the vector is sharded by owner, ncclAllReduce where it is left as partial
    sums — inside ncclGroupStart / End on the model's stream.  Gathering / summing over one rank is the identity: not a bit may move."""
    import torch
    from exahip import ExaModel
    m = ExaModel(ZOO[name]())
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=22)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    m.comm_init(0, 1, m.comm_unique_id())
    assert m.comm_info() == (0, 1, "rccl")
    m.set_reduce(False)                              # the callbacks leave their vectors alone: completion is issued below, explicitly
    vecs = {1: m.grad(xd).clone(), 2: m.cons(xd).clone(), 3: m.jac_coord(xd).clone(), 4: m.hess_coord(xd, yd, s).clone(),
            6: m.jtprod(xd, yd).clone(), 7: m.hprod(xd, yd, xd, s).clone()}
    torch.cuda.synchronize()
    kinds = set()
    for which, v in vecs.items():
        if v.numel() == 0:
            continue
        layout = m._L.exa_shard_layout(m.id, which)
        kinds.add(0 if which in (3, 4) or layout == 1 else 2)
        before = v.clone()
        m.comm_complete(which, v)
        torch.cuda.synchronize()
        assert torch.equal(v, before), which
    assert 0 in kinds                                # at least the COO vectors went through ncclAllGather
    for hess, v in ((False, vecs[3]), (True, vecs[4])):
        if v.numel():
            out = m.allgather_coo(v.clone(), hess=hess)
            torch.cuda.synchronize()
            assert torch.equal(out, v)
    m.comm_free()
    from exahip import capi
    with pytest.raises(capi.ExaHipError, match="needs an RCCL communicator"):
        m.comm_complete(1, vecs[1])


def test_host_reducer_hook_is_called_for_every_reducing_callback(libs):
    """exa_comm_hook: the library hands (device buffer, count, stream) of exactly the vectors that need the sum."""
    import torch
    from exahip import ExaModel
    m = ExaModel(ZOO["acopf30"]())
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=3)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    vd, wd = torch.ones(m.meta.nvar, dtype=torch.float64, device=dev), torch.ones(m.meta.ncon, dtype=torch.float64, device=dev)
    seen = []
    m.comm_hook(0, 1, lambda ptr, count, stream: seen.append(count) or 0)
    nv, nc = m.meta.nvar, m.meta.ncon
    m.obj(xd); m.grad(xd); m.cons(xd); m.jprod(xd, vd); m.jtprod(xd, wd); m.hprod(xd, yd, vd, s)
    # obj: 1 double; grad (the generator costs reach pg through a data column), J'v, Hv (atomics): partial sums of nvar.
    # cons / jprod: NOTHING is summed — a row's owner evaluates the row's augmentation terms itself (exa_cons1); ranks hold
    # complete row slices (made whole by all-gather-v, which a one-rank model does not need)
    assert seen == [1, nv, nv, nv]
    seen.clear()
    m.jac_coord(xd); m.hess_coord(xd, yd, s); m.jac_structure(); m.hess_structure()
    assert seen == []                                 # COO outputs need no collective
    m.eval_fused(xd, yd, s)
    assert seen == [1]
    # a stencil model (every index range-affine) is sharded by OWNER throughout: grad by variable range, cons by row, the
    # products by window — only obj is a sum
    lv = ExaModel(ZOO["lv1000"]())
    xl = torch.from_numpy(np.asarray(lv.meta.x0) + 0.01).to(dev)
    yl, vl = torch.ones(lv.meta.ncon, dtype=torch.float64, device=dev), torch.ones(lv.meta.nvar, dtype=torch.float64, device=dev)
    seen.clear()
    lv.comm_hook(0, 1, lambda ptr, count, stream: seen.append(count) or 0)
    lv.grad(xl); lv.cons(xl); lv.jprod(xl, vl); lv.jtprod(xl, yl); lv.hprod(xl, yl, vl, s); lv.hess_coord(xl, yl, s)
    assert seen == []
    lv.obj(xl)
    assert seen == [1]
    seen.clear()
    # a failing reducer is a status-2 error, not a crash
    from exahip import capi
    m.comm_free()
    m.comm_hook(0, 1, lambda *a: 7)
    with pytest.raises(capi.ExaHipError, match="status 2.*hook returned status 7"):
        m.grad(xd)


def test_tuning_is_explicit_persisted_and_never_inside_a_callback(libs, tmp_path, monkeypatch):
    """exa_tune measures block orders / product implementations once and persists the decisions next to the cached
    module; a later model with the same module, device and sizes starts with them, and no callback ever blocks to measure."""
    import torch
    from exahip import ExaModel, models
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))
    N = 3_000_000                                    # two patterns streaming >= 128 MB: both block orders exist
    dev = torch.device("cuda:0")
    m = ExaModel(models.luksan_vlcek_model(N))
    xd = torch.from_numpy(m.meta.x0 + 0.05).to(dev)
    yd = torch.ones(m.meta.ncon, dtype=torch.float64, device=dev)
    h = torch.empty(m.meta.nnzh, dtype=torch.float64, device=dev)
    # first call of a fresh model: asynchronous, nothing measured (the old in-callback tuner spent >= 60 ms here)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.hess_coord(xd, yd, 0.5, out=h)
    first_call_ms = 1e3 * (time.perf_counter() - t0)
    torch.cuda.synchronize()
    assert first_call_ms < 30.0, first_call_ms
    # nobody tuned: the PLAN-TIME default — interleaved, because both patterns walk x[1..N] and the call streams < 1.5 GB (fill_params)
    assert m._L.exa_block_order(m.id, 4) == 1 and m.product_mode() == (-1, -1)
    ref = h.clone()
    m.tune(3, xd, yd)
    files = [f for f in os.listdir(tmp_path) if f.endswith(".tune")]
    assert len(files) == 1 and files[0].startswith(m._L.exa_module_name(m.id).decode())
    order = [m._L.exa_block_order(m.id, w) for w in (2, 3, 4, 5)]
    modes = m.product_mode()
    assert all(o in (0, 1) for o in order) and all(v in (0, 1, 2) for v in modes)
    m.hess_coord(xd, yd, 0.5, out=h)
    assert torch.equal(h, ref)                       # the order of the workgroups does not change a single bit
    m2 = ExaModel(models.luksan_vlcek_model(N))
    assert [m2._L.exa_block_order(m2.id, w) for w in (2, 3, 4, 5)] == order
    g = m2.jtprod(xd, yd); m2.hprod(xd, yd, xd, 0.5)
    assert m2.product_mode() == modes
    np.testing.assert_allclose(g.cpu().numpy(), m.jtprod(xd, yd).cpu().numpy(), rtol=1e-12, atol=1e-12)
    # another size is another decision
    m3 = ExaModel(models.luksan_vlcek_model(N + 1))
    m3.jtprod(torch.from_numpy(m3.meta.x0).to(dev), torch.ones(m3.meta.ncon, dtype=torch.float64, device=dev))
    assert m3.product_mode()[0] == 2 and m3._L.exa_block_order(m3.id, 4) == 1       # untuned: owner-computes windows (LV has them), the plan-time order
    # ... and beyond 1.5 GB per call (the output no longer fits the Infinity Cache next to x) the plan-time order is the sequential one
    big = ExaModel(models.luksan_vlcek_model(20_000_000))
    assert big._L.exa_block_order(big.id, 4) == 0 and big._L.exa_hess_variant(big.id) in (1, 2)


def test_build_info_reports_how_the_module_was_obtained(libs, tmp_path, monkeypatch):
    from exahip import ExaModel, models
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("EXAHIP_HIPCC", "/nonexistent/hipcc")      # the in-process path needs no hipcc
    monkeypatch.delenv("EXAHIP_COMPILER", raising=False)          # (a suite run under EXAHIP_COMPILER=hipcc: this test is about hiprtc)
    t0 = time.perf_counter()
    m = ExaModel(models.luksan_vlcek_model(100))
    how, ms = m.build_info()
    assert how == "hiprtc" and 0 < ms < 1e3 * (time.perf_counter() - t0)
    assert ExaModel(models.luksan_vlcek_model(200)).build_info() == ("disk", 0.0)
    import oracle
    x = m.meta.x0 + 0.01
    np.testing.assert_allclose(m.grad(x), oracle.OracleModel(m.ir).grad(x), rtol=1e-12)


def test_eight_shards_from_resident_slices_reproduce_the_model(libs):
    """What `bench.py --gpus 8` (config 5) runs on every rank, replayed rank by rank on one GPU: shard r of 8, COO as a
    packed local slice, and ONLY the stretch of x / y the shard reads resident — handed over as a pointer shifted back
    by the stretch's first index.  The eight local slices, placed by exa_coo_slices, are the unsharded Hessian."""
    import ctypes
    import sys
    import torch
    import oracle
    from exahip import ExaModel, models
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import resident_ranges
    N, world = 100_003, 8
    m = ExaModel(models.luksan_vlcek_model(N))
    o = oracle.OracleModel(m.ir)
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=5)
    H = o.hess_coord(x, y, s)
    Hr, Hc = o.hess_structure()
    dev = torch.device("cuda:0")
    got = np.full(m.meta.nnzh, np.nan)
    L = m._L
    for rank in range(world):
        m.set_shard(rank, world)
        m.set_coo_local(True)
        vlo, vhi, ylo, yhi = resident_ranges(m, rank, world)
        # + the windows an owner-computes product reaches; the LAST rank also takes the remainder of the equal split (< world windows of
        # 248 variables, < world points: what makes the other ranks' pieces equal — one in-place all-gather, exa_collective_plan)
        # (window shares — whole windows of 248 variables — and point shares drift apart by less than a window per rank)
        assert vhi - vlo <= N // world + 4 + 2 * 256 + world * 248 and yhi - ylo <= N // world + world
        xs = torch.from_numpy(x[vlo:vhi].copy()).to(dev)
        ys = torch.from_numpy(y[ylo:yhi].copy()).to(dev)
        n = m.local_nnzh
        assert abs(n - m.meta.nnzh / world) <= 9 * world
        h = torch.full((n + 16,), float("nan"), dtype=torch.float64, device=dev)
        rc = L.exa_hess(m.id, ctypes.c_void_p(xs.data_ptr() - 8 * vlo), ctypes.c_void_p(ys.data_ptr() - 8 * ylo), s, ctypes.c_void_p(h.data_ptr()))
        assert rc == 0
        rows = torch.zeros(n, dtype=torch.int64, device=dev)
        cols = torch.zeros(n, dtype=torch.int64, device=dev)
        assert L.exa_hess_structure64(m.id, ctypes.c_void_p(rows.data_ptr()), ctypes.c_void_p(cols.data_ptr())) == 0
        torch.cuda.synchronize()
        hv, rv, cv = h.cpu().numpy(), rows.cpu().numpy(), cols.cpu().numpy()
        assert np.all(np.isnan(hv[n:])) and not np.any(np.isnan(hv[:n]))          # exactly the slice, nothing beyond
        for g0, l0, cnt in m.coo_slices(True):
            assert np.all(np.isnan(got[g0:g0 + cnt]))
            got[g0:g0 + cnt] = hv[l0:l0 + cnt]
            assert np.array_equal(rv[l0:l0 + cnt], Hr[g0:g0 + cnt]) and np.array_equal(cv[l0:l0 + cnt], Hc[g0:g0 + cnt])
    np.testing.assert_allclose(got, H, rtol=1e-10, atol=0)


@pytest.mark.parametrize("world", [8])
def test_bench_launch_path_with_eight_ranks(libs, world):
    """`bench.py --gpus 8` exactly as the driver launches it (torch.distributed.run, one rank per process), on the one GPU of
    the test box: EXAHIP_DIST_BACKEND=gloo makes the eight ranks share cuda:0 and reduce through exa_comm_hook.  The numbers
    mean nothing; the JSON contract, config 5 as the workload, the strong-scaling split, local-slice buffers tiling nnzh, the
    in-library collectives and their plan must all be there — the first 8-GPU run of the driver takes exactly this path with
    RCCL in place of the hook (round 3 exercised it with two ranks only)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, EXAHIP_DIST_BACKEND="gloo")
    points = 800_000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "5", "--warmup", "2", "--points", str(points),
           "--preheat-ms", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "strong" and d["unit"] == "nnz/s"
    assert d["config"]["baseline_config"] == 5 and f"split over {world} GPUs" in d["config"]["workload"]
    assert d["config"]["parallelism"] == f"iterator-shard x{world}, local-slice COO, no data-path collective"
    assert d["config"]["nnzh"] == 9 * points - 15
    # the eight packed local slices tile the Hessian COO: equal shares, the remainder on the last rank
    loc = d["config"]["local_nnzh_per_rank"]
    assert len(loc) == world and sum(loc) == d["config"]["nnzh"] and len(set(loc[:-1])) == 1 and 0 <= loc[-1] - loc[0] < 9 * world
    # each rank holds an eighth of the COO and its stencil stretch of x / y, not the model
    assert d["config"]["resident_per_gpu_bytes"] < (1.0 / world + 0.01) * 8 * (9 + 1 + 1) * points + 4096
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["higher_is_better"] is True
    assert d["collectives"].get("transport") == "hook" and d["collectives"]["grad_plus_allreduce_ms"] > 0, d["collectives"]
    assert d["collectives"]["coo_allgather_ms"] > 0 and d["collectives"]["coo_allgather_bytes"] == 8 * d["config"]["nnzh"]
    assert "all-gather-v" in d["collectives"]["grad_collective"]             # LV: grad! is sharded by variable owner
    assert d["collectives"]["n_ranks_seen"] == world                       # (RCCL: ncclCommCount; here the hook's world)
    # ... completed by ONE in-place all-gather of equal pieces (800 000 variables over 8 ranks: no surplus); the gathered Hessian:
    # one all-gather per pattern + the last rank's surplus slots as a broadcast
    assert d["collectives"]["grad_plan"] == [{"op": "allgather", "offset": 0, "count": points // world, "root": -1}]
    plan = d["collectives"]["coo_allgather_plan"]
    assert [op["op"] for op in plan].count("allgather") == 2 and all(op["op"] in ("allgather", "broadcast") for op in plan)
    assert sum(op["count"] * (world if op["op"] == "allgather" else 1) for op in plan) == d["config"]["nnzh"]
    # the same workload on one GPU, measured in the same run by rank 0: what a scaling ratio has to be computed against
    assert d["scale_base"]["n_gpus"] == 1 and d["scale_base"]["value"] > 0 and d["speedup_vs_scale_base"] > 0
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["value"] > 0
