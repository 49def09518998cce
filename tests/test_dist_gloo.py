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
