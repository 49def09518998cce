"""-m gpu: the N > 1 host layer with the real HIP evaluator.  Two processes (both on cuda:0 — the test box has one
GPU — over gloo) shard every iterator with exa_set_shard through exahip.dist.ShardedEvaluator: obj / grad / cons are
completed by all_reduce, the Jacobian / Hessian COO slices are disjoint and their union is the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "examodels.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from exahip import ExaModel
        from exahip.dist import ShardedEvaluator
        from zoo import ZOO, point
        dev = torch.device("cuda:0")
        out = {}
        for name in ("lv1000", "acopf30", "rocket50"):
            core = ZOO[name]()
            m = ExaModel(core)
            ev = ShardedEvaluator(m)
            o = oracle.OracleModel(m.ir)
            x, y, s = point(m.meta.x0, m.meta.ncon, seed=17)
            xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
            f = ev.obj(xd)
            g = ev.grad(xd).cpu().numpy()
            c = ev.cons(xd).cpu().numpy()
            h = torch.zeros(m.meta.nnzh, dtype=torch.float64, device=dev)
            ev.hess_coord(xd, yd, s, out=h)
            mine = (h != 0).cpu().numpy()                 # this rank's slice (a genuine zero entry counts for neither)
            whole = ev.gather_coo(h).cpu().numpy()
            j = torch.zeros(m.meta.nnzj, dtype=torch.float64, device=dev)
            ev.jac_coord(xd, out=j)
            jw = ev.gather_coo(j, hess=False).cpu().numpy()
            H, J = o.hess_coord(x, y, s), o.jac_coord(x)
            ok = (abs(f - o.obj(x)) <= 1e-10 * max(1, abs(o.obj(x))) and np.allclose(g, o.grad(x), rtol=1e-10, atol=1e-12)
                  and np.allclose(c, o.cons(x), rtol=1e-10, atol=1e-12) and np.allclose(whole, H, rtol=1e-10, atol=1e-12)
                  and np.allclose(jw, J, rtol=1e-10, atol=1e-12))
            assert ev.transport == "hook" and m.comm_info() == (rank, world, "hook")       # reduced inside libexahip
            # products through the same hook
            v = np.random.default_rng(3).standard_normal(m.meta.nvar)
            w = np.random.default_rng(4).standard_normal(m.meta.ncon)
            vd, wd = torch.from_numpy(v).to(dev), torch.from_numpy(w).to(dev)
            ok = ok and np.allclose(ev.jprod(xd, vd).cpu().numpy(), o.jprod(x, v), rtol=1e-10, atol=1e-12)
            ok = ok and np.allclose(ev.jtprod(xd, wd).cpu().numpy(), o.jtprod(x, w), rtol=1e-10, atol=1e-12)
            ok = ok and np.allclose(ev.hprod(xd, yd, vd, s).cpu().numpy(), o.hprod(x, y, v, s), rtol=1e-10, atol=1e-12)
            # local-slice COO: slice-sized buffers, pieces placed by exa_coo_slices reproduce this rank's part exactly
            m.set_coo_local(True)
            hl = m.hess_coord(xd, yd, s).cpu().numpy()
            jl = m.jac_coord(xd).cpu().numpy()
            hr, hc = m.hess_structure()
            jr, jc = m.jac_structure()
            Hr, Hc = o.hess_structure()
            Jr, Jc = o.jac_structure()
            assert hl.size == m.local_nnzh < m.meta.nnzh and jl.size == m.local_nnzj
            for (vals, rows, cols, ref, rref, cref, hess) in ((hl, hr, hc, H, Hr, Hc, True), (jl, jr, jc, J, Jr, Jc, False)):
                for g0, l0, cnt in m.coo_slices(hess):
                    ok = ok and np.allclose(vals[l0:l0 + cnt], ref[g0:g0 + cnt], rtol=1e-10, atol=1e-12)
                    ok = ok and np.array_equal(rows[l0:l0 + cnt], rref[g0:g0 + cnt]) and np.array_equal(cols[l0:l0 + cnt], cref[g0:g0 + cnt])
            # sorted (deterministic) products and the compressed COO work on the local slice; the model's matrix is the
            # sum of the ranks' matrices
            m.set_product_mode(1, 1)
            ok = ok and np.allclose(ev.jtprod(xd, wd).cpu().numpy(), o.jtprod(x, w), rtol=1e-10, atol=1e-12)
            ok = ok and np.allclose(ev.hprod(xd, yd, vd, s).cpu().numpy(), o.hprod(x, y, v, s), rtol=1e-10, atol=1e-12)
            from exahip import CompressedExaModel
            cm = CompressedExaModel(m)
            cr, cc = cm.hess_structure()
            cv = cm.hess_coord(xd, yd, s)
            dense = torch.zeros(m.meta.nvar * m.meta.nvar, dtype=torch.float64, device=dev)
            dense.index_add_(0, (cr - 1) * m.meta.nvar + (cc - 1), cv)
            dense = ev.sum_buffer(dense).cpu().numpy().reshape(m.meta.nvar, m.meta.nvar)
            Hd = np.zeros((m.meta.nvar, m.meta.nvar))
            np.add.at(Hd, (Hr - 1, Hc - 1), H)
            ok = ok and np.allclose(dense, Hd, rtol=1e-10, atol=1e-11) and cm.meta.nnzh <= m.local_nnzh
            out[name] = (bool(ok), int(mine.sum()))
        q.put((rank, out))
    except Exception as e:      # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_work(libs):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert isinstance(res[r], dict), res[r]
        assert all(ok for ok, _ in res[r].values()), res[r]
    # both ranks produced part of every Hessian
    for name in res[0]:
        assert res[0][name][1] > 0 and res[1][name][1] > 0
