"""-m gpu: the callbacks are asynchronous launches on the caller's stream with no allocation or synchronisation once a
callback has run once (SURVEY §8b "asynchronous w.r.t. the device"), so a solver iteration — obj, grad!, cons_nln!,
jac_coord!, hess_coord!, the products — can be captured into a hipGraph by the HOST application and replayed on new
iterates held in the same buffers.  (Measured on MI355X: a replay is not faster than the separate launches, which
already pipeline — ACOPF 0.055 vs 0.060 ms per iteration — so the library does not build graphs itself.)"""
import ctypes

import numpy as np
import pytest

from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


@pytest.mark.parametrize("name", ["acopf30", "rocket50", "lv1000", "conaug2d", "mixed"])
def test_iteration_captures_into_a_hip_graph_and_replays(libs, name):
    import torch
    from exahip import ExaModel
    m = ExaModel(ZOO[name]())
    L = m._L
    x, y, sigma = point(m.meta.x0, m.meta.ncon, seed=31)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    vd = torch.from_numpy(np.random.default_rng(3).standard_normal(m.meta.nvar)).to(dev)
    wd = torch.from_numpy(np.random.default_rng(4).standard_normal(max(1, m.meta.ncon))).to(dev)
    new = lambda n: torch.empty(max(1, n), dtype=torch.float64, device=dev)
    f, g, c, j, h = new(1), new(m.meta.nvar), new(m.meta.ncon), new(m.meta.nnzj), new(m.meta.nnzh)
    jv, jtv, hv = new(m.meta.ncon), new(m.meta.nvar), new(m.meta.nvar)
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def iteration():
        calls = [L.exa_obj_async(m.id, p(xd), p(f)), L.exa_grad(m.id, p(xd), p(g)), L.exa_cons(m.id, p(xd), p(c)),
                 L.exa_jac(m.id, p(xd), p(j)), L.exa_hess(m.id, p(xd), p(yd), ctypes.c_double(sigma), p(h)),
                 L.exa_jprod(m.id, p(xd), p(vd), p(jv)), L.exa_jtprod(m.id, p(xd), p(wd), p(jtv)),
                 L.exa_hprod(m.id, p(xd), p(yd), p(vd), ctypes.c_double(sigma), p(hv))]
        assert not any(calls), L.exa_last_error()

    outs = (f, g, c, j, h, jv, jtv, hv)

    def same(A, B):
        # bit-identical, except J'v and Hv: their scatter adds are FP64 atomics whose order is not fixed
        for k, (a, b) in enumerate(zip(A, B)):
            if k < 6:
                assert torch.equal(a, b), k
            else:
                assert float((a - b).abs().max()) <= 1e-12 * (1.0 + float(b.abs().max())), k
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        L.exa_set_stream(m.id, ctypes.c_void_p(s.cuda_stream))
        iteration()                                  # first use: scratch buffers are allocated here, not under capture
        s.synchronize()
        ref = [t.clone() for t in outs]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            iteration()
        for t in outs:
            t.fill_(-7.0)
        graph.replay()
        s.synchronize()
        same(ref, outs)
        xd.mul_(1.01)                                # a new iterate in the same buffer: replay, no re-capture
        yd.mul_(0.5)
        graph.replay()
        s.synchronize()
        got = [t.clone() for t in outs]
        iteration()
        s.synchronize()
        same(got, outs)
    L.exa_set_stream(m.id, ctypes.c_void_p(0))
