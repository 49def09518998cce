"""-m gpu: a solver iteration's evaluation set (fused obj+cons+jac+hess sweep, grad!, hprod!) captured into ONE
hipGraph through torch.cuda.graph and replayed: the library's launches are plain stream work (no synchronisation, no
host round trip), so they are capturable; replays at new x must equal fresh evaluations."""
import numpy as np
import pytest

from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


@pytest.mark.parametrize("name", ["lv1000", "acopf30", "rocket50"])
def test_evaluation_set_replays_from_a_hip_graph(libs, name):
    import torch
    from exahip import ExaModel
    m = ExaModel(ZOO[name]())
    dev = torch.device("cuda:0")
    x0, y0, sigma = point(m.meta.x0, m.meta.ncon, seed=4)
    x, y = torch.from_numpy(x0).to(dev), torch.from_numpy(y0).to(dev)
    v = torch.from_numpy(np.random.default_rng(7).standard_normal(m.meta.nvar)).to(dev)
    f = torch.zeros(1, dtype=torch.float64, device=dev)
    c = torch.zeros(m.meta.ncon, dtype=torch.float64, device=dev)
    j = torch.zeros(m.meta.nnzj, dtype=torch.float64, device=dev)
    h = torch.zeros(m.meta.nnzh, dtype=torch.float64, device=dev)
    g = torch.zeros(m.meta.nvar, dtype=torch.float64, device=dev)
    hv = torch.zeros(m.meta.nvar, dtype=torch.float64, device=dev)

    def evaluate():
        m.eval_fused(x, y, sigma, c=c, jac=j, hess=h, obj_out=f)
        m.grad(x, out=g)
        m.hprod(x, y, v, sigma, out=hv)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        evaluate()                    # first call outside the capture: block-order / product-mode measurements happen here
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        evaluate()
    # replay at a NEW point written into the captured input buffers
    x1, y1, _ = point(m.meta.x0, m.meta.ncon, seed=11)
    x.copy_(torch.from_numpy(x1))
    y.copy_(torch.from_numpy(y1))
    for t in (f, c, j, h, g, hv):
        t.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    got = [t.clone() for t in (f, c, j, h, g, hv)]
    xe, ye = torch.from_numpy(x1).to(dev), torch.from_numpy(y1).to(dev)
    ef, ec, ej, eh = m.eval_fused(xe, ye, sigma)
    eg = m.grad(xe)
    ehv = m.hprod(xe, ye, v, sigma)
    torch.cuda.synchronize()
    for a, b in zip(got[:4], (ef, ec, ej, eh)):
        assert torch.equal(a, b)                     # same kernels, same order of operations: bit-identical
    for a, b in zip(got[4:], (eg, ehv)):             # scatter kernels use FP64 atomics: order of additions may differ
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", ["lv1000", "rocket50", "acopf30"])
def test_compressed_evaluation_replays_from_a_hip_graph(libs, name):
    """exa_cjac / exa_chess (windowed sweep with its shared-entry and end-point kernels, or evaluation + gather) are
    plain stream work too: captured, replayed at a new point, bit-identical to a fresh evaluation (fixed orders)."""
    import torch
    from exahip import CompressedExaModel, ExaModel
    m = ExaModel(ZOO[name]())
    cm = CompressedExaModel(m)
    dev = torch.device("cuda:0")
    x0, y0, sigma = point(m.meta.x0, m.meta.ncon, seed=4)
    x, y = torch.from_numpy(x0).to(dev), torch.from_numpy(y0).to(dev)
    cj = torch.zeros(cm.meta.nnzj, dtype=torch.float64, device=dev)
    chs = torch.zeros(cm.meta.nnzh, dtype=torch.float64, device=dev)

    def evaluate():
        cm.jac_coord(x, out=cj)
        cm.hess_coord(x, y, sigma, out=chs)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        evaluate()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        evaluate()
    x1, y1, _ = point(m.meta.x0, m.meta.ncon, seed=11)
    x.copy_(torch.from_numpy(x1))
    y.copy_(torch.from_numpy(y1))
    cj.fill_(float("nan")); chs.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    xe, ye = torch.from_numpy(x1).to(dev), torch.from_numpy(y1).to(dev)
    assert torch.equal(cj, cm.jac_coord(xe)) and torch.equal(chs, cm.hess_coord(xe, ye, sigma))
