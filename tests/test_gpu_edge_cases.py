"""-m gpu: degenerate and boundary-sized models (tests/edgezoo.py) through every entry point of the HIP path, against
the oracle.  Outputs are poisoned with NaN first: each callback must overwrite every entry it owns — including the
zeros of rows / variables nothing contributes to — and must cope with empty outputs."""
import numpy as np
import pytest

from conftest import has_gpu
from edgezoo import EDGE

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]
TOL = dict(rtol=1e-10, atol=1e-12)


@pytest.fixture(scope="module")
def built(libs):
    from exahip import ExaModel
    import oracle
    out = {}
    for name, mk in EDGE.items():
        m = ExaModel(mk())
        out[name] = (m, oracle.OracleModel(m.ir))
    return out


@pytest.mark.parametrize("name", list(EDGE))
def test_all_entry_points(built, name):
    import torch
    m, o = built[name]
    dev = torch.device("cuda:0")
    r = np.random.default_rng(13)
    x = np.asarray(m.meta.x0) + 0.05 * r.uniform(-1, 1, m.meta.nvar)
    y = r.standard_normal(m.meta.ncon)
    v = r.standard_normal(m.meta.nvar)
    w = r.standard_normal(m.meta.ncon)
    assert (m.meta.nvar, m.meta.ncon, m.meta.nnzj, m.meta.nnzh) == (o.nvar, o.ncon, o.nnzj, o.nnzh)
    xd, yd, vd, wd = (torch.from_numpy(a).to(dev) for a in (x, y, v, w))

    def nan(n):
        return torch.full((n,), float("nan"), dtype=torch.float64, device=dev)

    np.testing.assert_allclose(m.obj(xd), o.obj(x), **TOL)
    np.testing.assert_allclose(m.grad(xd, out=nan(m.meta.nvar)).cpu().numpy(), o.grad(x), **TOL)
    np.testing.assert_allclose(m.cons(xd, out=nan(m.meta.ncon)).cpu().numpy(), o.cons(x), **TOL)
    np.testing.assert_allclose(m.jac_coord(xd, out=nan(m.meta.nnzj)).cpu().numpy(), o.jac_coord(x), **TOL)
    np.testing.assert_allclose(m.hess_coord(xd, yd, 0.5, out=nan(m.meta.nnzh)).cpu().numpy(), o.hess_coord(x, y, 0.5), **TOL)
    for (a, b), (ra, rb) in ((m.jac_structure(), o.jac_structure()), (m.hess_structure(), o.hess_structure())):
        assert np.array_equal(a, ra) and np.array_equal(b, rb)
    np.testing.assert_allclose(m.jprod(xd, vd, out=nan(m.meta.ncon)).cpu().numpy(), o.jprod(x, v), **TOL)
    np.testing.assert_allclose(m.jtprod(xd, wd, out=nan(m.meta.nvar)).cpu().numpy(), o.jtprod(x, w), **TOL)
    np.testing.assert_allclose(m.hprod(xd, yd, vd, 0.5, out=nan(m.meta.nvar)).cpu().numpy(), o.hprod(x, y, v, 0.5), **TOL)
    f, c, j, h = m.eval_fused(xd, yd, 0.5, c=nan(m.meta.ncon), jac=nan(m.meta.nnzj), hess=nan(m.meta.nnzh))
    torch.cuda.synchronize()
    np.testing.assert_allclose(f.item(), o.obj(x), **TOL)
    np.testing.assert_allclose(c.cpu().numpy(), o.cons(x), **TOL)
    np.testing.assert_allclose(j.cpu().numpy(), o.jac_coord(x), **TOL)
    np.testing.assert_allclose(h.cpu().numpy(), o.hess_coord(x, y, 0.5), **TOL)
    # host-pointer entry points on the same degenerate shapes
    np.testing.assert_allclose(m.cons(x), o.cons(x), **TOL)
    np.testing.assert_allclose(m.hess_coord(x, y, 1.0), o.hess_coord(x, y, 1.0), **TOL)
    np.testing.assert_allclose(m.grad(x), o.grad(x), **TOL)


@pytest.mark.parametrize("name", ["empty_iterators", "n257", "no_objective"])
def test_sharded_edge_models(built, name):
    """three ranks' partial results of a tiny model (some shards are empty) add up to the unsharded evaluation"""
    import torch
    m, o = built[name]
    dev = torch.device("cuda:0")
    x = np.asarray(m.meta.x0) + 0.02
    y = np.linspace(-1, 1, m.meta.ncon)
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    acc_h = np.zeros(m.meta.nnzh)
    acc_c = np.zeros(m.meta.ncon)
    acc_g = np.zeros(m.meta.nvar)
    f = 0.0
    try:
        for r in range(3):
            m.set_shard(r, 3)
            h = torch.zeros(m.meta.nnzh, dtype=torch.float64, device=dev)
            acc_h += m.hess_coord(xd, yd, 0.5, out=h).cpu().numpy()
            acc_c += m.cons(xd).cpu().numpy()
            acc_g += m.grad(xd).cpu().numpy()
            f += m.obj(xd)
    finally:
        m.set_shard(0, 1)
    np.testing.assert_allclose(acc_h, o.hess_coord(x, y, 0.5), **TOL)
    np.testing.assert_allclose(acc_c, o.cons(x), **TOL)
    np.testing.assert_allclose(acc_g, o.grad(x), **TOL)
    np.testing.assert_allclose(f, o.obj(x), **TOL)
