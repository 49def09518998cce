"""-m gpu: what a rank of a sharded model writes when nothing completes it (no communicator) — SURVEY §8e.  The ranks are
replayed one after the other on the one GPU of the test box into ONE NaN-poisoned buffer per callback:
  * cons / jprod: every rank writes the base rows of its own data points, COMPLETE (the row's owner evaluates the row's
    augmentation terms itself), and nothing else — no zero-fill, no sum;
  * grad of a range-affine objective: every rank writes the variables it owns, complete, from whatever data points touch them;
  * only obj, and grad / J'v / Hv of data-indexed models, are partial sums over the whole vector.
Reference: what is summed on one device in KA ext :273-336."""
import numpy as np
import pytest

from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _close(a, ref):
    np.testing.assert_allclose(a, ref, rtol=1e-10, atol=1e-11)


def _cases():
    from exahip import models
    c = {n: ZOO[n] for n in ("lv1000", "rocket50", "acopf30", "mixed", "conaug2d", "stepped", "cops_chain", "lv_split_20x2")}
    c["lv70001"] = lambda: models.luksan_vlcek_model(70_001)
    return c


@pytest.mark.parametrize("name", sorted(_cases()))
@pytest.mark.parametrize("G", [2, 5])
def test_ranks_write_complete_disjoint_pieces(libs, name, G):
    import torch
    from exahip import ExaModel
    import oracle
    m = ExaModel(_cases()[name]())
    o = oracle.OracleModel(m.ir)
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=3)
    v = np.random.default_rng(5).standard_normal(m.meta.nvar)
    dev = torch.device("cuda:0")
    xd, yd, vd = (torch.from_numpy(a).to(dev) for a in (x, y, v))
    nan = float("nan")
    c = torch.full((max(1, m.meta.ncon),), nan, dtype=torch.float64, device=dev)
    jv = torch.full((max(1, m.meta.ncon),), nan, dtype=torch.float64, device=dev)
    g = torch.full((m.meta.nvar,), nan, dtype=torch.float64, device=dev)
    pull_only = m.shard_layout("grad") == "pieces"
    if name in ("lv1000", "lv70001", "rocket50"):
        assert pull_only                     # range-affine objective with few slots: gathered per variable
    if name in ("acopf30", "mixed"):
        assert not pull_only                 # an objective pattern scatters through a data column
    assert m.shard_layout("cons") == "pieces" and m.shard_layout("obj") == "partial"
    jv_pieces = m.shard_layout("jprod") == "pieces"
    assert jv_pieces == (name not in ("mixed", "conaug2d", "lv_split_20x2"))      # their augmentation terms are not c * x[k]: two-stage Jv
    jvsum = np.zeros(max(1, m.meta.ncon))
    gsum = np.zeros(m.meta.nvar)
    cw = np.zeros(max(1, m.meta.ncon), dtype=int)
    gw = np.zeros(m.meta.nvar, dtype=int)
    f = 0.0
    try:
        for r in range(G):
            m.set_shard(r, G)
            before_c, before_g = torch.isnan(c).cpu().numpy(), torch.isnan(g).cpu().numpy()
            if m.meta.ncon:
                m.cons(xd, out=c)
                if jv_pieces:
                    m.jprod(xd, vd, out=jv)
                else:
                    jvsum += m.jprod(xd, vd).cpu().numpy()
            if pull_only:
                m.grad(xd, out=g)
            else:
                gsum += m.grad(xd).cpu().numpy()
            f += m.obj(xd)
            torch.cuda.synchronize()
            cw += (before_c & ~torch.isnan(c).cpu().numpy()).astype(int)
            gw += (before_g & ~torch.isnan(g).cpu().numpy()).astype(int)
    finally:
        m.set_shard(0, 1)
    assert abs(f - o.obj(x)) <= 1e-10 * max(1.0, abs(o.obj(x)))
    if m.meta.ncon:
        assert cw[:m.meta.ncon].min() == 1 and cw[:m.meta.ncon].max() == 1, name      # every row written by exactly one rank
        _close(c.cpu().numpy()[:m.meta.ncon], o.cons(x))
        _close((jv.cpu().numpy() if jv_pieces else jvsum)[:m.meta.ncon], o.jprod(x, v))
    if pull_only:
        assert gw.min() == 1 and gw.max() == 1, name                                    # every variable by exactly one rank
        _close(g.cpu().numpy(), o.grad(x))
    else:
        _close(gsum, o.grad(x))


def test_allgather_coo_world1_is_a_copy(libs):
    """exa_allgather_coo with one rank: the packed / in-place vector is the whole vector (a device copy when the buffers
    differ).  Between ranks it is an all-gather-v of the slot ranges: tests/test_gpu_dist.py (two ranks, host reducer)."""
    import torch
    from exahip import ExaModel
    m = ExaModel(ZOO["lv1000"]())
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=3)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    h = m.hess_coord(xd, yd, s)
    out = torch.full((m.meta.nnzh,), float("nan"), dtype=torch.float64, device=dev)
    m.allgather_coo(h, hess=True, out=out)
    assert torch.equal(out, h)
    assert m.allgather_coo(h, hess=True).data_ptr() == h.data_ptr()
