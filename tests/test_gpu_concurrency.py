"""-m gpu: distinct model ids are independent (ExaModelsCompiler/test/runtests.jl:299-307; SURVEY §8b threading):
two host threads drive two models on two HIP streams at the same time (ctypes releases the GIL during the calls, so
the library really is entered concurrently) while a third keeps creating and freeing models (registry churn)."""
import threading

import numpy as np
import pytest

from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def test_two_threads_two_models_two_streams(libs):
    import torch
    from exahip import ExaModel
    dev = torch.device("cuda:0")
    jobs = {}
    for name in ("lv1000", "rocket50"):
        m = ExaModel(ZOO[name]())
        x0, y0, s = point(m.meta.x0, m.meta.ncon, seed=21)
        x, y = torch.from_numpy(x0).to(dev), torch.from_numpy(y0).to(dev)
        ref = (m.obj(x), m.cons(x).clone(), m.jac_coord(x).clone(), m.hess_coord(x, y, s).clone())
        jobs[name] = (m, x, y, s, ref)
    torch.cuda.synchronize()
    errors = []
    stop = threading.Event()

    def worker(name):
        m, x, y, s, ref = jobs[name]
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                xs, ys = x.clone(), y.clone()
                for _ in range(150):
                    f = m.obj(xs)
                    c, j, h = m.cons(xs), m.jac_coord(xs), m.hess_coord(xs, ys, s)
                    stream.synchronize()
                    assert f == ref[0] and torch.equal(c, ref[1]) and torch.equal(j, ref[2]) and torch.equal(h, ref[3])
        except Exception as e:          # noqa: BLE001
            errors.append((name, repr(e)))

    def churn():
        try:
            while not stop.is_set():
                t = ExaModel(ZOO["lv20"]())
                assert t.meta.nvar == 20 and t.obj(np.full(20, 2.0)) == 100.0 * 4 * 19 + 19
                del t
        except Exception as e:          # noqa: BLE001
            errors.append(("churn", repr(e)))

    threads = [threading.Thread(target=worker, args=(n,)) for n in jobs] + [threading.Thread(target=churn)]
    for t in threads:
        t.start()
    for t in threads[:2]:
        t.join()
    stop.set()
    threads[2].join()
    assert not errors, errors


def test_concurrent_model_builds(libs):
    """several threads build (plan + generate + load) different models at once; each result is then checked"""
    from exahip import ExaModel
    import oracle
    names = ["lv20", "mixed", "rocket50", "acopf30", "stepped", "conaug2d", "cops_chain", "lv_split_20x2"]
    built, errors = {}, []

    def build(name):
        try:
            for _ in range(3):
                built[name] = ExaModel(ZOO[name]())
        except Exception as e:          # noqa: BLE001
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=build, args=(n,)) for n in names]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for name, m in built.items():
        o = oracle.OracleModel(m.ir)
        x, y, s = point(m.meta.x0, m.meta.ncon, seed=2)
        np.testing.assert_allclose(m.hess_coord(x, y, s), o.hess_coord(x, y, s), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(m.jtprod(x, y), o.jtprod(x, y), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("which", ["lv", "acopf", "rocket"])
def test_obj_is_reproducible_back_to_back(libs, which):
    """obj = per-workgroup partial sums + a fold in a fixed order — by the workgroup of exa_obj that arrives last (up to 512
    workgroups: ACOPF 4, rocket 489; the arrival counter is re-armed by the kernel) or by a second launch (LV 3e6: 1465).
    2000 back-to-back evaluations — interleaved with the fused sweep, which shares the partial-sum buffer — must give one
    and the same bit pattern, and that value must be the CPU sum's to rounding."""
    import ctypes

    import torch
    from exahip import ExaModel, models
    core = (models.luksan_vlcek_model(3_000_000) if which == "lv" else models.rocket_model(1_000_000) if which == "rocket"
            else models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0)))
    m = ExaModel(core)
    L = m._L
    dev = torch.device("cuda:0")
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=3)
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    n = 2000
    f = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    c = torch.empty(m.meta.ncon, dtype=torch.float64, device=dev)
    j = torch.empty(m.meta.nnzj, dtype=torch.float64, device=dev)
    h = torch.empty(m.meta.nnzh, dtype=torch.float64, device=dev)
    p = lambda t, k=0: ctypes.c_void_p(t.data_ptr() + 8 * k)
    for k in range(n):
        if k % 50 == 7:
            assert L.exa_eval_fused(m.id, p(xd), p(yd), ctypes.c_double(s), p(f, k), p(c), p(j), p(h)) == 0
        else:
            assert L.exa_obj_async(m.id, p(xd), p(f, k)) == 0
    torch.cuda.synchronize()
    got = f.cpu().numpy()
    fused = np.arange(n) % 50 == 7          # (another tiling of the data points: its own partial sums, its own last bits)
    for sel in (fused, ~fused):
        bits = got[sel].view(np.int64)
        assert np.all(bits == bits[0]), np.unique(got[sel])
    assert abs(got[7] - got[0]) <= 1e-12 * abs(got[0])
    if which == "lv":
        xs = x
        ref = float(np.sum(100.0 * (xs[:-1] ** 2 - xs[1:]) ** 2 + (xs[:-1] - 1.0) ** 2))
        assert abs(got[0] - ref) <= 1e-12 * abs(ref)
    else:
        import oracle
        ref = oracle.OracleModel(m.ir).obj(x)
        assert abs(got[0] - ref) <= 1e-10 * max(1.0, abs(ref))
