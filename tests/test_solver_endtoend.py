"""docs/src/develop.md:84-106 end to end: a Newton-KKT iteration that only ever calls obj / cons / grad /
jac_coord / hess_coord / the two structure callbacks reaches the published Ipopt solution and multipliers of
luksan_vlcek_model(10) from the model's own starting point.  Unlike the stationarity check in
test_known_answers.py this exercises the Hessian (sign of y, obj_weight, lower-triangular layout): a wrong
Hessian convention does not converge quadratically to x*."""
import numpy as np
import pytest

from conftest import has_gpu
from exahip import models
from kktsolve import newton_kkt
from test_known_answers import LSTAR, XSTAR


def _check(m, x0, ncon):
    x, y, its = newton_kkt(m, x0, ncon)
    assert its <= 12
    np.testing.assert_allclose(x, XSTAR, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(y, LSTAR, rtol=1e-5, atol=1e-8)
    assert abs(m.obj(x) - 6.232458632437464) < 1e-9


def test_oracle_newton_reaches_published_point(libs):
    import oracle
    m = oracle.OracleModel(models.luksan_vlcek_model(10).to_ir())
    _check(m, m.meta()[0], m.ncon)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_hip_newton_reaches_published_point(libs):
    from exahip import ExaModel
    m = ExaModel(models.luksan_vlcek_model(10))
    _check(m, m.meta.x0, m.meta.ncon)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_hip_compressed_newton_reaches_published_point(libs):
    """UtilsTest.jl:26-41: the CompressedNLPModel wrapper solves to the same point."""
    import torch
    from exahip import CompressedExaModel, ExaModel
    cm = CompressedExaModel(ExaModel(models.luksan_vlcek_model(10)))
    assert cm.meta.nnzh < cm.inner.meta.nnzh          # duplicates were merged

    class OnDevice:                                    # the compressed wrapper takes device tensors only
        def __getattr__(self, name):
            fn = getattr(cm, name)

            def call(*a):
                a = [torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).cuda() if isinstance(v, np.ndarray) else v
                     for v in a]
                out = fn(*a)
                if isinstance(out, tuple):
                    return tuple(o.cpu().numpy() for o in out)
                return out.cpu().numpy() if torch.is_tensor(out) else out
            return call

    _check(OnDevice(), cm.meta.x0, cm.meta.ncon)
