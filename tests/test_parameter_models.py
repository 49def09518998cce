"""test/NLPTest/parameter_test.jl replayed: closed-form expectations of the two tiny models (:227-264) and
"parametric == non-parametric" for the split Luksan-Vlcek model under every parameter set the reference uses
(:266-365), including the products its `test_function_evaluations` compares (:191-225).
CPU half runs on the test oracle; the `gpu` half runs the HIP path through the C ABI on the same models."""
import numpy as np
import pytest

from conftest import has_gpu
from paramzoo import AFTER_BUILD, DEFAULT_THETA, PARAM_SETS, dense, lv_parametric, param_only, real_only

RTOL = 1e-10


def _oracle(core):
    import oracle
    return oracle.OracleModel(core.to_ir())


def _hip(core):
    from exahip import ExaModel
    return ExaModel(core)


def _sizes(m):
    if callable(m.meta):          # the oracle wrapper
        return m.nvar, m.ncon, m.nnzj, m.nnzh, np.asarray(m.meta()[0])
    return m.meta.nvar, m.meta.ncon, m.meta.nnzj, m.meta.nnzh, np.asarray(m.meta.x0)


def _closed_form(m, theta):
    rs = np.random.default_rng(11)
    x, y = rs.uniform(0, 1, 10), rs.standard_normal(2)
    assert abs(m.obj(x) - np.sum(theta)) <= 1e-14
    np.testing.assert_allclose(m.cons(x), np.asarray(theta) - np.sum((x - 1) ** 2), rtol=1e-13)
    rows, cols = m.jac_structure()
    J = dense(rows, cols, m.jac_coord(x), (2, 10))
    np.testing.assert_allclose(J, np.vstack([-2 * (x - 1)] * 2), rtol=1e-13)
    rows, cols = m.hess_structure()
    H = dense(rows, cols, m.hess_coord(x, y, 1.0), (10, 10), symmetric=True)
    np.testing.assert_allclose(H, np.diag(np.full(10, -2 * np.sum(y))), rtol=1e-13, atol=1e-14)


def _same_evaluations(mp, mn):
    """parameter_test.jl:191-225 at a random point instead of x0 (x0 would hide index errors: it alternates)."""
    sp, sn = _sizes(mp), _sizes(mn)
    assert sp[:4] == sn[:4]
    nvar, ncon, x0 = sp[0], sp[1], sp[4]
    rs = np.random.default_rng(5)
    x = x0 + 0.1 * rs.uniform(-1, 1, nvar)
    y = rs.standard_normal(ncon)
    v = rs.standard_normal(nvar)
    w = rs.standard_normal(ncon)
    tol = dict(rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(mp.obj(x), mn.obj(x), **tol)
    np.testing.assert_allclose(mp.cons(x), mn.cons(x), **tol)
    np.testing.assert_allclose(mp.grad(x), mn.grad(x), **tol)
    np.testing.assert_allclose(mp.jac_coord(x), mn.jac_coord(x), **tol)
    np.testing.assert_allclose(mp.hess_coord(x, y, 0.5), mn.hess_coord(x, y, 0.5), **tol)
    for a, b in zip(mp.jac_structure() + mp.hess_structure(), mn.jac_structure() + mn.hess_structure()):
        assert np.array_equal(a, b)
    np.testing.assert_allclose(mp.jprod(x, v), mn.jprod(x, v), **tol)
    np.testing.assert_allclose(mp.jtprod(x, w), mn.jtprod(x, w), **tol)
    np.testing.assert_allclose(mp.hprod(x, y, v, 1.0), mn.hprod(x, y, v, 1.0), **tol)


# ---- CPU: the oracle ---------------------------------------------------------------------------------------------
def test_oracle_real_only(libs):
    _closed_form(_oracle(real_only()), [1.0, 1.0])


def test_oracle_param_only(libs):
    theta = np.random.default_rng(2).uniform(0, 1, 2)
    _closed_form(_oracle(param_only(theta)), theta)


def test_metadata_counts(libs):
    """parameter_test.jl:272-283"""
    cp, _ = lv_parametric(4, 3, True)
    cn, _ = lv_parametric(4, 3, False)
    assert (cp.nvar, cp.ncon, cp.npar, cn.npar) == (cn.nvar, cn.ncon, 7, 0)
    assert len(cp.patterns) == len(cn.patterns)


@pytest.mark.parametrize("which", list(PARAM_SETS))
def test_oracle_parametric_equals_inlined(libs, which):
    cp, _ = lv_parametric(3, 2, True, PARAM_SETS[which])
    cn, _ = lv_parametric(3, 2, False, PARAM_SETS[which])
    _same_evaluations(_oracle(cp), _oracle(cn))


def test_oracle_modify_after_build(libs):
    cp, th = lv_parametric(3, 2, True)
    mp = _oracle(cp)
    mp.set_value(th.offset, AFTER_BUILD)
    cn, _ = lv_parametric(3, 2, False, AFTER_BUILD)
    _same_evaluations(mp, _oracle(cn))


# ---- GPU: the HIP path -------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_hip_real_only_and_param_only(libs):
    _closed_form(_hip(real_only()), [1.0, 1.0])
    theta = np.random.default_rng(2).uniform(0, 1, 2)
    _closed_form(_hip(param_only(theta)), theta)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("which", list(PARAM_SETS))
def test_hip_parametric_equals_inlined_and_oracle(libs, which):
    cp, _ = lv_parametric(3, 2, True, PARAM_SETS[which])
    cn, _ = lv_parametric(3, 2, False, PARAM_SETS[which])
    mp = _hip(cp)
    _same_evaluations(mp, _hip(cn))
    _same_evaluations(mp, _oracle(cn))


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_hip_modify_after_build(libs):
    """parameter_test.jl:348-365: the model's parameters are its own state after the build."""
    cp, th = lv_parametric(3, 2, True)
    mp = _hip(cp)
    mp.set_value(th, AFTER_BUILD)
    cn, _ = lv_parametric(3, 2, False, AFTER_BUILD)
    _same_evaluations(mp, _hip(cn))
    assert list(np.concatenate(cp.theta)) == DEFAULT_THETA      # the core keeps what it was built with


def test_get_set_value_api(libs):
    """test/GetterSetterTest/GetterSetterTest.jl:20-36 (parameters; the start/bound getters are host metadata)."""
    from exahip import ExaCore, ExaModel, rng
    c = ExaCore()
    x = c.add_var(3, start=1.0, lvar=-2.0, uvar=5.0)
    y = c.add_var(2, start=0.5, lvar=0.0, uvar=1.0)
    t1 = c.add_par([10.0, 20.0, 30.0])
    t2 = c.add_par(2, value=7.0)
    c.add_con(lambda i: x[i] + x[i + 1] + t1[i] * t2[1], rng(1, 2), lcon=-1.0, ucon=1.0, start=0.1)
    c.add_con(y[1] - y[2], lcon=0.0, ucon=0.0, start=0.5)
    m = ExaModel(c, device=False)
    assert list(m.get_value(t1)) == [10.0, 20.0, 30.0] and list(m.get_value(t2)) == [7.0, 7.0]
    m.set_value(t1, [1.0, 2.0, 3.0])
    m.set_value(t2, [9.0, 8.0])
    assert list(m.get_value(t1)) == [1.0, 2.0, 3.0] and list(m.get_value(t2)) == [9.0, 8.0]
    for par, bad in ((t1, [1.0, 2.0]), (t2, [1.0])):
        with pytest.raises(ValueError):
            m.set_value(par, bad)
    assert list(m.meta.x0) == [1.0, 1.0, 1.0, 0.5, 0.5] and list(m.meta.lvar) == [-2.0] * 3 + [0.0] * 2
    assert list(m.meta.uvar) == [5.0] * 3 + [1.0] * 2 and list(m.meta.lcon) == [-1.0, -1.0, 0.0] and list(m.meta.ucon) == [1.0, 1.0, 0.0]
