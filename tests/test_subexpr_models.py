"""test/NLPTest/subexpr_test.jl replayed without a solver: where the reference solves with Ipopt and compares the
solution with a stated optimum, these tests check that the stated optimum is a stationary point with the stated
objective value (obj, grad from the evaluators only), that reduced subexpressions add no variables/constraints,
and that a model written with subexpressions evaluates identically to the same model written out by hand.
CPU half: the test oracle.  `gpu` half: the HIP path."""
import numpy as np
import pytest

from conftest import has_gpu
from exahip import ExaCore, product, rng

needs_gpu = pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")


def _oracle(core):
    import oracle
    m = oracle.OracleModel(core.to_ir())
    m.x0 = np.asarray(m.meta()[0])
    return m


def _hip(core):
    from exahip import ExaModel
    m = ExaModel(core)
    m.x0 = np.asarray(m.meta.x0)
    m.nvar, m.ncon = m.meta.nvar, m.meta.ncon
    return m


BACKENDS = [pytest.param(_oracle, id="oracle"),
            pytest.param(_hip, id="hip", marks=[pytest.mark.gpu, needs_gpu])]


def concrete_index_constant_term():
    """subexpr_test.jl:8-27: s[i] = x[i] + (i - 1); objective (s[2] + x[i])^2 — a CONCRETE index into s."""
    c = ExaCore()
    x = c.add_var(4, start=[1.0, 2.0, 3.0, 4.0])
    s = c.add_expr(lambda i: x[i] + (i - 1), rng(1, 4))
    c.add_obj(lambda i: (s[2] + x[i]) ** 2, rng(1, 4))
    return c


def basic(with_subexpr):
    """subexpr_test.jl:205-232"""
    c = ExaCore()
    x = c.add_var(10, start=1.0)
    if with_subexpr:
        s = c.add_expr(lambda i: x[i] ** 2, rng(1, 10))
        c.add_obj(lambda i: (s[i] + s[i + 1]) ** 2, rng(1, 9))
        c.add_con(lambda i: s[i] - 1, rng(1, 10), lcon=0.0)
    else:
        c.add_obj(lambda i: (x[i] ** 2 + x[i + 1] ** 2) ** 2, rng(1, 9))
        c.add_con(lambda i: x[i] ** 2 - 1, rng(1, 10), lcon=0.0)
    return c


def multidim(T=3, N=2):
    """subexpr_test.jl:271-300: dx[t,i] = x[t,i] - x[t-1,i] on 0-based 2-D variables."""
    c = ExaCore()
    x = c.add_var(rng(0, T), rng(0, N), start=0.5)
    dx = c.add_expr(lambda p: x[p[0], p[1]] - x[p[0] - 1, p[1]], product(rng(1, T), rng(1, N)))
    c.add_obj(lambda p: dx[p[0], p[1]] ** 2, product(rng(1, T), rng(1, N)))
    c.add_con(lambda i: x[0, i] - 0.0, rng(0, N))
    c.add_con(lambda i: x[T, i] - 1.0, rng(0, N))
    return c, dx


def nested():
    """subexpr_test.jl:305-327: s2 = s1 + s1 = 2 x^2; minimise (s2 - 2)^2 -> x = 1."""
    c = ExaCore()
    x = c.add_var(5, start=1.0, lvar=0.1)
    s1 = c.add_expr(lambda i: x[i] ** 2, rng(1, 5))
    s2 = c.add_expr(lambda i: s1[i] + s1[i], rng(1, 5))
    c.add_obj(lambda i: (s2[i] - 2) ** 2, rng(1, 5))
    return c


def zero_based(T=3):
    """subexpr_test.jl:362-388: V[t] = 2 u[t] + 1 over t in 0:T must not shift to u[t-1]."""
    c = ExaCore()
    u = c.add_var(rng(0, T), start=1.0)
    V = c.add_expr(lambda t: u[t] * 2 + 1, rng(0, T))
    c.add_obj(lambda t: (V[t] - 3) ** 2, rng(0, T))
    return c


def zero_based_nested(T=2, N=2):
    """subexpr_test.jl:393-422"""
    c = ExaCore()
    x = c.add_var(rng(0, T), rng(0, N), start=1.0)
    s1 = c.add_expr(lambda p: x[p[0], p[1]] + 1, product(rng(0, T), rng(0, N)))
    s2 = c.add_expr(lambda p: s1[p[0], p[1]] * 2, product(rng(0, T), rng(0, N)))
    c.add_obj(lambda p: (s2[p[0], p[1]] - 4) ** 2, product(rng(0, T), rng(0, N)))
    return c


@pytest.mark.parametrize("backend", BACKENDS)
def test_concrete_index_constant_term(libs, backend):
    m = backend(concrete_index_constant_term())
    x0 = m.x0
    want = sum((x0[1] + 1 + x0[i]) ** 2 for i in range(4))
    assert abs(m.obj(x0) - want) <= 1e-12 * want
    g = np.array([2 * (x0[1] + 1 + x0[i]) for i in range(4)])
    g[1] += sum(2 * (x0[1] + 1 + x0[i]) for i in range(4))
    np.testing.assert_allclose(m.grad(x0), g, rtol=1e-13)


@pytest.mark.parametrize("backend", BACKENDS)
def test_reduced_equals_written_out(libs, backend):
    m1, m2 = backend(basic(False)), backend(basic(True))
    assert (m1.nvar, m1.ncon) == (m2.nvar, m2.ncon) == (10, 10)
    rs = np.random.default_rng(3)
    x, y = m1.x0 + 0.3 * rs.uniform(-1, 1, 10), rs.standard_normal(10)
    np.testing.assert_allclose(m2.obj(x), m1.obj(x), rtol=1e-13)
    np.testing.assert_allclose(m2.cons(x), m1.cons(x), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(m2.grad(x), m1.grad(x), rtol=1e-13)
    for a, b in zip(m2.jac_structure() + m2.hess_structure(), m1.jac_structure() + m1.hess_structure()):
        assert np.array_equal(a, b)
    np.testing.assert_allclose(m2.jac_coord(x), m1.jac_coord(x), rtol=1e-13)
    np.testing.assert_allclose(m2.hess_coord(x, y, 1.0), m1.hess_coord(x, y, 1.0), rtol=1e-13, atol=1e-13)


def test_multidim_shape(libs):
    c, dx = multidim()
    assert dx.length == 6 and [(r.start, r.stop) for r in dx.size] == [(1, 3), (1, 2)]
    assert (c.nvar, c.ncon) == (12, 6)


@pytest.mark.parametrize("backend", BACKENDS)
def test_multidim_values(libs, backend):
    T, N = 3, 2
    c, _ = multidim(T, N)
    m = backend(c)
    x = np.random.default_rng(8).uniform(-1, 1, (T + 1) * (N + 1))
    X = x.reshape(N + 1, T + 1).T                        # column-major: X[t, i]
    want = sum((X[t, i] - X[t - 1, i]) ** 2 for t in range(1, T + 1) for i in range(1, N + 1))
    np.testing.assert_allclose(m.obj(x), want, rtol=1e-13)
    np.testing.assert_allclose(m.cons(x), np.concatenate([X[0, :], X[T, :] - 1.0]), rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mk,n", [(nested, 5), (zero_based, 4), (zero_based_nested, 9)])
def test_stated_optimum_is_stationary(libs, backend, mk, n):
    m = backend(mk())
    assert (m.nvar, m.ncon) == (n, 0)
    ones = np.ones(n)
    assert abs(m.obj(ones)) <= 1e-28
    assert np.max(np.abs(m.grad(ones))) <= 1e-14
    # ... and a strict one: the objective grows in every coordinate direction
    for k in range(n):
        e = ones.copy()
        e[k] += 1e-3
        assert m.obj(e) > 1e-7
    # off the optimum the gradient is that of the written-out function, not of a shifted index
    x = ones + 0.1 * np.arange(1, n + 1)
    if mk is zero_based:
        np.testing.assert_allclose(m.grad(x), 4 * (2 * x + 1 - 3), rtol=1e-13)
    elif mk is nested:
        np.testing.assert_allclose(m.grad(x), 2 * (2 * x ** 2 - 2) * 4 * x, rtol=1e-13)
    else:
        np.testing.assert_allclose(m.grad(x), 4 * ((x + 1) * 2 - 4), rtol=1e-13)
