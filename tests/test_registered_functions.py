"""User-registered functions: exa_register_univariate / exa_register_bivariate (include/exahip.h) — the reference's
`@register_univariate` / `@register_bivariate` (src/register.jl:56-74, 123-276) with the rules as HIP device expressions.

How they are checked without teaching the test oracle new functions: a function registered with the rules of a TABLE entry (`mysin` =
sin, `myhyp` = hypot, ...) must give, inside any tree, the same slot maps as the built-in node (the layout depends on the node TYPE being
univariate / bivariate, never on the function) and the same values to rounding; the oracle evaluates the built-in twin.
CPU: registration semantics, planning, generated text.  `-m gpu`: values through the C ABI."""
import numpy as np
import pytest

from conftest import has_gpu

needs_gpu = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _fns():
    from exahip import graph as G
    mysin = G.register_univariate("mysin", "sin($1)", "cos($1)", "-$2", py=np.sin)
    myexp = G.register_univariate("myexp", "exp($1)", "$2", "$3")
    mycube = G.register_univariate("mycube", "exa_user_cube($1)", "3.0 * $1 * $1", "6.0 * $1",
                                   helpers="static __device__ __forceinline__ double exa_user_cube(double t) { return t * t * t; }")
    myhyp = G.register_bivariate("myhyp", "hypot($1, $2)", "$1 / $3", "$2 / $3", "($3 * $3 - $1 * $1) / ($3 * $3 * $3)",
                                 "-($1 * $2) / ($3 * $3 * $3)", "($3 * $3 - $2 * $2) / ($3 * $3 * $3)")
    mymul = G.register_bivariate("mymul", "$1 * $2", "$2", "$1", "=0", "=1", "=0")
    return mysin, myexp, mycube, myhyp, mymul


def _pair(n=60):
    """The same model twice: with registered functions and with their built-in twins."""
    from exahip import ExaCore, Table, graph as G, rng
    mysin, myexp, mycube, myhyp, mymul = _fns()

    def build(sin, exp, cube, hyp, mul):
        c = ExaCore()
        x = c.add_var(n, start=np.linspace(0.4, 1.4, n))
        th = c.add_par(2, value=[1.5, 0.25])
        tab = Table(i=np.arange(1, n - 1, 2), j=np.arange(3, n + 1, 2)[: len(np.arange(1, n - 1, 2))], w=np.linspace(0.5, 2.0, len(np.arange(1, n - 1, 2))))
        c.add_obj(lambda i: sin(x[i] - x[i + 1]) * exp(-x[i]) + cube(x[i + 1]) + hyp(x[i], x[i + 1] * th[1]), rng(1, n - 1))
        g = c.add_con(lambda i: mul(sin(x[i]), exp(x[i + 1] * 0.5)) + hyp(x[i + 2], 2.0) + hyp(th[2], cube(x[i])) + mul(3.0, x[i]) * x[i + 1], rng(1, n - 2))
        c.add_con(lambda d: d.w * sin(x[d.i] * x[d.j]) + cube(exp(x[d.i]) - x[d.j]), tab)
        c.add_con_aug(g, lambda k: (k, mul(x[k], x[k + 4]) + sin(x[k + 2])), rng(1, 5))
        return c

    user = build(mysin, myexp, mycube, myhyp, mymul)
    twin = build(G.sin, G.exp, lambda t: t ** 3, G.hypot, lambda a, b: a * b)
    return user, twin


def test_registration_semantics(libs):
    from exahip import capi, graph as G
    L = capi.lib()
    a = L.exa_register_univariate(b"reg_twice", b"tanh($1)", b"1.0 - $2 * $2", b"-2.0 * $2 * $3", None)
    assert a >= 1000 and a == L.exa_register_univariate(b"reg_twice", b"tanh($1)", b"1.0 - $2 * $2", b"-2.0 * $2 * $3", None)   # same rules: same id
    assert L.exa_register_univariate(b"reg_twice", b"tanh($1)", b"1.0", b"=0", None) == -1 and b"already registered" in L.exa_last_error()
    assert L.exa_register_univariate(b"bad name", b"$1", b"=1", b"=0", None) == -1 and b"identifier" in L.exa_last_error()
    assert L.exa_register_univariate(b"needs_rule", b"$1", b"", b"=0", None) == -1 and b"needs an expression" in L.exa_last_error()
    assert L.exa_register_univariate(b"bad_ph", b"$1 + $2", b"=1", b"=0", None) == -1 and b"placeholder" in L.exa_last_error()   # f has no $2
    assert L.exa_register_bivariate(b"bad_ph2", b"$1 * $2", b"$4", b"$1", b"=0", b"=1", b"=0", None) == -1
    assert L.exa_register_univariate(None, b"$1", b"=1", b"=0", None) == -1
    b = L.exa_register_bivariate(b"reg_bin", b"$1 * $2", b"$2", b"$1", b"=0", b"=1", b"=0", None)
    assert b >= 1000
    with pytest.raises(ValueError):
        G.register_univariate("reg_twice", "tanh($1)", "1.0", "=0")


def test_unregistered_ids_are_refused_and_registered_ones_plan_like_their_twins(libs):
    from exahip import ExaModel
    import ctypes
    user, twin = _pair()
    mu, mt = ExaModel(user, device=False), ExaModel(twin, device=False)
    assert (mu.meta.nvar, mu.meta.ncon, mu.meta.nnzj, mu.meta.nnzh) == (mt.meta.nvar, mt.meta.ncon, mt.meta.nnzj, mt.meta.nnzh)
    for k in range(4):
        a, b = mu.pattern_info(k), mt.pattern_info(k)
        assert a == b, (k, a, b)
    src = mu.kernel_source()
    assert "exa_user_cube" in src and "user-registered function `mycube`" in src and "hypot(" in src
    assert "exa_user_cube" not in mt.kernel_source()
    mu.compile()                                # the rules compile for gfx950 (hipcc / hiprtc, no device needed)
    # an id nobody registered
    from exahip import ExaCore, graph as G, rng
    G.UN_ID["ghost"] = 4999
    try:
        c = ExaCore()
        x = c.add_var(4)
        c.add_obj(lambda i: G.Node1("ghost", x[i]), rng(1, 4))
        with pytest.raises(Exception, match="unknown univariate function"):
            ExaModel(c, device=False)
    finally:
        del G.UN_ID["ghost"]


def test_a_rule_that_does_not_compile_fails_the_build_with_the_compiler_message(libs):
    from exahip import ExaCore, ExaModel, graph as G, rng
    broken = G.register_univariate("broken_rule", "no_such_function($1)", "=1", "=0")
    c = ExaCore()
    x = c.add_var(4)
    c.add_obj(lambda i: broken(x[i]), rng(1, 4))
    m = ExaModel(c, device=False)
    with pytest.raises(Exception, match="no_such_function"):
        m.compile()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_registered_functions_equal_their_builtin_twins_on_hip(libs):
    from exahip import ExaModel
    import oracle
    user, twin = _pair()
    m = ExaModel(user)
    o = oracle.OracleModel(ExaModel(twin, device=False).ir)        # the oracle never sees a user function: it evaluates the twin
    x = np.asarray(m.meta.x0) + 0.05 * np.random.default_rng(0).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(3).standard_normal(m.meta.ncon)
    for a, b in zip(m.jac_structure() + m.hess_structure(), o.jac_structure() + o.hess_structure()):
        assert np.array_equal(a, b)

    def rel(a, b):
        a, b = np.asarray(a, float), np.asarray(b, float)
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * max(1.0, float(np.max(np.abs(b)))))))

    assert abs(m.obj(x) - o.obj(x)) <= 1e-12 * abs(o.obj(x))
    for name, a, b in (("cons", m.cons(x), o.cons(x)), ("grad", m.grad(x), o.grad(x)), ("jac", m.jac_coord(x), o.jac_coord(x)),
                       ("hess", m.hess_coord(x, y, 0.7), o.hess_coord(x, y, 0.7)), ("jprod", m.jprod(x, v), o.jprod(x, v)),
                       ("jtprod", m.jtprod(x, w), o.jtprod(x, w)), ("hprod", m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7))):
        assert rel(a, b) <= 1e-12, (name, rel(a, b))
    # the fused sweep and the compressed COO go through the same rules
    import torch
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    f, g, c, jac, hess = m.eval_all(xd, yd, 0.7)
    torch.cuda.synchronize()
    assert rel(hess.cpu().numpy(), o.hess_coord(x, y, 0.7)) <= 1e-12 and rel(g.cpu().numpy(), o.grad(x)) <= 1e-12
    from exahip import CompressedExaModel
    ch = CompressedExaModel(m).hess_coord(xd, yd, 0.7)          # duplicate-summed COO: same total
    torch.cuda.synchronize()
    assert abs(float(ch.sum()) - float(o.hess_coord(x, y, 0.7).sum())) <= 1e-10 * float(np.abs(o.hess_coord(x, y, 0.7)).sum())
