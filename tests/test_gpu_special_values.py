"""-m gpu: NaN / Inf / domain-edge propagation.  The reference never raises during evaluation: values outside a
function's domain give NaN or Inf exactly as the closed-form tables say (src/functionlist.jl:6-81, the atanh / acoth
domain guards at :58-59), and NaN / Inf inputs propagate.  Every univariate table entry is evaluated (value, first
and second derivative = cons / jac / hess of c_i = f(x_i)) at special arguments on the HIP path and must agree with
the oracle entry by entry: same NaNs, same signed infinities, finite values to 1e-10."""
import numpy as np
import pytest

from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

SPECIAL = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 1e-300, -1e-300, 1e300, -1e300,
                    1.0000000001, 0.9999999999, 90.0, 180.0, 1e6, 710.0, -745.0])


def _agree(got, ref, what):
    got, ref = np.asarray(got), np.asarray(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN pattern differs\n{got}\n{ref}"
    inf = np.isinf(ref)
    assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], ref[inf]), f"{what}: Inf pattern differs\n{got}\n{ref}"
    fin = np.isfinite(ref)
    # 1e-9: some arguments sit right at a pole (acoth'' at 1 + 1e-10 has condition number 1e10)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-9, atol=1e-300, err_msg=what)


def _model(fn):
    from exahip import ExaCore, rng
    from exahip.graph import Node1
    n = len(SPECIAL)
    c = ExaCore()
    x = c.add_var(n)
    c.add_con(lambda i: Node1(fn, x[i]), rng(1, n))
    c.add_obj(lambda i: Node1(fn, x[i]), rng(1, n))
    return c


def test_every_univariate_at_special_arguments(libs):
    from exahip import ExaModel
    from exahip.graph import SPECIAL_UN, UN_FNS
    import oracle
    y = np.ones(len(SPECIAL))
    bad = []
    # (the SpecialFunctions entries raise DomainError outside their domains in the reference — no NaN convention to keep: they are
    # compared over their domains in tests/test_gpu_special_functions.py)
    for fn in [f for f in UN_FNS if f not in SPECIAL_UN]:
        m = ExaModel(_model(fn))
        o = oracle.OracleModel(m.ir)
        with np.errstate(all="ignore"):
            try:
                _agree(m.cons(SPECIAL), o.cons(SPECIAL), f"{fn}: value")
                _agree(m.jac_coord(SPECIAL), o.jac_coord(SPECIAL), f"{fn}: first derivative")
                _agree(m.hess_coord(SPECIAL, y, 1.0), o.hess_coord(SPECIAL, y, 1.0), f"{fn}: second derivative")
                _agree(m.grad(SPECIAL), o.grad(SPECIAL), f"{fn}: gradient")
                v = np.linspace(0.5, 1.5, len(SPECIAL))
                m.set_product_mode(0, 0)               # the matrix-free sweeps, not the COO-based alternative
                _agree(m.hprod(SPECIAL, y, v, 1.0), o.hprod(SPECIAL, y, v, 1.0), f"{fn}: H*v")
                _agree(m.jtprod(SPECIAL, y), o.jtprod(SPECIAL, y), f"{fn}: J'*v")
                _agree(m.jprod(SPECIAL, v), o.jprod(SPECIAL, v), f"{fn}: J*v")
            except AssertionError as e:
                bad.append(str(e)[:600])
    assert not bad, "\n\n".join(bad)


def test_bivariates_at_special_arguments(libs):
    from exahip import ExaCore, ExaModel, rng
    from exahip.graph import BIN_FNS, Node2
    import oracle
    a = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5, 1e300])
    A, B = (v.ravel() for v in np.meshgrid(a, a))
    n = len(A)
    bad = []
    for fn in [f for f in BIN_FNS if f not in ("beta", "logbeta")]:
        c = ExaCore()
        x = c.add_var(2 * n)
        c.add_con(lambda i: Node2(fn, x[i], x[i + n]), rng(1, n))
        m = ExaModel(c)
        o = oracle.OracleModel(m.ir)
        xs = np.concatenate([A, B])
        if fn == "/":
            # the generated rules for x1 / x2 reuse the quotient: d2 = -(x1/x2)(1/x2), d22 = 2(x1/x2)(1/x2)(1/x2) — one
            # division instead of the table's three ((-x1)/x2^2, (2x1)/x2^3, functionlist.jl:75).  Algebraically equal;
            # they part only where x2^2 or x2^3 overflows (|x2| > 1.3e154: the table yields 0 or NaN, the product form the
            # correctly rounded value).  Documented deviation; those arguments are left out here.
            xs = np.where(np.abs(xs) == 1e300, 3.0, xs)
        with np.errstate(all="ignore"):
            try:
                _agree(m.cons(xs), o.cons(xs), f"{fn}: value")
                _agree(m.jac_coord(xs), o.jac_coord(xs), f"{fn}: first derivatives")
                _agree(m.hess_coord(xs, np.ones(n), 1.0), o.hess_coord(xs, np.ones(n), 1.0), f"{fn}: second derivatives")
            except AssertionError as e:
                bad.append(str(e)[:700])
    assert not bad, "\n\n".join(bad)


def test_degree_functions_are_exact_at_their_zeros_and_poles(libs):
    """Julia's sind / cosd / tand reduce the argument exactly (rem(x, 360)) and pick the quadrant before converting to
    radians: sind(180) == 0, cosd(90) == 0, tand(90) == Inf, cscd(180) == Inf, secd(90) == Inf, cotd(0) == Inf,
    cotd(90) == 0 (true Julia values: Base special/trig.jl).  sin(deg2rad(fmod(x, 360))) gives 1.2e-16, 6e-17, 1.6e16,
    8e15 ... instead.  Both the oracle and the generated code must return the exact values."""
    from exahip import ExaCore, ExaModel, rng
    from exahip.graph import Node1
    import oracle
    cases = {"sind": [(180.0, 0.0), (-180.0, -0.0), (360.0, 0.0), (0.0, 0.0), (90.0, 1.0), (270.0, -1.0), (-90.0, -1.0), (720.0 + 30.0, None)],
             "cosd": [(90.0, 0.0), (270.0, 0.0), (-90.0, 0.0), (180.0, -1.0), (0.0, 1.0), (360.0, 1.0), (60.0, None)],
             "tand": [(90.0, np.inf), (-90.0, -np.inf), (0.0, 0.0), (45.0, 1.0), (270.0, -np.inf)],
             "cscd": [(180.0, np.inf), (90.0, 1.0), (0.0, np.inf)],
             "secd": [(90.0, np.inf), (0.0, 1.0), (180.0, -1.0)],
             "cotd": [(0.0, np.inf), (90.0, 0.0), (45.0, 1.0)]}
    closed = {"sind": lambda t: np.sin(np.deg2rad(t)), "cosd": lambda t: np.cos(np.deg2rad(t))}
    for fn, pts in cases.items():
        xs = np.array([p[0] for p in pts])
        c = ExaCore()
        x = c.add_var(len(xs))
        c.add_con(lambda i: Node1(fn, x[i]), rng(1, len(xs)))
        m = ExaModel(c)
        o = oracle.OracleModel(m.ir)
        with np.errstate(all="ignore"):
            got, ref = m.cons(xs), o.cons(xs)
        for k, (arg, want) in enumerate(pts):
            if want is None:
                want = closed[fn](arg % 360.0)
                assert abs(got[k] - want) < 1e-15 and abs(ref[k] - want) < 1e-15, (fn, arg, got[k], ref[k], want)
                continue
            for who, v in (("hip", got[k]), ("oracle", ref[k])):
                # (== : the sign of a zero is not compared — cons_nln! accumulates onto a zero-filled vector in the
                # reference, which turns sind(-180) = -0.0 into +0.0 there, and a plain store keeps it)
                assert v == want, (who, fn, arg, v, want)


def _tiny_models():
    """(name, core builder, x, reference's value of c / J / H, this library's default value) for the two numeric deviations
    DESIGN.md §4 declares."""
    from exahip import ExaCore, rng
    from exahip.graph import log

    def zero_times_log():
        c = ExaCore()
        x = c.add_var(1)
        c.add_con(lambda i: 0 * log(x[i]) + x[i], rng(1, 1))        # reference: 0 * (-Inf) = NaN at x = 0
        return c

    def plus_zero():
        c = ExaCore()
        x = c.add_var(1)
        c.add_con(lambda i: (x[i] + 0.0) * (x[i] + 0.0), rng(1, 1))
        return c

    def quotient():
        c = ExaCore()
        x = c.add_var(2)
        c.add_con(lambda i: x[i] / x[i + 1], rng(1, 1))
        return c
    return zero_times_log, plus_zero, quotient


def test_the_two_declared_numeric_deviations(libs, monkeypatch):
    """DESIGN.md §4 declares two places where the default build does NOT return the reference's special values; this test
    states them — the reference's value (the oracle restates functionlist.jl / register.jl literally) next to this
    library's — and checks that EXAHIP_STRICT_IEEE=1 removes both.
      1. exact-literal identities: 0*z -> 0, 0/z -> 0 (reference: NaN when z is Inf or NaN), z+0 -> z (reference: +0.0 for
         z = -0.0);
      2. x1/x2 between two variables: partials from the quotient, -(x1/x2)(1/x2) and 2(x1/x2)(1/x2)^2, where the table has
         (-x1)/x2^2 and (2x1)/x2^3 (functionlist.jl:75) — they part where x2^2 or x2^3 over/underflows."""
    from exahip import ExaModel
    import oracle
    zero_times_log, plus_zero, quotient = _tiny_models()
    one = np.ones(1)
    with np.errstate(all="ignore"):
        # ---- 1a. 0 * log(x) + x at x = 0
        x = np.array([0.0])
        for strict in ("0", "1"):
            monkeypatch.setenv("EXAHIP_STRICT_IEEE", strict)
            m = ExaModel(zero_times_log())
            o = oracle.OracleModel(m.ir)
            ref_c, ref_j = o.cons(x)[0], o.jac_coord(x)
            assert np.isnan(ref_c) and np.all(np.isnan(ref_j))                     # the reference: 0 * -Inf, 0 * Inf
            got_c, got_j = m.cons(x)[0], m.jac_coord(x)
            if strict == "1":
                assert np.isnan(got_c) and np.all(np.isnan(got_j))
            else:
                assert got_c == 0.0 and np.nansum(got_j) == 1.0, (got_c, got_j)    # this library: the term is folded away
        # ---- 1b. (x + 0)(x + 0) at x = -0.0: the reference's -0.0 + 0.0 is +0.0, the folded form keeps -0.0
        x = np.array([-0.0])
        for strict in ("0", "1"):
            monkeypatch.setenv("EXAHIP_STRICT_IEEE", strict)
            m = ExaModel(plus_zero())
            o = oracle.OracleModel(m.ir)
            assert not np.signbit(o.jac_coord(x)[0])
            assert np.signbit(m.jac_coord(x)[0]) == (strict == "0")
            assert m.cons(x)[0] == 0.0 == o.cons(x)[0] and m.hess_coord(x, one, 1.0).sum() == o.hess_coord(x, one, 1.0).sum() == 2.0
        # ---- 2. x1 / x2 at |x2| = 1e300 (x2^2 overflows) and 1e-200 (x2^3 underflows)
        for x in (np.array([3.0, 1e300]), np.array([3.0, -1e300]), np.array([1e-10, 1e-200])):
            monkeypatch.setenv("EXAHIP_STRICT_IEEE", "1")
            ms = ExaModel(quotient())
            o = oracle.OracleModel(ms.ir)
            for a, b in ((ms.cons(x), o.cons(x)), (ms.jac_coord(x), o.jac_coord(x)), (ms.hess_coord(x, one, 1.0), o.hess_coord(x, one, 1.0))):
                _agree(a, b, f"strict x1/x2 at {x}")
            monkeypatch.setenv("EXAHIP_STRICT_IEEE", "0")
            md = ExaModel(quotient())
            assert md.cons(x)[0] == o.cons(x)[0]
            jd, jr = md.jac_coord(x), o.jac_coord(x)
            hd, hr = md.hess_coord(x, one, 1.0), o.hess_coord(x, one, 1.0)
            exact_d2 = -(x[0] / x[1]) / x[1]
            if abs(x[1]) == 1e300:
                # reference: (-x1)/x2^2 = -3/Inf = -0.0 and (2x1)/x2^3 = 0.0; this library: the correctly rounded -3e-600 -> -0.0
                # as well (both underflow) — the VALUES agree here, the forms part only in the signed zeros / denormals
                assert jr[1] == 0.0 and jd[1] == 0.0 and np.all(hd[np.isfinite(hr)] == hr[np.isfinite(hr)])
            else:
                # x2 = 1e-200: the reference's x2^3 underflows to 0 -> (2x1)/0 = Inf, the quotient form stays finite/Inf by
                # its own arithmetic; first partial: reference -1e-10/1e-400 -> -Inf, quotient form -(1e190)(1e200) -> -Inf
                assert np.isinf(jr[1]) and jd[1] == exact_d2 or np.isinf(jd[1])
    monkeypatch.delenv("EXAHIP_STRICT_IEEE", raising=False)


def test_mirrored_sincos_arguments_share_one_evaluation_bit_for_bit(libs, monkeypatch):
    """sin / cos of `a - b` where sincos(b - a) is already in the kernel (the two ends of an ACOPF branch,
    test/NLPTest/power.jl:62-78: va_f - va_t and va_t - va_f) are taken from that evaluation: a - b == -(b - a) exactly and
    the sincos is exactly odd / even.  Every output must be bit-identical to the kernels that evaluate both — including
    where the two angles are EQUAL (sin(+0.0) = +0.0, not -0.0)."""
    import torch
    from exahip import ExaModel
    core = ZOO["acopf30"]()
    outs = {}
    for sym in ("1", "0"):
        monkeypatch.setenv("EXAHIP_SYM_TRIG", sym)
        m = ExaModel(core)
        assert ("0.0 - t" in m.kernel_source()) == (sym == "1")
        x, y, s = point(m.meta.x0, m.meta.ncon, seed=9)
        x[:30] = 0.25                                  # all voltage angles equal: every branch has a - b == +0.0
        dev = torch.device("cuda:0")
        xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
        f, c, j, h = m.eval_fused(xd, yd, s)
        got = [m.cons(xd), m.jac_coord(xd), m.hess_coord(xd, yd, s), c, j, h, m.grad(xd)]
        x2, _, _ = point(m.meta.x0, m.meta.ncon, seed=10)
        x2d = torch.from_numpy(x2).to(dev)
        got += [m.cons(x2d), m.jac_coord(x2d), m.hess_coord(x2d, yd, s)]
        torch.cuda.synchronize()
        outs[sym] = [t.cpu().numpy().view(np.int64).copy() for t in got]
    for a, b in zip(outs["1"], outs["0"]):
        assert np.array_equal(a, b)
