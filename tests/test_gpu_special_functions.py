"""-m gpu: the SpecialFunctions extension (reference ext/functionlist.jl:6-124; used by the reference's own AD test list,
test/ADTest/ADTest.jl:59-120) on the HIP path, entry by entry over each function's domain: value, first and second derivative
(= cons / jac / hess of c_i = f(x_i)), gradient and the three matrix-free products against the oracle (oracle/exa_special.h:
glibc + long double / __float128 restatements, themselves pinned by tests/golden/special_golden.json at 40 digits).

Tolerance 1e-10 relative, floored at 1e-3 of the largest entry (zeros of the oscillating functions: Bessel, Airy, digamma's root).
The CPU-side counterparts (-m "not gpu"): tests/test_golden_oracle.py (oracle vs mpmath), test_special_functions_plan below."""
import numpy as np
import pytest

from conftest import has_gpu

# per function: the arguments it is evaluated at (inside the domain SpecialFunctions.jl accepts)
_r = np.random.default_rng(7)


def _u(lo, hi, n=48):
    return _r.uniform(lo, hi, n)


DOMAIN = {
    "erf": np.concatenate([_u(-6, 6), [0.0, 1e-300, 30.0, -30.0]]),
    "erfc": np.concatenate([_u(-6, 26), [0.0, 27.0]]),
    "erfi": np.concatenate([_u(-12, 12), [0.0, 5.999, 6.0, 6.001, 20.0]]),
    "erfcx": np.concatenate([_u(-5, 40), [0.0, 1e3, 1e8]]),
    "digamma": np.concatenate([_u(0.01, 40), _u(-6.9, -0.1), [1e-3, 1e4, 1.0, 2.0]]),
    "trigamma": np.concatenate([_u(0.01, 40), _u(-6.9, -0.1), [1e-3, 1e4]]),
    "invdigamma": np.concatenate([_u(-12, 6), [-2.22, -2.2200001, 0.0]]),
    "gamma": np.concatenate([_u(0.05, 30), _u(-5.9, -0.1), [170.0, 1e-5]]),
    "airyai": np.concatenate([_u(-30, 30), [0.0, 8.999, 9.0, 9.001, -8.999, -9.0, -9.001, 60.0]]),
    "airybi": np.concatenate([_u(-30, 30), [0.0, 8.999, 9.0, -9.0, 50.0]]),
    "airyaiprime": np.concatenate([_u(-30, 30), [0.0, 9.0, -9.0]]),
    "airybiprime": np.concatenate([_u(-30, 30), [0.0, 9.0, -9.0]]),
    "besselj0": np.concatenate([_u(-50, 50), [0.0, 1e3]]),
    "bessely0": np.concatenate([_u(0.01, 60), [1e-8, 1e3]]),
    "besselj1": np.concatenate([_u(-50, 50), [0.0, 1e3]]),
    "bessely1": np.concatenate([_u(0.01, 60), [1e-3, 1e3]]),
    # (not beyond |x| ~ 1e2: the table's second derivative -2D - 2x(1 - 2xD), ext/functionlist.jl:89, cancels like 1/x^2 there — at 1e6
    # two correctly rounded D give second derivatives 1e-7 apart: conditioning of the formula, the reference's too)
    "dawson": np.concatenate([_u(-12, 12), [0.0, 6.0, 40.0, -40.0]]),
    "erfinv": np.concatenate([_u(-0.999, 0.999), [0.0, 1 - 1e-12, -1 + 1e-12, 1e-300]]),
    "erfcinv": np.concatenate([_u(0.001, 1.999), [1.0, 1e-300, 1e-20, 2 - 1e-12]]),
}


def close(got, ref, tol, what):
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    assert np.array_equal(np.isfinite(got), np.isfinite(ref)), f"{what}: finite pattern differs\n{got}\n{ref}"
    fin = np.isfinite(ref)
    if not fin.any():
        return 0.0
    scale = np.maximum(np.abs(ref[fin]), 1e-3 * max(1e-300, float(np.max(np.abs(ref[fin])))))
    err = float(np.max(np.abs(got[fin] - ref[fin]) / scale))
    assert err <= tol, f"{what}: {err:.3e} at argument index {int(np.argmax(np.abs(got[fin] - ref[fin]) / scale))}"
    return err


def _un_model(fn, pts):
    from exahip import ExaCore, rng
    from exahip.graph import Node1
    c = ExaCore()
    x = c.add_var(len(pts), start=pts)
    c.add_con(lambda i: Node1(fn, x[i]), rng(1, len(pts)))
    c.add_obj(lambda i: Node1(fn, x[i]), rng(1, len(pts)))
    return c


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("fn", list(DOMAIN))
def test_special_univariate_over_its_domain(libs, fn):
    from exahip import ExaModel
    import oracle
    pts = DOMAIN[fn]
    m = ExaModel(_un_model(fn, pts))
    o = oracle.OracleModel(m.ir)
    y = np.linspace(0.5, 1.5, len(pts))
    v = np.linspace(-1.0, 2.0, len(pts))
    # per-entry comparison (each data point is its own entry: the 1e-3 floor is against the largest of the argument list — which is
    # why the lists do not mix overflowing and tiny values of one function except where stated)
    tol = 1e-10
    # second derivatives of the oscillating / cancelling formulas (ext/functionlist.jl:69-84: (-J0 + J2)/2 ...) inherit the absolute
    # error of ocml's and glibc's Bessel functions near their zeros
    tol2 = 1e-8 if fn.startswith("bessel") else 1e-9
    with np.errstate(all="ignore"):
        close(m.cons(pts), o.cons(pts), tol * (10 if fn.startswith("bessel") else 1), f"{fn}: value")
        close(m.jac_coord(pts), o.jac_coord(pts), tol2, f"{fn}: first derivative")
        close(m.hess_coord(pts, y, 0.7), o.hess_coord(pts, y, 0.7), tol2, f"{fn}: second derivative")
        close(m.grad(pts), o.grad(pts), tol2, f"{fn}: gradient")
        m.set_product_mode(0, 0)
        close(m.hprod(pts, y, v, 0.7), o.hprod(pts, y, v, 0.7), tol2, f"{fn}: H*v")
        close(m.jtprod(pts, y), o.jtprod(pts, y), tol2, f"{fn}: J'*v")
        close(m.jprod(pts, v), o.jprod(pts, v), tol2, f"{fn}: J*v")


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("fn", ["beta", "logbeta"])
def test_beta_and_logbeta_over_a_grid(libs, fn):
    """Both operands variable, and each one fixed in turn (FirstFixed / SecondFixed: register.jl:231-266 take d2/d22 or d1/d11 of the
    same table entry); positive arguments and negative non-integer first arguments (the sign of Gamma)."""
    from exahip import ExaCore, ExaModel, rng
    from exahip.graph import Node2
    import oracle
    a = np.concatenate([_r.uniform(0.05, 25, 40), _r.uniform(-2.9, -2.1, 6), _r.uniform(-0.9, -0.1, 6)])
    b = np.concatenate([_r.uniform(0.05, 25, 40), _r.uniform(3.05, 3.9, 12)])
    n = len(a)
    c = ExaCore()
    x = c.add_var(2 * n, start=np.concatenate([a, b]))
    c.add_con(lambda i: Node2(fn, x[i], x[i + n]), rng(1, n))
    c.add_con(lambda i: Node2(fn, x[i], 2.5) + Node2(fn, 0.75, x[i + n]), rng(1, n))
    c.add_obj(lambda i: Node2(fn, x[i], x[i + n]) * 1e-3, rng(1, n))
    m = ExaModel(c)
    o = oracle.OracleModel(m.ir)
    xs = np.concatenate([a, b])
    y = np.linspace(0.5, 1.5, 2 * n)
    with np.errstate(all="ignore"):
        close(m.cons(xs), o.cons(xs), 1e-10, f"{fn}: value")
        close(m.jac_coord(xs), o.jac_coord(xs), 1e-9, f"{fn}: first derivatives")
        close(m.hess_coord(xs, y, 0.7), o.hess_coord(xs, y, 0.7), 1e-9, f"{fn}: second derivatives")
        close(m.grad(xs), o.grad(xs), 1e-9, f"{fn}: gradient")


def test_special_functions_plan(libs):
    """-m "not gpu": a model that uses the extension plans and generates (its module carries the special prelude and the others do
    not — their cached code objects stay valid), the planner refuses function ids past the table, and the three mirrors of the table
    (include/exahip_ir.h, exahip/graph.py, the Julia shim: tests/test_plan_and_abi.py) list the same names in the same order."""
    import re
    import os
    from exahip import ExaModel, models
    from exahip.graph import BIN_FNS, SPECIAL_UN, UN_FNS
    from zoo import ZOO
    src = ExaModel(ZOO["specialfn"](), device=False).kernel_source()
    assert "exa_polygamma" in src and "exa_airy<" in src and "EXA_INVSQRTPI" in src
    plain = ExaModel(models.luksan_vlcek_model(20), device=False).kernel_source()
    assert "exa_polygamma" not in plain and "EXA_INVSQRTPI" not in plain
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "exahip_ir.h")).read()
    body = re.sub(r"/\*.*?\*/", "", re.search(r"enum exa_un_fn \{(.*?)\}", hdr, re.S).group(1), flags=re.S)
    names = [t.split("=")[0].strip()[len("EXA_U_"):].lower() for t in body.split(",") if t.strip() and "COUNT" not in t]
    alias = {"plus": "+", "minus": "-"}
    assert [alias.get(n, n) for n in names] == UN_FNS
    assert SPECIAL_UN[0] == "erf" and SPECIAL_UN[-1] == "erfcinv" and len(SPECIAL_UN) == 19
    assert BIN_FNS[-2:] == ["beta", "logbeta"]


def test_device_routines_on_the_host_against_mpmath():
    """-m "not gpu": the text of the special prelude compiled for the host WITH contraction (hipcc's default) is within 1e-12 of 40-digit
    mpmath on every routine (tools/special_accuracy.py; measured 7e-14 at worst) — the error-free transformations of the double-double
    Airy series survive -ffp-contract=fast only because they are fenced (`#pragma clang fp contract(off)`): the first GPU run was 9e-7 off."""
    import os
    import sys
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no ROCm clang")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import special_accuracy
    assert special_accuracy.main() == 0
