"""Golden fixtures of the whole zoo + Luksan-Vlcek N = 1e4 (BASELINE.json configs[0]): tests/golden/zoo_fixtures/, written
by tests/golden/make_zoo_fixtures.py from the oracle.  The same files are what tools/reference_check.jl compares the REAL
reference (ExaModels v0.12, backend = nothing) with on a machine that has Julia — structure ==, values 1e-10 — so a
green run of this file plus a green run of that script closes the chain  reference == fixtures == HIP path.

  not gpu: the oracle still reproduces the committed fixtures (they are a pin of the restatement, not a moving target);
  gpu:     the HIP path through the C ABI reproduces them, all seven callbacks + the three products."""
import os
import sys

import numpy as np
import pytest

from conftest import has_gpu, parity, parity_cons

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_zoo_fixtures as fx  # noqa: E402

NAMES = sorted(fx.models())
RTOL = 1e-10


def close(a, ref, what):
    a, ref = np.asarray(a), np.asarray(ref)
    assert a.shape == ref.shape, what
    if ref.size:
        scale = np.maximum(np.abs(ref), 1e-3 * max(1.0, float(np.max(np.abs(ref)))))
        err = float(np.max(np.abs(a - ref) / scale))
        assert err <= RTOL, (what, err)


def test_fixture_set_is_complete():
    for n in NAMES:
        assert os.path.exists(os.path.join(fx.OUT, n + ".json")) and os.path.exists(os.path.join(fx.OUT, n + ".bin")), n
    sc, _ = fx.load("lv10000")
    assert (sc["nvar"], sc["ncon"], sc["nnzj"], sc["nnzh"]) == (10_000, 9_998, 29_994, 89_985)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_fixtures(libs, name):
    import oracle
    sc, a = fx.load(name)
    o = oracle.OracleModel(fx.models()[name]().to_ir())
    assert (o.nvar, o.ncon, o.nnzj, o.nnzh) == (sc["nvar"], sc["ncon"], sc["nnzj"], sc["nnzh"])
    x, y, s = a["x"], a["y"], sc["sigma"]
    jr, jc = o.jac_structure()
    hr, hc = o.hess_structure()
    assert np.array_equal(jr, a["jac_rows"]) and np.array_equal(jc, a["jac_cols"])
    assert np.array_equal(hr, a["hess_rows"]) and np.array_equal(hc, a["hess_cols"])
    assert o.obj(x) == sc["obj"]
    for got, key in ((o.cons(x), "cons"), (o.grad(x), "grad"), (o.jac_coord(x), "jac_vals"), (o.hess_coord(x, y, s), "hess_vals"),
                     (o.jprod(x, a["u"]), "jprod"), (o.jtprod(x, a["v"]), "jtprod"), (o.hprod(x, y, a["u"], s), "hprod")):
        assert np.array_equal(got, a[key]), key           # same code, same machine arithmetic: bit for bit


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_reproduces_the_fixtures(libs, name):
    from exahip import ExaModel
    sc, a = fx.load(name)
    m = ExaModel(fx.models()[name]())
    assert (m.meta.nvar, m.meta.ncon, m.meta.nnzj, m.meta.nnzh) == (sc["nvar"], sc["ncon"], sc["nnzj"], sc["nnzh"])
    x, y, s = a["x"], a["y"], sc["sigma"]
    jr, jc = m.jac_structure(dtype=np.int32)
    hr, hc = m.hess_structure(dtype=np.int32)
    assert np.array_equal(jr, a["jac_rows"]) and np.array_equal(jc, a["jac_cols"])
    assert np.array_equal(hr, a["hess_rows"]) and np.array_equal(hc, a["hess_cols"])
    assert abs(m.obj(x) - sc["obj"]) <= RTOL * max(1.0, abs(sc["obj"]))
    # strict = component-wise |a - ref| / |ref| <= 1e-10 (north_star's bar) on the four value callbacks; cons_nln! through the
    # __float128 arbiter for rows that cancel (conftest.parity_cons); the products are sums over a row / column that may cancel
    # as a whole: floored measure asserted, strict one reported
    import oracle
    o = oracle.OracleModel(fx.models()[name]().to_ir())
    close(m.cons(x), a["cons"], "cons")
    parity_cons("fixture:" + name, m.cons(x), o, x, RTOL)
    parity("fixture:" + name, "grad", m.grad(x), a["grad"], RTOL, strict_rtol=RTOL)
    parity("fixture:" + name, "jac", m.jac_coord(x), a["jac_vals"], RTOL, strict_rtol=RTOL)
    parity("fixture:" + name, "hess", m.hess_coord(x, y, s), a["hess_vals"], RTOL, strict_rtol=RTOL)
    parity("fixture:" + name, "jprod", m.jprod(x, a["u"]), a["jprod"], RTOL)
    parity("fixture:" + name, "jtprod", m.jtprod(x, a["v"]), a["jtprod"], RTOL)
    parity("fixture:" + name, "hprod", m.hprod(x, y, a["u"], s), a["hprod"], RTOL)


# ---- fixtures written by the REAL reference (tools/reference_check.jl --dump tests/golden/zoo_fixtures_reference) ---------------
REF_DIR = os.path.join(HERE, "golden", "zoo_fixtures_reference")
REF_NAMES = sorted(f[:-5] for f in os.listdir(REF_DIR) if f.endswith(".json")) if os.path.isdir(REF_DIR) else []


def _load_ref(name):
    import json
    with open(os.path.join(REF_DIR, name + ".json")) as fh:
        man = json.load(fh)
    raw = open(os.path.join(REF_DIR, name + ".bin"), "rb").read()
    return man["scalars"], {k: np.frombuffer(raw, dtype="<" + dt, count=n, offset=off) for k, (dt, n, off) in man["arrays"].items()}


def test_reference_check_kit_rebuilds_every_fixture_model():
    """tools/reference_check.jl states, with the reference's own macros, every model the fixtures hold (static check: no
    Julia in the build container), and knows the --dump mode that turns its outputs into committed fixtures."""
    jl = open(os.path.join(os.path.dirname(HERE), "tools", "reference_check.jl")).read()
    import re
    listed = set(re.findall(r'\("(\w+)", a ->', jl))
    assert listed == set(NAMES), (sorted(set(NAMES) - listed), sorted(listed - set(NAMES)))
    assert "--dump" in jl and "zoo_fixtures_reference" in jl and "function dump_fixture" in jl


@pytest.mark.skipif(not REF_NAMES, reason="no reference-generated fixtures committed yet (needs a machine with Julia: tools/reference_check.jl --dump)")
@pytest.mark.parametrize("name", REF_NAMES or ["-"])
def test_oracle_equals_what_the_reference_computed(libs, name):
    """Structure == and values 1e-10 against a Julia process's outputs: the Hessian slot order pinned by the reference itself."""
    import oracle
    sc, a = _load_ref(name)
    o = oracle.OracleModel(fx.models()[name]().to_ir())
    x, y, s = a["x"], a["y"], sc["sigma"]
    jr, jc = o.jac_structure()
    hr, hc = o.hess_structure()
    assert np.array_equal(jr, a["jac_rows"]) and np.array_equal(jc, a["jac_cols"])
    assert np.array_equal(hr, a["hess_rows"]) and np.array_equal(hc, a["hess_cols"])
    assert abs(o.obj(x) - sc["obj"]) <= RTOL * max(1.0, abs(sc["obj"]))
    for got, key in ((o.cons(x), "cons"), (o.grad(x), "grad"), (o.jac_coord(x), "jac_vals"), (o.hess_coord(x, y, s), "hess_vals"),
                     (o.jprod(x, a["u"]), "jprod"), (o.jtprod(x, a["v"]), "jtprod"), (o.hprod(x, y, a["u"], s), "hprod")):
        close(got, a[key], key)
