"""-m gpu: every environment knob that changes the generated code or the path taken has a parity test (DESIGN.md §10 lists
them: knob -> test).  The knobs covered elsewhere: EXAHIP_STRICT_IEEE / EXAHIP_SYM_TRIG (test_gpu_special_values.py),
EXAHIP_HESS_VARIANT / EXAHIP_LDS_MAX (test_gpu_parity.py), EXAHIP_CWINDOW / EXAHIP_CSCATTER (test_gpu_compressed.py),
EXAHIP_PRODUCT_WINDOW (test_gpu_product_windows.py), EXAHIP_COMPILER / EXAHIP_HIPCC / EXAHIP_HIPCC_FLAGS / EXAHIP_CACHE_DIR
(test_plan_and_abi.py, test_gpu_comm.py)."""
import os

import numpy as np
import pytest

from conftest import has_gpu
from zoo import ZOO, point

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _everything(m, x, y, s, v, w):
    import torch
    dev = torch.device("cuda:0")
    xd, yd, vd, wd = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, y, v, w))
    f, g, c, j, h = m.eval_all(xd, yd, s)
    out = {"obj": m.obj(xd), "grad": m.grad(xd), "cons": m.cons(xd), "jac": m.jac_coord(xd), "hess": m.hess_coord(xd, yd, s),
           "jprod": m.jprod(xd, vd), "jtprod": m.jtprod(xd, wd), "hprod": m.hprod(xd, yd, vd, s), "all_c": c, "all_j": j, "all_h": h}
    torch.cuda.synchronize()
    return {k: (t.cpu().numpy().copy() if hasattr(t, "cpu") else t) for k, t in out.items()}


@pytest.mark.parametrize("name", ["acopf30", "rocket50", "lv1000", "mixed"])
def test_group_knob_ungrouped_kernels_give_the_same_bits(libs, name, monkeypatch):
    """EXAHIP_GROUP=0: every pattern evaluated on its own (no fused groups, no chained groups).  Grouping shares loads and
    sincos evaluations, it changes no formula: outputs agree to the last bits (not always bit for bit — the compiler contracts
    multiply-adds differently in a larger body; the scattered products also merge same-target contributions in registers)."""
    from exahip import ExaModel
    import oracle
    m1 = ExaModel(ZOO[name]())
    monkeypatch.setenv("EXAHIP_GROUP", "0")
    m0 = ExaModel(ZOO[name]())
    assert m0._L.exa_module_name(m0.id) != m1._L.exa_module_name(m1.id) or name in ("lv1000", "mixed")       # another module where groups exist
    o = oracle.OracleModel(m1.ir)
    x, y, s = point(m1.meta.x0, m1.meta.ncon, seed=9)
    v = np.random.default_rng(5).standard_normal(m1.meta.nvar)
    w = np.random.default_rng(6).standard_normal(max(1, m1.meta.ncon))[:m1.meta.ncon]
    a, b = _everything(m1, x, y, s, v, w), _everything(m0, x, y, s, v, w)
    for k in ("cons", "jac", "hess", "jprod", "all_c", "all_j", "all_h", "grad", "jtprod", "hprod"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-12, atol=1e-13, err_msg=k)
    np.testing.assert_allclose(b["hess"], o.hess_coord(x, y, s), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(b["hprod"], o.hprod(x, y, v, s), rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("name", ["acopf30", "lv1000"])
def test_fast_trig_knob_ocml_sincos_gives_the_oracle_values_too(libs, name, monkeypatch):
    """EXAHIP_FAST_TRIG=0: ocml's sincos (0.7 ulp) instead of the lean FP64 one (1.3 ulp): both within the 1e-10 bar."""
    from exahip import ExaModel
    import oracle
    monkeypatch.setenv("EXAHIP_FAST_TRIG", "0")
    m = ExaModel(ZOO[name]())
    assert "exa_sincos(t" not in m.kernel_source().split("extern \"C\" __global__ void __launch_bounds__(EXA_BLOCK) exa_obj")[1]
    o = oracle.OracleModel(m.ir)
    x, y, s = point(m.meta.x0, m.meta.ncon, seed=10)
    v = np.random.default_rng(5).standard_normal(m.meta.nvar)
    w = np.random.default_rng(6).standard_normal(m.meta.ncon)
    a = _everything(m, x, y, s, v, w)
    for k, ref in (("grad", o.grad(x)), ("cons", o.cons(x)), ("jac", o.jac_coord(x)), ("hess", o.hess_coord(x, y, s)),
                   ("jtprod", o.jtprod(x, w)), ("hprod", o.hprod(x, y, v, s))):
        np.testing.assert_allclose(a[k], ref, rtol=1e-10, atol=1e-11)


def test_keep_source_and_verbose_knobs_change_nothing_but_leave_evidence(libs, tmp_path, monkeypatch, capfd):
    """EXAHIP_KEEP_SOURCE=1 writes the generated .hip next to the cached code object (both modules of the model);
    EXAHIP_VERBOSE=1 prints one line per decision on stderr.  Neither changes a module's name or a result."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from exahip import ExaModel, models\n"
            "m = ExaModel(models.luksan_vlcek_model(3000))\n"
            "import numpy as np\n"
            "print(m._L.exa_module_name(m.id).decode(), float(np.sum(m.hprod(m.meta.x0, np.ones(m.meta.ncon), np.ones(m.meta.nvar), 0.5))))\n"
            % (os.path.join(root, "examodels.jl_amd"), os.path.join(root, "tests")))
    outs = []
    for env in ({}, {"EXAHIP_KEEP_SOURCE": "1", "EXAHIP_VERBOSE": "1"}):
        d = tmp_path / ("b" if env else "a")
        d.mkdir()
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, EXAHIP_CACHE_DIR=str(d), **env), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((r.stdout.strip().splitlines()[-1], r.stderr, sorted(os.listdir(d))))
    assert outs[0][0] == outs[1][0]                                      # same module, same numbers
    assert not any(f.endswith(".hip") for f in outs[0][2]) and sum(f.endswith(".hip") for f in outs[1][2]) == 2
    assert "[exahip]" not in outs[0][1] and "[exahip] windowed hprod" in outs[1][1]


def test_fast_trig_2_sine_alone_in_value_contexts(libs, monkeypatch):
    """EXAHIP_FAST_TRIG=2: cons_nln! / obj take the sine or the cosine ALONE (exa_sin1 / exa_cos1: reduction modulo pi, one odd polynomial)
    instead of the sincos pair — <= 3 ulp against 40-digit mpmath over small, large and near-multiple-of-pi/2 arguments (the default
    exa_sincos: 1.33, ocml 0.71), i.e. 7e-16 relative against the 1e-10 bar; derivative kernels keep exa_sincos."""
    import mpmath
    from exahip import ExaCore, ExaModel, models, rng
    from exahip.graph import cos, sin
    import oracle
    monkeypatch.setenv("EXAHIP_FAST_TRIG", "2")
    mpmath.mp.dps = 40
    r = np.random.default_rng(0)
    xs = np.concatenate([
        r.uniform(-10, 10, 1500), 10 ** r.uniform(-8, 5.9, 1500) * r.choice([-1, 1], 1500),
        (np.arange(1, 1001) * (np.pi / 2)) * (1 + r.uniform(-1e-12, 1e-12, 1000)),
        np.array([0.0, 1e-300, 823549.0, 823550.0, 1e6, 1e15, 3.0e20]),
    ])
    n = len(xs)
    c = ExaCore()
    x = c.add_var(n)
    c.add_con(lambda i: sin(x[i]), rng(1, n))
    c.add_con(lambda i: cos(x[i]), rng(1, n))
    m = ExaModel(c)
    src = m.kernel_source()
    assert "exa_sin1(" in src.split("exa_cons1(")[-1] and "exa_cos1(" in src
    val, jac = m.cons(xs), m.jac_coord(xs)
    for got, fn, bound in ((val[:n], mpmath.sin, 3.0), (val[n:], mpmath.cos, 3.0), (jac[:n], mpmath.cos, 2.0), (-jac[n:], mpmath.sin, 2.0)):
        worst = 0.0
        for g, xv in zip(got, xs):
            t = fn(mpmath.mpf(float(xv)))
            if t != 0:
                worst = max(worst, float(abs(mpmath.mpf(float(g)) - t) / mpmath.mpf(2) ** (mpmath.floor(mpmath.log(abs(t), 2)) - 52)))
        assert worst <= bound, (fn, worst)
    # and a benchmark model under the knob: every callback within the bar
    lv = ExaModel(models.luksan_vlcek_model(1000))
    o = oracle.OracleModel(lv.ir)
    xx, yy, s = point(lv.meta.x0, lv.meta.ncon, seed=3)
    np.testing.assert_allclose(lv.cons(xx), o.cons(xx), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(lv.hess_coord(xx, yy, s), o.hess_coord(xx, yy, s), rtol=1e-10, atol=1e-12)


def test_fast_exp_is_within_an_ulp_and_the_knob_brings_ocml_back(libs, monkeypatch):
    """The prelude's exa_exp (two-FMA reduction by ln 2, degree-9 near-minimax polynomial, v_ldexp_f64; 24 vector instructions where
    ocml's takes 40): <= 1.5 ulp against 40-digit mpmath over small, wide and boundary arguments, exact special values (0 -> 1,
    overflow -> Inf, underflow -> 0, NaN -> NaN, +-Inf); EXAHIP_FAST_EXP=0 generates ocml's exp instead — both inside the bar on the
    models that use exp."""
    import mpmath
    from exahip import ExaCore, ExaModel, rng
    from exahip.graph import exp
    import oracle
    mpmath.mp.dps = 40
    r = np.random.default_rng(0)
    xs = np.concatenate([
        r.uniform(-1, 1, 1500), r.uniform(-40, 40, 1500), r.uniform(-700, 709.7, 1500),
        (np.arange(-500, 501) + 0.5) * np.log(2) * (1 + r.uniform(-1e-13, 1e-13, 1001)),       # next to the rounding boundaries of k
        #  the subnormal results, exp(x) < 2.2e-308, and the approach to the overflow threshold
        r.uniform(-745.13, -708.39, 2000), r.uniform(709.0, 709.782712893384, 2000),
        np.array([0.0, -0.0, 1e-300, -1e-300, 709.782712893384]),
    ])
    n = len(xs)
    c = ExaCore()
    x = c.add_var(n)
    c.add_con(lambda i: exp(x[i]), rng(1, n))
    m = ExaModel(c)
    assert "exa_exp(" in m.kernel_source().split("exa_powi")[-1]
    val, jac = m.cons(xs), m.jac_coord(xs)
    for got in (val, jac):
        worst = 0.0
        for g, xv in zip(got, xs):
            t = mpmath.exp(mpmath.mpf(float(xv)))
            ulp_exp = max(int(mpmath.floor(mpmath.log(t, 2))) - 52, -1074)          # (subnormal results: the spacing stays 2^-1074)
            worst = max(worst, float(abs(mpmath.mpf(float(g)) - t) / mpmath.mpf(2) ** ulp_exp))
        assert worst <= 1.5, worst
    sp = np.zeros(n)
    sp[:9] = [0.0, 709.79, 1e300, np.inf, -745.2, -1e300, -np.inf, np.nan, -745.0]
    got = m.cons(sp)
    assert got[0] == 1.0 and np.all(np.isinf(got[1:4])) and np.all(got[4:7] == 0.0) and np.isnan(got[7]) and got[8] == 5e-324
    # the knob: ocml's exp in the generated text, the oracle's values either way
    for name in ("rocket50", "lv1000"):
        m1 = ExaModel(ZOO[name]())
        monkeypatch.setenv("EXAHIP_FAST_EXP", "0")
        m0 = ExaModel(ZOO[name]())
        monkeypatch.delenv("EXAHIP_FAST_EXP")
        assert "exa_exp(" in m1.kernel_source().split("exa_powi")[-1] and "exa_exp(" not in m0.kernel_source().split("exa_powi")[-1]
        o = oracle.OracleModel(m1.ir)
        xx, yy, s = point(m1.meta.x0, m1.meta.ncon, seed=12)
        for mm in (m1, m0):
            np.testing.assert_allclose(mm.hess_coord(xx, yy, s), o.hess_coord(xx, yy, s), rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(mm.jac_coord(xx), o.jac_coord(xx), rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(mm.cons(xx), o.cons(xx), rtol=1e-10, atol=1e-12)


def test_hess_throttle_is_the_same_kernel_at_another_occupancy(libs, tmp_path, monkeypatch):
    """EXAHIP_HESS_DYN_LDS=bytes: the chained hess_coord! kernels launched with dynamic LDS nobody uses (three or two workgroups per CU instead
    of as many as the registers allow) — the same kernel, the same bits.  Without the variable exa_tune measures none / three / two next to
    the kernels themselves; its decision is one of the candidates, is persisted with the kernel choice, and a later model starts with it."""
    import torch
    from exahip import ExaModel, models
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))
    N = 2_000_000
    dev = torch.device("cuda:0")
    monkeypatch.setenv("EXAHIP_HESS_VARIANT", "1")
    m0 = ExaModel(models.luksan_vlcek_model(N))
    assert m0._L.exa_hess_variant(m0.id) == 1 and m0._L.exa_hess_throttle(m0.id) == 0
    xd = torch.from_numpy(m0.meta.x0 + 0.05).to(dev)
    yd = torch.ones(m0.meta.ncon, dtype=torch.float64, device=dev)
    ref = m0.hess_coord(xd, yd, 0.5).clone()
    for dyn in (26000, 40000):
        monkeypatch.setenv("EXAHIP_HESS_DYN_LDS", str(dyn))
        m = ExaModel(models.luksan_vlcek_model(N))
        assert m._L.exa_hess_throttle(m.id) == dyn
        assert torch.equal(m.hess_coord(xd, yd, 0.5), ref)
    monkeypatch.delenv("EXAHIP_HESS_DYN_LDS")
    monkeypatch.delenv("EXAHIP_HESS_VARIANT")
    mt = ExaModel(models.luksan_vlcek_model(N))
    mt.tune(1, xd, yd)
    v, d = mt._L.exa_hess_variant(mt.id), mt._L.exa_hess_throttle(mt.id)
    assert v in (0, 1, 2) and (d == 0 if v == 0 else 0 <= d <= 49152)
    assert torch.equal(mt.hess_coord(xd, yd, 0.5), ref) or v == 0        # (the plain kernel contracts its multiply-adds in another body)
    np.testing.assert_allclose(mt.hess_coord(xd, yd, 0.5).cpu().numpy(), ref.cpu().numpy(), rtol=1e-12, atol=1e-300)
    m2 = ExaModel(models.luksan_vlcek_model(N))
    assert (m2._L.exa_hess_variant(m2.id), m2._L.exa_hess_throttle(m2.id)) == (v, d)


@pytest.mark.parametrize("name", sorted(ZOO))       # (round 6: the loops are software-pipelined two-stage code of their own — every zoo model)
def test_tile_loop_kernels_write_the_same_bits(libs, name, monkeypatch):
    """exa_jacl / exa_consl: jac_coord! / cons_nln! as a loop over n consecutive block-map entries per workgroup (launched by default where the
    block map is long: LV 1e7, covered at full size by test_gpu_fullsize.py; EXAHIP_TILE_LOOP=n forces it, 0 switches it off).  The same pattern
    functions on the same tiles: every output bit for bit, into NaN-poisoned buffers, whole and as a 3-way sharded replay."""
    import torch
    from exahip import ExaModel
    dev = torch.device("cuda:0")
    monkeypatch.setenv("EXAHIP_TILE_LOOP", "0")
    m0 = ExaModel(ZOO[name]())
    x, y, s = point(m0.meta.x0, m0.meta.ncon, seed=15)
    xd = torch.from_numpy(x).to(dev)
    ref_c, ref_j = m0.cons(xd).clone(), m0.jac_coord(xd).clone()
    for n in (3, 8):
        monkeypatch.setenv("EXAHIP_TILE_LOOP", str(n))
        m = ExaModel(ZOO[name]())
        c = torch.full((max(1, m.meta.ncon) + 4,), float("nan"), dtype=torch.float64, device=dev)
        j = torch.full((max(1, m.meta.nnzj) + 4,), float("nan"), dtype=torch.float64, device=dev)
        m.cons(xd, out=c); m.jac_coord(xd, out=j)
        torch.cuda.synchronize()
        assert torch.equal(c[:m.meta.ncon], ref_c) and torch.equal(j[:m.meta.nnzj], ref_j)
        assert bool(torch.isnan(c[m.meta.ncon:]).all()) and bool(torch.isnan(j[m.meta.nnzj:]).all())
        whole = np.full(m.meta.nnzj, np.nan)
        for rank in range(3):
            m.set_shard(rank, 3)
            m.set_coo_local(False)
            j.fill_(float("nan"))
            m.jac_coord(xd, out=j)
            torch.cuda.synchronize()
            part = j.cpu().numpy()[:m.meta.nnzj]
            mine = ~np.isnan(part)
            assert np.all(np.isnan(whole[mine]))
            whole[mine] = part[mine]
        assert np.array_equal(whole, ref_j.cpu().numpy())
