"""The drop-in boundary is a C ABI, not a Python API: a plain-C program (tests/c/abi_client.c) builds the LV pattern
table by hand, links libexahip.so and drives it.  CPU: plan-only mode.  GPU: every callback, compared with the oracle
fed by the PYTHON front-end's table for the same model — so the hand-written C table and the Python lowering agree."""
import os
import subprocess

import numpy as np
import pytest

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "examodels.jl_amd", "exahip")


@pytest.fixture(scope="module")
def client(libs, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cabi") / "abi_client")
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_client.c"),
                           "-o", exe, "-L", LIBDIR, "-lexahip", "-lm", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def run(exe, mode):
    env = dict(os.environ)
    # a C host has no torch: the system ROCm runtime is used
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe, mode], capture_output=True, text=True, env=env, timeout=300)
    return out.returncode, out.stdout + out.stderr


@pytest.mark.parametrize("mode", ["plan", "plan-user"])
def test_c_client_plan_only(client, mode):
    """(`plan-user`: sin and exp registered from C — exa_register_univariate_fused / exa_register_univariate — same plan.)"""
    rc, out = run(client, mode)
    assert rc == 0 and out.strip().endswith("OK"), out
    assert "nvar 10 ncon 8 nnzj 24 nnzh 75" in out
    assert "con o2step 6 comp2 1 1 2 3 1 2 3 1 3 4 2 5 5 1 6 5 6" in out


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("mode", ["eval", "eval-user"])
def test_c_client_full_evaluation(client, mode):
    import oracle
    from exahip import models
    rc, out = run(client, mode)
    assert rc == 0 and out.strip().endswith("OK"), out
    vals = {}
    for line in out.splitlines():
        k, *rest = line.split()
        if k in ("obj", "cons", "grad", "jac", "hess", "hrows", "hcols", "jprod", "jtprod", "hprod", "hprodobj", "hessobj", "jrows", "jcols"):
            vals[k] = np.array([float(v) for v in rest])
    o = oracle.OracleModel(models.luksan_vlcek_model(10).to_ir())
    x0 = o.meta()[0]
    y = 1.0 + 0.1 * np.arange(8)
    np.testing.assert_allclose(vals["obj"][0], o.obj(x0), rtol=1e-12)
    np.testing.assert_allclose(vals["cons"], o.cons(x0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(vals["grad"], o.grad(x0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(vals["jac"], o.jac_coord(x0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(vals["hess"], o.hess_coord(x0, y, 0.5), rtol=1e-12, atol=1e-12)
    r, c = o.hess_structure()
    assert np.array_equal(vals["hrows"].astype(int), r) and np.array_equal(vals["hcols"].astype(int), c)
    v, w, y0 = 0.3 * np.arange(10) - 1.0, 0.5 - 0.2 * np.arange(8), np.zeros(8)
    np.testing.assert_allclose(vals["jprod"], o.jprod(x0, v), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(vals["jtprod"], o.jtprod(x0, w), rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(vals["hprod"], o.hprod(x0, y, v, 0.5), rtol=1e-12, atol=1e-10)
    # the objective-only forms: y == NULL
    np.testing.assert_allclose(vals["hprodobj"], o.hprod(x0, y0, v, 0.5), rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(vals["hessobj"], o.hess_coord(x0, y0, 0.5), rtol=1e-12, atol=1e-12)
    jr, jc = o.jac_structure()
    assert np.array_equal(vals["jrows"].astype(int), jr) and np.array_equal(vals["jcols"].astype(int), jc)
