"""Every compiled kernel is asked what it needs (exa_build_audit): a module holding a kernel beyond the 256 architectural VGPRs is
built again with the conservative allocator flags.  No device needed: hiprtc / hipcc cross-compile, the resources are read from
the code object's metadata.  (The GPU side — that the fallback build is RIGHT where the default one is wrong — is
tests/test_gpu_poison.py and tests/sweeps/canary/.)"""
import re

import pytest

import randexpr
from exahip import ExaModel, models


def kernels_in(source):
    return {re.search(r"\b(exa_\w+)\(", line).group(1) for line in source.splitlines() if "__global__" in line}


@pytest.fixture()
def fresh_cache(tmp_path, monkeypatch):
    monkeypatch.setenv("EXAHIP_CACHE_DIR", str(tmp_path))
    return tmp_path


@pytest.mark.parametrize("make", [lambda: models.luksan_vlcek_model(200), lambda: models.rocket_model(40),
                                  lambda: models.ac_power_model(models.synthetic_power_data(30, 45, 6, seed=1))], ids=["lv", "rocket", "acopf"])
def test_every_kernel_of_every_module_is_checked(make, fresh_cache):
    m = ExaModel(make(), device=False)
    m.compile()
    audit = m.build_audit()
    for k, which in ((0, "model"), (1, "products")):
        want = kernels_in(m.module_source(k))
        got = {a["kernel"] for a in audit if a["module"] == which}
        assert got == want, (which, want ^ got)
        if k == 0:
            assert {"exa_obj", "exa_fused", "exa_hess", "exa_cons1", "exa_grad", "exa_jtprod", "exa_hprod"} <= got      # the cross-lane kernels VERDICT r3 named
    # the benchmark models live well inside the architectural registers: default flags, nothing over-sized
    assert all(a["fits"] and a["flags"] == "default" for a in audit), [a for a in audit if not a["fits"]]


def test_an_oversized_module_is_rebuilt_with_the_safe_flags_and_keeps_its_windows(fresh_cache):
    # the reproducer of profiles/NOTES.md: exa_hprodw inlines 12 pattern evaluations (256 VGPRs + AGPRs under the default allocator)
    mk = lambda: randexpr.build_range_model(1, npts=1000, unit=True, blocks=True)      # noqa: E731
    m = ExaModel(mk(), device=False)
    m.compile()
    audit = m.build_audit()
    prod = [a for a in audit if a["module"] == "products"]
    assert prod and {a["flags"] for a in prod} == {"safe"}
    assert any(a["kernel"] == "exa_hprodw" and not a["fits"] for a in prod)          # still beyond 256 registers: the flags change the allocation, not the need
    assert {a["kernel"] for a in prod} == kernels_in(m.module_source(1))
    assert m.product_info("hprod")[0] == 2 and m.product_info("jtprod")[0] == 2      # the windows are KEPT (round 3 gave them up)
    names = [n for n, _ in m.code_objects()]
    assert names[1].endswith("_safe")
    # the decision is a note of the module: a later build arrives at the same objects without compiling twice
    m2 = ExaModel(mk(), device=False)
    m2.compile()
    assert [n for n, _ in m2.code_objects()] == names
    assert m2.build_info()[0] == "disk"


def test_the_fallback_can_be_switched_off(fresh_cache, monkeypatch):
    monkeypatch.setenv("EXAHIP_SAFE_FLAGS", "none")
    m = ExaModel(randexpr.build_range_model(1, npts=1000, unit=True, blocks=True), device=False)
    m.compile()
    prod = [a for a in m.build_audit() if a["module"] == "products"]
    assert {a["flags"] for a in prod} == {"default"} and any(not a["fits"] for a in prod)
