"""exahip.matpower: MATPOWER case file -> the ACOPF tables of test/NLPTest/power.jl:31-93 (the data path the reference
takes through PowerModels.jl, restated).  tests/golden/case5_test.m is a five-bus network written for this test."""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_case_file_becomes_the_power_tables(libs):
    from exahip import matpower
    d = matpower.load(os.path.join(HERE, "golden", "case5_test.m"))
    bus, gen, arc, br = d["bus"].cols, d["gen"].cols, d["arc"].cols, d["branch"].cols
    # numbering in file order; bus ids 1, 2, 3, 10, 7 -> 1..5
    assert list(bus["i"]) == [1, 2, 3, 4, 5] and list(d["ref_buses"]) == [1]
    np.testing.assert_allclose(bus["pd"], [0.0, 1.2, 0.8, 0.0, 0.6])
    np.testing.assert_allclose(bus["bs"], [0.0, 0.05, 0.0, 0.0, 0.0])
    np.testing.assert_allclose(bus["gs"], [0.0, 0.0, 0.02, 0.0, 0.0])
    # the generator with status 0 is dropped; costs in per-unit power: c2 * base^2, c1 * base, c0; the linear cost is padded
    assert list(gen["bus"]) == [1, 2, 5] and len(gen["i"]) == 3
    np.testing.assert_allclose(gen["cost1"], [0.02 * 1e4, 0.0, 0.0])
    np.testing.assert_allclose(gen["cost2"], [1400.0, 2000.0, 3000.0])
    np.testing.assert_allclose(gen["cost3"], [100.0, 50.0, 10.0])
    np.testing.assert_allclose(d["pmax"], [2.0, 1.2, 1.5])
    np.testing.assert_allclose(d["qmin"], [-1.5, -0.6, -1.0])
    # the branch with status 0 is dropped: 5 branches, 10 arcs (from sides first)
    assert list(br["f_bus"]) == [1, 1, 2, 4, 3] and list(br["t_bus"]) == [2, 3, 4, 5, 4]
    assert list(br["f_idx"]) == [1, 2, 3, 4, 5] and list(br["t_idx"]) == [6, 7, 8, 9, 10]
    assert list(arc["bus"]) == [1, 1, 2, 4, 3, 2, 3, 4, 5, 4]
    # branch 1 (no transformer): y = 1 / (0.01 + 0.1j); c1 = c3 = -g, c2 = c4 = -b, c5 = c7 = g, c6 = c8 = b + 0.01
    y = 1.0 / complex(0.010, 0.100)
    g, b = y.real, y.imag
    for name, want in (("c1", -g), ("c2", -b), ("c3", -g), ("c4", -b), ("c5", g), ("c6", b + 0.01), ("c7", g), ("c8", b + 0.01)):
        assert math.isclose(br[name][0], want, rel_tol=1e-14), name
    # branch 3: tap 1.05, shift 3 degrees (power.jl:62-78 with tr + j ti = tap exp(j shift))
    y = 1.0 / complex(0.005, 0.050)
    g, b = y.real, y.imag
    tr, ti = 1.05 * math.cos(math.radians(3.0)), 1.05 * math.sin(math.radians(3.0))
    ttm = tr * tr + ti * ti
    for name, want in (("c1", (-g * tr - b * ti) / ttm), ("c2", (-b * tr + g * ti) / ttm), ("c3", (-g * tr + b * ti) / ttm),
                       ("c4", (-b * tr - g * ti) / ttm), ("c5", g / ttm), ("c6", b / ttm), ("c7", g), ("c8", b)):
        assert math.isclose(br[name][2], want, rel_tol=1e-14), name
    # ... and no rating: the bound implied by +-20 degrees and vmax = 1.1 at both ends
    cmax = math.sqrt(1.1 ** 2 + 1.1 ** 2 - 2 * 1.1 * 1.1 * math.cos(math.radians(20.0)))
    assert math.isclose(br["rate_a_sq"][2], (abs(y) * 1.1 * cmax) ** 2, rel_tol=1e-13)
    np.testing.assert_allclose(br["rate_a_sq"][[0, 1, 3, 4]], np.array([2.5, 1.5, 1.8, 1.6]) ** 2)
    np.testing.assert_allclose(d["angmax"], np.deg2rad([30.0, 30.0, 20.0, 30.0, 30.0]))
    np.testing.assert_allclose(d["rate_a_lo"], -d["rate_a"])


def test_the_model_of_a_case_file_evaluates(libs):
    """The tables feed models.ac_power_model unchanged; at a flat start with generation covering the load the power-balance
    rows of a lossless-ish network are small and the oracle's derivatives are consistent (directional finite difference)."""
    import oracle
    from exahip import matpower, models
    core = models.ac_power_model(matpower.load(os.path.join(HERE, "golden", "case5_test.m")))
    ir = core.to_ir()
    o = oracle.OracleModel(ir)
    nbus, ngen, nbr = 5, 3, 5
    assert o.nvar == 2 * nbus + 2 * ngen + 4 * nbr and o.ncon == 1 + 4 * nbr + nbr + 2 * nbr + 2 * nbus
    x = ir.x0 + 0.01 * np.random.default_rng(0).standard_normal(o.nvar)
    y = np.random.default_rng(1).standard_normal(o.ncon)
    u = np.random.default_rng(2).standard_normal(o.nvar)
    eps = 1e-6
    fd = (o.cons(x + eps * u) - o.cons(x - eps * u)) / (2 * eps)
    np.testing.assert_allclose(o.jprod(x, u), fd, rtol=1e-6, atol=1e-6)
    g = lambda z: o.grad(z) + o.jtprod(z, y)          # noqa: E731  gradient of the Lagrangian (sigma = 1)
    fdh = (g(x + eps * u) - g(x - eps * u)) / (2 * eps)
    np.testing.assert_allclose(o.hprod(x, y, u, 1.0), fdh, rtol=1e-5, atol=1e-5)
