"""Known-answer tests the reference's own suite holds for this path, replayed on the test oracle (CPU).

Each test cites the reference test whose expected values it restates.  The same model builders are evaluated by
the HIP path in tests/test_gpu_known_answers.py.
"""
import numpy as np
import pytest

from exahip import ExaCore, Table, graph, models, product, rng
from exahip.graph import Constant


def oracle_model(core):
    import oracle
    return oracle.OracleModel(core.to_ir())


# ---- model builders (shared with the GPU test) -----------------------------------------------------------------
def conaug_1d_sugar(N=9):
    """conaug_test.jl:37-48: `@add_con(c, g, Constant(0) for _ = 1:N)`, `@add_con!(c, g[i] += x[i] + x[i+1])`; x = ones -> all 2."""
    c = ExaCore()
    x = c.add_var(N + 1)
    g = c.add_con(lambda _: Constant(0), rng(1, N), lcon=-1.0, ucon=1.0)
    c.add_con_aug(None, lambda i: g[i] + (x[i] + x[i + 1]), rng(1, N))
    return c, np.ones(N + 1), np.full(N, 2.0)


def conaug_2d_int(N=3, M=4):
    """conaug_test.jl:73-95"""
    c = ExaCore()
    x = c.add_var(N, M)
    g = c.add_con(N, M, lcon=-np.inf, ucon=0.0)
    itr = [(i, j) for j in range(1, M + 1) for i in range(1, N)]
    c.add_con_aug(None, lambda p: g[p[0], p[1]] + (x[p[0], p[1]] - x[p[0] + 1, p[1]]), itr)
    x0 = np.arange(1.0, N * M + 1)
    exp = np.array([(float((j - 1) * N + i) - float((j - 1) * N + i + 1)) if i < N else 0.0
                    for j in range(1, M + 1) for i in range(1, N + 1)])
    return c, x0, exp, len(itr) * 2


def conaug_2d_range(N=3, M=4):
    """conaug_test.jl:97-123: range-based dims with a non-unit start (r2 = 2:M+1)."""
    c = ExaCore()
    x = c.add_var(N, M + 1)
    g = c.add_con(rng(1, N), rng(2, M + 1), lcon=-np.inf, ucon=0.0)
    itr = [(i, j) for j in range(2, M + 2) for i in range(1, N)]
    c.add_con_aug(None, lambda p: g[p[0], p[1]] + (x[p[0], p[1]] - x[p[0] + 1, p[1]]), itr)
    x0 = np.arange(1.0, N * (M + 1) + 1)
    exp = np.array([-1.0 if i < N else 0.0 for _ in range(1, M + 1) for i in range(1, N + 1)])
    return c, x0, exp, len(itr) * 2


def conaug_3d(N=2, M=3, K=4):
    """conaug_test.jl:125-143"""
    c = ExaCore()
    x = c.add_var(N * M * K)
    g = c.add_con(N, M, K, lcon=0.0, ucon=0.0)
    itr = [(i, j, k) for k in range(1, K + 1) for j in range(1, M + 1) for i in range(1, N + 1)]
    c.add_con_aug(None, lambda p: g[p[0], p[1], p[2]] + x[(p[2] - 1) * N * M + (p[1] - 1) * N + p[0]] * 2, itr)
    return c, np.ones(N * M * K), np.full(N * M * K, 2.0), len(itr)


def conaug_multi(N=4, M=5):
    """conaug_test.jl:180-214: forward and backward differences on one 2-D constraint."""
    c = ExaCore()
    x = c.add_var(N, M)
    g = c.add_con(N, M, lcon=-np.inf, ucon=np.inf)
    fwd = [(i, j) for j in range(1, M + 1) for i in range(1, N)]
    bwd = [(i, j) for j in range(1, M + 1) for i in range(2, N + 1)]
    c.add_con_aug(None, lambda p: g[p[0], p[1]] + (x[p[0], p[1]] - x[p[0] + 1, p[1]]), fwd)
    c.add_con_aug(None, lambda p: g[p[0], p[1]] + (x[p[0] - 1, p[1]] - x[p[0], p[1]]), bwd)
    x0 = np.arange(1.0, N * M + 1)
    exp = []
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            b = (j - 1) * N
            exp.append(float(b + 1) - float(b + 2) if i == 1 else (float(b + N - 1) - float(b + N) if i == N else float(b + i - 1) - float(b + i + 1)))
    return c, x0, np.array(exp), (len(fwd) + len(bwd)) * 2


def par_nonunit():
    """feature_test.jl:100-112: add_par(core, 2:4; value=[10,20,30]); θ[j]*x[1] for j in 2:4; x = ones."""
    c = ExaCore()
    x = c.add_var(5)
    th = c.add_par(rng(2, 4), value=[10.0, 20.0, 30.0])
    c.add_con(lambda j: th[j] * x[1], rng(2, 4))
    return c, np.ones(5), np.array([10.0, 20.0, 30.0])


def par_multidim():
    """feature_test.jl:114-126: add_par(core, 3, 2:5; value=1:12); θ[i,j]*x[1] for (1,2),(2,3),(3,4)."""
    c = ExaCore()
    x = c.add_var(12)
    th = c.add_par(3, rng(2, 5), value=np.arange(1.0, 13.0))
    c.add_con(lambda p: th[p[0], p[1]] * x[1], [(1, 2), (2, 3), (3, 4)])
    return c, np.ones(12), np.array([1.0, 5.0, 9.0])


def par_set_value():
    """feature_test.jl:159-168: set_value! before the model is built."""
    c = ExaCore()
    th = c.add_par(rng(2, 4), value=np.ones(3))
    c.set_value(th, [5.0, 6.0, 7.0])
    x = c.add_var(1)
    c.add_con(lambda j: th[j] * x[1], rng(2, 4))
    return c, np.ones(1), np.array([5.0, 6.0, 7.0])


def expr_nonunit():
    """feature_test.jl:171-182: add_expr(x[i]^2 for i in 2:4); s[j] for j in 2:4; x = 1:5 -> [4,9,16]."""
    c = ExaCore()
    x = c.add_var(5)
    s = c.add_expr(lambda i: x[i] ** 2, rng(2, 4))
    c.add_con(lambda j: s[j], rng(2, 4))
    return c, np.arange(1.0, 6.0), np.array([4.0, 9.0, 16.0])


def data_axis():
    """feature_test.jl:128-157 (zip axis): y[i] - v for (i, v) in zip(2:4, [5,45,0]); y = 0."""
    c = ExaCore()
    y = c.add_var(6, start=0.0)
    c.add_con(lambda p: y[p[0]] - p[1], [(2, 5.0), (3, 45.0), (4, 0.0)])
    return c, np.zeros(6), np.array([-5.0, -45.0, 0.0])


CONS_CASES = {
    "conaug_1d_sugar": lambda: conaug_1d_sugar()[:3],
    "conaug_2d_int": lambda: conaug_2d_int()[:3],
    "conaug_2d_range": lambda: conaug_2d_range()[:3],
    "conaug_3d": lambda: conaug_3d()[:3],
    "conaug_multi": lambda: conaug_multi()[:3],
    "par_nonunit": par_nonunit,
    "par_multidim": par_multidim,
    "par_set_value": par_set_value,
    "expr_nonunit": expr_nonunit,
    "data_axis": data_axis,
}


@pytest.mark.parametrize("name", list(CONS_CASES))
def test_cons_known_answers(libs, name):
    core, x0, expected = CONS_CASES[name]()
    o = oracle_model(core)
    assert o.ncon == len(expected)
    np.testing.assert_allclose(o.cons(x0), expected, rtol=0, atol=1e-14)


def test_conaug_nnzj_counts(libs):
    for mk in (conaug_2d_int, conaug_2d_range, conaug_3d, conaug_multi):
        core, _, exp, nnzj = mk()
        o = oracle_model(core)
        assert o.nnzj == nnzj and o.ncon == len(exp)


def test_conaug_old_and_sugar_syntax_agree(libs):
    """conaug_test.jl:11-35"""
    N = 9
    c1 = ExaCore()
    x1 = c1.add_var(N + 1)
    g1 = c1.add_con(N, lcon=-1.0, ucon=1.0)
    c1.add_con_aug(g1, lambda i: (i, x1[i] + x1[i + 1]), rng(1, N))
    c2 = ExaCore()
    x2 = c2.add_var(N + 1)
    g2 = c2.add_con(N, lcon=-1.0, ucon=1.0)
    c2.add_con_aug(None, lambda i: g2[i] + (x2[i] + x2[i + 1]), rng(1, N))
    o1, o2 = oracle_model(c1), oracle_model(c2)
    assert (o1.nnzj, o1.nnzh) == (o2.nnzj, o2.nnzh)
    x0 = np.random.default_rng(0).uniform(size=N + 1)
    np.testing.assert_array_equal(o1.cons(x0), o2.cons(x0))


def test_concrete_mode_known_objective(libs):
    """test/ConcreteModeTest.jl:58-64: sum (x_i - 1)^2 at x = 1.5, N = 4 -> 1.0;
    ExaModelsCompiler/test/runtests.jl:305-306 analogue: obj at ones of sum x_i^2 is N."""
    c = ExaCore()
    x = c.add_var(4, start=1.5)
    c.add_obj(lambda i: (x[i] - 1) ** 2, rng(1, 4))
    o = oracle_model(c)
    x0 = o.meta()[0]
    assert o.obj(x0) == 1.0
    c = ExaCore()
    x = c.add_var(7)
    c.add_obj(lambda i: x[i] ** 2, rng(1, 7))
    assert oracle_model(c).obj(np.ones(7)) == 7.0


# ---- Luksan-Vlcek: layout tables and the published KKT point ---------------------------------------------------
XSTAR = np.array([-0.9505563573613093, 0.9139008176388945, 0.9890905176644905, 0.9985592422681151, 0.9998087408802769,
                  0.9999745932450963, 0.9999966246997642, 0.9999995512524277, 0.999999944919307, 0.999999930070643])
LSTAR = np.array([4.1358568305002255, -1.876494903703342, -0.06556333356358675, -0.021931863018312875,
                  -0.0019537261317119302, -0.00032910445671233547, -3.8788212776372465e-5, -7.376592164341867e-6])


def test_lv10_published_kkt_point(libs):
    """docs/src/develop.md:84-106: Ipopt solution and multipliers of luksan_vlcek_model(10).  With only the
    evaluators: c(x*) = 0 and grad f(x*) + J(x*)^T lambda* = 0 — pins obj-grad / cons / jac sign conventions and
    the COO layout end to end without a solver."""
    o = oracle_model(models.luksan_vlcek_model(10))
    assert np.max(np.abs(o.cons(XSTAR))) < 1e-8
    jr, jc = o.jac_structure()
    J = np.zeros((o.ncon, o.nvar))
    np.add.at(J, (jr - 1, jc - 1), o.jac_coord(XSTAR))
    assert np.max(np.abs(o.grad(XSTAR) + J.T @ LSTAR)) < 1e-6


def test_lv_layout_tables(libs):
    """SURVEY App. B (derived from hessian.jl / simdfunction.jl rules): LV objective comp1=(1,2,1), comp2=(1,2,3,1),
    o2step=3; constraint 10 first-order visits -> o1step 3, 17 second-order visits -> o2step 6 with keys
    (a,a),(b,b),(a,b),(b,a),(c,c),(c,a); nnzh = 9N-15; the constraint block comes first in the Hessian."""
    N = 20
    o = oracle_model(models.luksan_vlcek_model(N))
    assert (o.nvar, o.ncon, o.nnzj, o.nnzh, o.nnzg) == (N, N - 2, 3 * (N - 2), 9 * N - 15, 2 * (N - 1))
    con, obj = o.pattern_info(0), o.pattern_info(1)
    assert (con["o1step"], con["o2step"], con["n1"], con["n2"], con["o2"]) == (3, 6, 10, 17, 0)
    assert (obj["o1step"], obj["o2step"], obj["n1"], obj["n2"], obj["o2"]) == (2, 3, 3, 4, 6 * (N - 2))
    assert o.pattern_comp(1, 1) == [1, 2, 1] and o.pattern_comp(1, 2) == [1, 2, 3, 1]
    r, c = o.hess_structure()
    i = 1   # first data point of the constraint: a = x[2], b = x[3], c = x[1]
    assert list(zip(r[:6], c[:6])) == [(i + 1, i + 1), (i + 2, i + 2), (i + 2, i + 1), (i + 2, i + 1), (i, i), (i + 1, i)]
    k = 6 * (N - 2)   # first data point of the objective, i = 2: slots (1,1), (2,2), (2,1)
    assert list(zip(r[k:k + 3], c[k:k + 3])) == [(1, 1), (2, 2), (2, 1)]
    # insertion order decides the ranges (docs/src/performance.jl:13-17 adds the objective first)
    o2 = oracle_model(models.luksan_vlcek_model(N, obj_first=True))
    assert o2.pattern_info(0)["o2"] == 0 and o2.pattern_info(1)["o2"] == 3 * (N - 1)


def test_lv_split_variant_layout(libs):
    """test/NLPTest/luksan.jl:17-26: base con1 (o2step 1) + augmentation con2 (o2step 6), 2-D variables."""
    N, M = 20, 2
    o = oracle_model(models.luksan_vlcek_split_model(N, M))
    assert o.ncon == (N - 2) * M
    assert o.pattern_info(0)["o2step"] == 1 and o.pattern_info(1)["o2step"] == 6
    assert o.pattern_info(1)["o0"] == o.pattern_info(0)["o0"]          # augmentation rows land on the base block
    # same numbers as the merged 1-D model when M == 1
    o1 = oracle_model(models.luksan_vlcek_split_model(N, 1))
    om = oracle_model(models.luksan_vlcek_model(N))
    x = models.lv_x0(N) + 0.05
    np.testing.assert_allclose(o1.cons(x), om.cons(x), rtol=1e-14)
    np.testing.assert_allclose(o1.obj(x), om.obj(x), rtol=1e-14)


def test_acopf_layout_table(libs):
    """SURVEY App. B / test/NLPTest/power.jl:112-213: per-block strides of the 15 ACOPF blocks and the totals
    nnzj = nref + 30 nbr + 2 nbus + 2 ngen, nnzh = ngen + 44 nbr + 2 nbus."""
    nbus, nbr, ngen = 30, 41, 6
    o = oracle_model(models.ac_power_model(models.synthetic_power_data(nbus, nbr, ngen, seed=3)))
    steps = [(o.pattern_info(k)["o1step"], o.pattern_info(k)["o2step"]) for k in range(o.npatterns)]
    assert steps == [(1, 1), (1, 0), (5, 10), (5, 10), (5, 10), (5, 10), (2, 0), (2, 2), (2, 2), (1, 1), (1, 1),
                     (1, 0), (1, 0), (1, 0), (1, 0)]
    assert o.nvar == 2 * nbus + 2 * ngen + 4 * nbr
    assert o.ncon == 1 + 7 * nbr + 2 * nbus
    assert o.nnzj == 1 + 30 * nbr + 2 * nbus + 2 * ngen
    assert o.nnzh == ngen + 44 * nbr + 2 * nbus


def test_lv_compiled_baseline_equals_interpreter(libs):
    """bench.py's cpu_baseline times a hand-specialised straight-line C version of the LV Hessian (what Julia's
    compiler makes of shessian!); it must produce exactly the interpreter's numbers."""
    import oracle
    N = 5000
    o = oracle_model(models.luksan_vlcek_model(N))
    x = models.lv_x0(N) + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)
    y = np.random.default_rng(1).standard_normal(N - 2)
    ref = o.hess_coord(x, y, 0.5)
    np.testing.assert_allclose(oracle.lv_hess_compiled(N, x, y, 0.5), ref, rtol=1e-14, atol=0)
    np.testing.assert_allclose(oracle.lv_hess_compiled(N, x, y, 0.5, threads=4), ref, rtol=1e-14, atol=0)


# ---- the reference's own Ipopt logs of the parametric LV-10 model (docs/src/parameters.md) -----------------------------------
# OBSERVED from a Julia run of the reference (the rendered output of docs/src/parameters.jl): counts, objective values at x0 for
# three parameter sets, and three full-step Newton trajectories whose every iterate depends on the Hessian VALUES at the previous
# one.  Columns as printed: objective, inf_pr, inf_du (scaled by Ipopt's objective scaling 100 / max|grad f(x0)|), ||d||.
def doc_parametric_lv10(N=10):
    """docs/src/parameters.md:24-82: θ = [100, 1]; objective FIRST (θ[1] (x[i-1]² - x[i])² + (x[i-1] - θ[2])², i = 2:N), then the
    LV constraint over i = 1:N-2 — the insertion order decides the Hessian slot offsets."""
    from exahip.graph import exp, sin
    c = ExaCore()
    th = c.add_par([100.0, 1.0])
    x = c.add_var(N, start=np.array([models.luksan_vlcek_x0(i) for i in range(1, N + 1)]))
    c.add_obj(lambda i: th[1] * (x[i - 1] ** 2 - x[i]) ** 2 + (x[i - 1] - th[2]) ** 2, rng(2, N))
    c.add_con(lambda i: 3 * x[i + 1] ** 3 + 2 * x[i + 2] - 5 + sin(x[i + 1] - x[i + 2]) * sin(x[i + 1] + x[i + 2]) + 4 * x[i + 1]
              - x[i] * exp(x[i] - x[i + 1]) - 3, rng(1, N - 2))
    return c, th


IPOPT_LOGS = {          # theta -> (rows of docs/src/parameters.md, unscaled final objective, final constraint violation, final scaled inf_du)
    (100.0, 1.0): ("""
   0  2.0570000e+03 2.48e+01 2.73e+01  -1.0 0.00e+00
   1  1.0953147e+03 1.49e+01 8.27e+01  -1.0 2.20e+00
   2  3.2865521e+02 4.28e+00 1.36e+02  -1.0 1.43e+00
   3  1.3995370e+01 3.09e-01 2.18e+01  -1.0 5.63e-01
   4  6.2325715e+00 1.73e-02 8.47e-01  -1.0 2.10e-01
   5  6.2324586e+00 1.15e-05 8.16e-04  -1.7 3.35e-03
   6  6.2324586e+00 8.35e-12 7.97e-10  -5.7 2.00e-06""", 6.232458632437464, 8.3542062156993779e-12, 7.9746301767912855e-10),     # :148-154, :177, :161, :160
    (200.0, 1.0): ("""
   0  4.0898000e+03 2.48e+01 2.70e+01  -1.0 0.00e+00
   1  2.1810502e+03 1.49e+01 8.27e+01  -1.0 2.20e+00
   2  6.5137192e+02 4.27e+00 1.36e+02  -1.0 1.43e+00
   3  2.4064340e+01 3.08e-01 2.18e+01  -1.0 5.62e-01
   4  8.6476680e+00 1.72e-02 8.45e-01  -1.0 2.10e-01
   5  8.6474398e+00 1.15e-05 8.07e-04  -1.7 3.39e-03
   6  8.6474398e+00 8.42e-12 7.91e-10  -5.7 2.03e-06""", 8.647439751691484, 8.4190432403374871e-12, 7.9051456515071309e-10),     # :207-213, :236, :220, :219
    (200.0, 0.5): ("""
   0  4.0810500e+03 2.48e+01 2.69e+01  -1.0 0.00e+00
   1  2.1767809e+03 1.49e+01 8.26e+01  -1.0 2.20e+00
   2  6.5050886e+02 4.27e+00 1.36e+02  -1.0 1.43e+00
   3  2.4276149e+01 3.07e-01 2.18e+01  -1.0 5.61e-01
   4  8.8465512e+00 1.72e-02 8.43e-01  -1.0 2.09e-01
   5  8.8451636e+00 1.15e-05 8.04e-04  -1.7 3.40e-03
   6  8.8451630e+00 8.47e-12 7.88e-10  -5.7 2.05e-06""", 8.845162987294774, 8.4678930534209940e-12, 7.8812124187921384e-10),     # :266-272, :295, :279, :278
}


def check_ipopt_logs(m, x0, set_theta):
    """m: anything with the NLPModels callbacks (oracle or HIP model); set_theta(list) = the reference's set_parameter!"""
    from kktsolve import newton_full_step
    sizes = m if hasattr(m, "nnzj") else m.meta
    assert (sizes.nvar, sizes.ncon, sizes.nnzj, sizes.nnzh) == (10, 8, 24, 75)      # :104-116, :133-139
    for theta, (log, fstar, cviol, dinf) in IPOPT_LOGS.items():
        set_theta(list(theta))
        gmax = float(np.max(np.abs(m.grad(x0))))
        if theta == (100.0, 1.0):
            assert abs(gmax - 792.0) < 1e-9                    # :159-160: scaled / unscaled objective = 100 / 792
            assert abs(100.0 / gmax - 7.8692659500473017e-01 / 6.2324586324374636e+00) < 1e-15
        rows, x, y = newton_full_step(m, x0, sizes.ncon, 6)
        printed = [ln.split() for ln in log.strip().splitlines()]
        for k, (p, r) in enumerate(zip(printed, rows)):
            obj, inf_pr, inf_du, dnorm = r[0], r[1], r[2] * 100.0 / gmax, r[3]
            assert "%.7e" % obj == p[1], (theta, k, "objective", obj, p[1])
            if k < 6:           # every printed digit
                assert "%.2e" % inf_pr == p[2], (theta, k, "inf_pr", inf_pr, p[2])
                assert "%.2e" % inf_du == p[3], (theta, k, "inf_du", inf_du, p[3])
            else:               # the last iterate's residuals are 1e-4 above the rounding of the linear solves: the log's full-precision figures, 0.1 %
                assert abs(inf_pr - cviol) <= 1e-3 * cviol and abs(inf_du - dinf) <= 1e-3 * dinf, (theta, inf_pr, cviol, inf_du, dinf)
            assert "%.2e" % dnorm == p[5] or abs(dnorm - float(p[5])) <= 0.006 * float(p[5]), (theta, k, "||d||", dnorm, p[5])
        assert abs(m.obj(x) - fstar) <= 1e-9 * fstar, (theta, m.obj(x), fstar)


def test_reference_ipopt_logs_of_the_parametric_lv10(libs):
    """docs/src/parameters.md:95-295 — the only numbers under /root/reference that depend on hess_coord! VALUES."""
    core, th = doc_parametric_lv10()
    o = oracle_model(core)
    x0 = o.meta()[0]
    assert o.obj(x0) == 2057.0                                  # :148
    assert "%.2e" % np.max(np.abs(o.cons(x0))) == "2.48e+01"
    check_ipopt_logs(o, x0, lambda v: o.set_value(th.offset, v))
    o.set_value(th.offset, [200.0, 1.0])
    assert abs(o.obj(x0) - 4089.8) < 1e-9                       # :207
    o.set_value(th.offset, [200.0, 0.5])
    assert abs(o.obj(x0) - 4081.05) < 1e-9                      # :266
