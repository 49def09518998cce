"""Benchmark / test models stated through the ExaCore mirror.

Each builder cites the reference source whose expression text it restates; expression *shape* (operand
order, literal kinds Int vs Float) is kept because it fixes the COO slot order and the rounding.
"""
from __future__ import annotations

import numpy as np

from .core import ExaCore, Table, product, rng
from .graph import cos, exp, sin, sqrt


# ---------------------------------------------------------------------------------------------------------------
# Luksan-Vlcek (Rosenbrock-type) — BASELINE.json configs 1, 2, 5
# ---------------------------------------------------------------------------------------------------------------
def luksan_vlcek_x0(i):
    """test/NLPTest/luksan.jl:13-15"""
    return -1.2 if i % 2 == 1 else 1.0


def lv_x0(N):
    x0 = np.ones(N)
    x0[0::2] = -1.2      # i odd (1-based)
    return x0


def luksan_vlcek_model(N, obj_first=False):
    """1-D model of benchmark/runbenchmark.jl:163-169 ("rosenrock") == docs/src/gpu.jl:10-40.
    Constraint added BEFORE the objective (so its Hessian slots come first) unless obj_first
    (docs/src/performance.jl:13-17)."""
    c = ExaCore()
    x = c.add_var(N, start=lv_x0(N))

    def con(i):
        return (3 * x[i + 1] ** 3 + 2 * x[i + 2] - 5 + sin(x[i + 1] - x[i + 2]) * sin(x[i + 1] + x[i + 2])
                + 4 * x[i + 1] - x[i] * exp(x[i] - x[i + 1]) - 3)

    def obj(i):
        return 100 * (x[i - 1] ** 2 - x[i]) ** 2 + (x[i - 1] - 1) ** 2

    if obj_first:
        c.add_obj(obj, rng(2, N))
        c.add_con(con, rng(1, N - 2))
    else:
        c.add_con(con, rng(1, N - 2))
        c.add_obj(obj, rng(2, N))
    return c


def luksan_vlcek_split_model(N, M=1):
    """2-D variant of test/NLPTest/luksan.jl:17-26: x[N, M]; constraint split into base con1 +
    augmentation con2 with a tuple target (i, j); product iterators."""
    c = ExaCore()
    x0 = np.array([[luksan_vlcek_x0(i) for _ in range(M)] for i in range(1, N + 1)])
    x = c.add_var(N, M, start=x0)

    def con1(p):
        i, j = p
        return 3 * x[i + 1, j] ** 3 + 2 * x[i + 2, j] - 5

    def con2(p):
        i, j = p
        return ((i, j), sin(x[i + 1, j] - x[i + 2, j]) * sin(x[i + 1, j] + x[i + 2, j]) + 4 * x[i + 1, j]
                - x[i, j] * exp(x[i, j] - x[i + 1, j]) - 3)

    def obj(p):
        i, j = p
        return 100 * (x[i - 1, j] ** 2 - x[i, j]) ** 2 + (x[i - 1, j] - 1) ** 2

    s = c.add_con(con1, product(rng(1, N - 2), rng(1, M)))
    c.add_con_aug(s, con2, product(rng(1, N - 2), rng(1, M)))
    c.add_obj(obj, product(rng(2, N), rng(1, M)))
    return c


def luksan_vlcek_struct_model(N, M=1):
    """test/NLPTest/luksan_struct.jl:1-20: the split model iterated over an array of NESTED structs — data = [I1(I2((i, j)))],
    fields reached through the access path `i.i.i[1]`, `i.i.i[2]` (graph.jl:190-196 DataIndexed of DataIndexed).  Here: a list of
    nested namedtuples, the same path p.i.i[0] / p.i.i[1]; each distinct path becomes one SoA column of the pattern's iterator."""
    from collections import namedtuple
    I2 = namedtuple("I2", "i")
    I1 = namedtuple("I1", "i")
    c = ExaCore()
    data = [[I1(I2((i, j))) for j in range(1, M + 1)] for i in range(1, N + 1)]        # data[i-1][j-1], column-major when flattened below
    x0 = np.array([[luksan_vlcek_x0(data[i][j].i.i[0]) for j in range(M)] for i in range(N)])
    x = c.add_var(N, M, start=x0)
    rows = np.empty((N - 2, M), dtype=object)                                         # data[1:end-2, :]: a Matrix of structs, size (N - 2, M)
    for i in range(N - 2):
        for j in range(M):
            rows[i, j] = data[i][j]

    def con1(p):
        i, j = p.i.i[0], p.i.i[1]
        return 3 * x[i + 1, j] ** 3 + 2 * x[i + 2, j] - 5

    def con2(p):
        i, j = p.i.i[0], p.i.i[1]
        return ((i, j), sin(x[i + 1, j] - x[i + 2, j]) * sin(x[i + 1, j] + x[i + 2, j]) + 4 * x[i + 1, j]
                - x[i, j] * exp(x[i, j] - x[i + 1, j]) - 3)

    def obj(p):
        i, j = p
        return 100 * (x[i - 1, j] ** 2 - x[i, j]) ** 2 + (x[i - 1, j] - 1) ** 2

    s = c.add_con(con1, rows)
    c.add_con_aug(s, con2, rows)
    c.add_obj(obj, product(rng(2, N), rng(1, M)))
    return c


# ---------------------------------------------------------------------------------------------------------------
# Goddard rocket (COPS-3 "rocket") — BASELINE.json config 3
# ---------------------------------------------------------------------------------------------------------------
def rocket_model(nh):
    """Goddard rocket, trapezoidal collocation.  ONLY the velocity pattern is in the reference
    (README.md:20-26, copied operand for operand); the rest is restated from the COPS 3.0 problem
    description and is this build's own model — COPSBenchmark.jl is not vendored.
    Nodes 0..nh are stored 1-based: variable index i+1 holds node i."""
    h_0, v_0, m_0, g_0 = 1.0, 0.0, 1.0, 1.0
    T_c, h_c, v_c, m_c = 3.5, 500.0, 620.0, 0.6
    c_ = 0.5 * np.sqrt(g_0 * h_0)
    m_f = m_c * m_0
    D_c = 0.5 * v_c * (m_0 / g_0)
    T_max = T_c * m_0 * g_0

    core = ExaCore(minimize=False)
    K = nh + 1
    node = np.arange(0, K)
    h = core.add_var(rng(0, nh), start=np.ones(K), lvar=h_0)
    v = core.add_var(rng(0, nh), start=(node / nh) * (1.0 - node / nh), lvar=0.0)
    m = core.add_var(rng(0, nh), start=(m_f - m_0) * (node / nh) + m_0, lvar=m_f, uvar=m_0)
    tau = core.add_var(rng(0, nh), start=T_max / 2.0, lvar=0.0, uvar=T_max)
    dt = core.add_var(1, start=1.0 / nh, lvar=0.0)

    # objective: maximise final altitude
    core.add_obj(lambda i: h[i], rng(nh, nh))

    # altitude dynamics
    core.add_con(lambda i: -h[i] + h[i - 1] + 0.5 * dt[1] * (v[i] + v[i - 1]), rng(1, nh))
    # velocity dynamics — README.md:20-26
    core.add_con(
        lambda i: -v[i] + v[i - 1] + 0.5 * dt[1] * (
            (tau[i] - D_c * v[i] ** 2 * exp(-h_c * (h[i] - h_0) / h_0) - m[i] * g_0 * (h_0 / h[i]) ** 2) / m[i]
            + (tau[i - 1] - D_c * v[i - 1] ** 2 * exp(-h_c * (h[i - 1] - h_0) / h_0)
               - m[i - 1] * g_0 * (h_0 / h[i - 1]) ** 2) / m[i - 1]),
        rng(1, nh))
    # mass dynamics
    core.add_con(lambda i: -m[i] + m[i - 1] - 0.5 * dt[1] * (tau[i] + tau[i - 1]) / c_, rng(1, nh))
    # boundary rows
    core.add_con(lambda i: h[i] - h_0, rng(0, 0))
    core.add_con(lambda i: v[i] - v_0, rng(0, 0))
    core.add_con(lambda i: m[i] - m_0, rng(0, 0))
    core.add_con(lambda i: m[i] - m_f, rng(nh, nh))
    return core


# ---------------------------------------------------------------------------------------------------------------
# AC optimal power flow — BASELINE.json config 4
# ---------------------------------------------------------------------------------------------------------------
def synthetic_power_data(nbus, nbr, ngen, seed=0, topology="random"):
    """Synthetic ACOPF data with the field layout of test/NLPTest/power.jl:31-93 (the PGLIB case files and the
    ExaPowerIO artifact are not available offline — SURVEY §8d config 4).
    topology "random": a spanning chain plus extra branches between uniformly random buses, listed in random order —
    every gather of a bus variable lands on its own cache line (the worst case for the branch-indexed patterns).
    topology "bus":    what a PGLIB / MATPOWER case file looks like — branches LISTED BY FROM-BUS, ends close in the
    numbering (buses of one area are numbered together): a spanning chain, extra branches to a bus a geometrically
    distributed distance away (mean 12, 90 %), a few long ties between areas (10 %); degree distribution as in
    transmission cases (mean 2*nbr/nbus ~ 3.2, most buses 2-4, hubs up to ~10)."""
    r = np.random.default_rng(seed)
    f_bus = np.empty(nbr, dtype=np.int64)
    t_bus = np.empty(nbr, dtype=np.int64)
    nchain = min(nbr, nbus - 1)
    f_bus[:nchain] = np.arange(1, nchain + 1)
    t_bus[:nchain] = np.arange(2, nchain + 2)
    extra = nbr - nchain
    if topology == "bus":
        if extra > 0:
            f = r.integers(1, nbus + 1, size=extra)
            near = r.geometric(1.0 / 12.0, size=extra) + 1                   # 2, 3, ...: the chain already joins neighbours
            far = r.integers(1, nbus, size=extra)
            d = np.where(r.uniform(size=extra) < 0.9, near, far)
            t = f + d * np.where(r.uniform(size=extra) < 0.5, 1, -1)
            t = (t - 1) % nbus + 1                                             # (the numbering wraps around)
            t = np.where(t == f, np.where(f < nbus, f + 1, f - 1), t)
            f_bus[nchain:] = np.minimum(f, t)
            t_bus[nchain:] = np.maximum(f, t)
        order = np.lexsort((t_bus, f_bus))                                     # listed by from-bus, like a case file
        f_bus, t_bus = f_bus[order], t_bus[order]
    elif topology == "random":
        if extra > 0:
            f = r.integers(1, nbus + 1, size=extra)
            t = r.integers(1, nbus, size=extra)
            t = np.where(t >= f, t + 1, t)          # t != f
            f_bus[nchain:] = f
            t_bus[nchain:] = t
        perm = r.permutation(nbr)
        f_bus, t_bus = f_bus[perm], t_bus[perm]
    else:
        raise ValueError("topology must be 'random' or 'bus'")
    bidx = np.arange(1, nbr + 1)
    # arcs: k = 1..nbr "from" side, nbr+1..2nbr "to" side (ref[:arcs] = arcs_from ++ arcs_to)
    arc_i = np.arange(1, 2 * nbr + 1)
    arc_bus = np.concatenate([f_bus, t_bus])
    rate_a = r.uniform(1.0, 10.0, size=nbr)
    coef = {f"c{k}": r.uniform(-10.0, 10.0, size=nbr) for k in range(1, 9)}
    branch = Table(i=bidx, j=np.ones(nbr, dtype=np.int64), f_idx=bidx, t_idx=bidx + nbr, f_bus=f_bus,
                   t_bus=t_bus, rate_a_sq=rate_a ** 2, **coef)
    bus = Table(i=np.arange(1, nbus + 1), pd=r.uniform(0.0, 2.0, nbus), gs=r.uniform(0.0, 0.1, nbus),
                qd=r.uniform(-0.5, 0.5, nbus), bs=r.uniform(0.0, 0.2, nbus))
    gen = Table(i=np.arange(1, ngen + 1), cost1=r.uniform(0.0, 1.0, ngen), cost2=r.uniform(1.0, 50.0, ngen),
                cost3=r.uniform(0.0, 10.0, ngen), bus=np.sort(r.choice(nbus, size=ngen, replace=ngen > nbus)) + 1)
    arc = Table(i=arc_i, rate_a=np.concatenate([rate_a, rate_a]), bus=arc_bus)
    return dict(
        bus=bus, gen=gen, arc=arc, branch=branch, ref_buses=np.array([1], dtype=np.int64),
        vmax=np.full(nbus, 1.1), vmin=np.full(nbus, 0.9),
        pmax=r.uniform(1.0, 5.0, ngen), pmin=np.zeros(ngen),
        qmax=r.uniform(1.0, 3.0, ngen), qmin=-r.uniform(1.0, 3.0, ngen),
        rate_a=np.concatenate([rate_a, rate_a]), rate_a_lo=-np.concatenate([rate_a, rate_a]),
        angmax=np.full(nbr, np.pi / 6), angmin=np.full(nbr, -np.pi / 6),
    )


def ac_power_model(data, core=None):
    """test/NLPTest/power.jl:112-213 (`__exa_ac_power_model`): variable order va, vm, pg, qg, p, q;
    15 blocks (1 objective, 10 constraints, 4 augmentations).
    With `core` = a recipe core (`ExaCore(examples=(data,))`) and `data` = its placeholder the same statements give
    the DATA-DEFINED recipe: every size is a table length, every bound a data field."""
    from .recipe import length
    w = ExaCore() if core is None else core
    nbus, ngen, narc = length(data["bus"]), length(data["gen"]), length(data["arc"])
    va = w.add_var(nbus, name="va")
    vm = w.add_var(nbus, start=1.0, lvar=data["vmin"], uvar=data["vmax"], name="vm")
    pg = w.add_var(ngen, lvar=data["pmin"], uvar=data["pmax"], name="pg")
    qg = w.add_var(ngen, lvar=data["qmin"], uvar=data["qmax"], name="qg")
    p = w.add_var(narc, lvar=data["rate_a_lo"], uvar=data["rate_a"], name="p")
    q = w.add_var(narc, lvar=data["rate_a_lo"], uvar=data["rate_a"], name="q")

    w.add_obj(lambda g: g.cost1 * pg[g.i] ** 2 + g.cost2 * pg[g.i] + g.cost3, data["gen"])

    w.add_con(lambda i: va[i], data["ref_buses"])
    w.add_con(
        lambda b: p[b.f_idx] - b.c5 * vm[b.f_bus] ** 2
        - b.c3 * (vm[b.f_bus] * vm[b.t_bus] * cos(va[b.f_bus] - va[b.t_bus]))
        - b.c4 * (vm[b.f_bus] * vm[b.t_bus] * sin(va[b.f_bus] - va[b.t_bus])), data["branch"])
    w.add_con(
        lambda b: q[b.f_idx] + b.c6 * vm[b.f_bus] ** 2
        + b.c4 * (vm[b.f_bus] * vm[b.t_bus] * cos(va[b.f_bus] - va[b.t_bus]))
        - b.c3 * (vm[b.f_bus] * vm[b.t_bus] * sin(va[b.f_bus] - va[b.t_bus])), data["branch"])
    w.add_con(
        lambda b: p[b.t_idx] - b.c7 * vm[b.t_bus] ** 2
        - b.c1 * (vm[b.t_bus] * vm[b.f_bus] * cos(va[b.t_bus] - va[b.f_bus]))
        - b.c2 * (vm[b.t_bus] * vm[b.f_bus] * sin(va[b.t_bus] - va[b.f_bus])), data["branch"])
    w.add_con(
        lambda b: q[b.t_idx] + b.c8 * vm[b.t_bus] ** 2
        + b.c2 * (vm[b.t_bus] * vm[b.f_bus] * cos(va[b.t_bus] - va[b.f_bus]))
        - b.c1 * (vm[b.t_bus] * vm[b.f_bus] * sin(va[b.t_bus] - va[b.f_bus])), data["branch"])
    w.add_con(lambda b: va[b.f_bus] - va[b.t_bus], data["branch"], lcon=data["angmin"], ucon=data["angmax"])
    w.add_con(lambda b: p[b.f_idx] ** 2 + q[b.f_idx] ** 2 - b.rate_a_sq, data["branch"],
              lcon=-np.inf)
    w.add_con(lambda b: p[b.t_idx] ** 2 + q[b.t_idx] ** 2 - b.rate_a_sq, data["branch"],
              lcon=-np.inf)
    c9 = w.add_con(lambda b: b.pd + b.gs * vm[b.i] ** 2, data["bus"])
    c10 = w.add_con(lambda b: b.qd - b.bs * vm[b.i] ** 2, data["bus"])
    w.add_con_aug(c9, lambda a: (a.bus, p[a.i]), data["arc"])
    w.add_con_aug(c10, lambda a: (a.bus, q[a.i]), data["arc"])
    w.add_con_aug(c9, lambda g: (g.bus, -pg[g.i]), data["gen"])
    w.add_con_aug(c10, lambda g: (g.bus, -qg[g.i]), data["gen"])
    return w


def acopf_start(core, seed=2):
    """Evaluation point for the synthetic ACOPF: vm = 1 (x0), everything else U(-0.1, 0.1)."""
    r = np.random.default_rng(seed)
    ir = core.to_ir()
    x = ir.x0.copy()
    u = r.uniform(-0.1, 0.1, size=x.size)
    return np.where(x == 0.0, u, x + 0.0 * u)


# ---------------------------------------------------------------------------------------------------------------
# COPS hanging chain / electrons on a sphere — the other two models of the reference's benchmark harness
# ---------------------------------------------------------------------------------------------------------------
def cops_chain_model(n):
    """benchmark/runbenchmark.jl:239-264 (`cops_chain_model`)."""
    nh = max(2, (n - 4) // 4)
    L, a, b = 4, 1, 3
    tmin = 1 / 4 if b > a else 3 / 4
    tf = 1.0
    h = tf / nh
    k = np.arange(1, nh + 2)
    c = ExaCore()
    u = c.add_var(nh + 1, start=4 * abs(b - a) * (k / nh - tmin))
    x1 = c.add_var(nh + 1, start=4 * abs(b - a) * k / nh * (1 / 2 * k / nh - tmin) + a)
    x2 = c.add_var(nh + 1, start=(4 * abs(b - a) * k / nh * (1 / 2 * k / nh - tmin) + a) * (4 * abs(b - a) * (k / nh - tmin)))
    x3 = c.add_var(nh + 1, start=4 * abs(b - a) * (k / nh - tmin))
    c.add_obj(x2[nh + 1])
    c.add_con(lambda j: x1[j + 1] - x1[j] - 1 / 2 * h * (u[j] + u[j + 1]), rng(1, nh))
    c.add_con(x1[1] - a)
    c.add_con(x1[nh + 1] - b)
    c.add_con(x2[1])
    c.add_con(x3[1])
    c.add_con(x3[nh + 1] - L)
    c.add_con(lambda j: x2[j + 1] - x2[j] - 1 / 2 * h * (x1[j] * sqrt(1 + u[j] ** 2) + x1[j + 1] * sqrt(1 + u[j + 1] ** 2)), rng(1, nh))
    c.add_con(lambda j: x3[j + 1] - x3[j] - 1 / 2 * h * (sqrt(1 + u[j] ** 2) + sqrt(1 + u[j + 1] ** 2)), rng(1, nh))
    return c


def cops_elec_model(npts, seed=2713):
    """benchmark/runbenchmark.jl:267-282 (`cops_elec_model`); the start point uses numpy's generator instead of
    Julia's Random.seed!(2713) stream (values differ, the model does not)."""
    r = np.random.default_rng(seed)
    theta = 2 * np.pi * r.uniform(size=npts)
    phi = np.pi * r.uniform(size=npts)
    ii, jj = np.triu_indices(npts, k=1)
    itr = Table(i=ii + 1, j=jj + 1)
    core = ExaCore()
    x = core.add_var(rng(1, npts), start=np.cos(theta) * np.sin(phi))
    y = core.add_var(rng(1, npts), start=np.sin(theta) * np.sin(phi))
    z = core.add_var(rng(1, npts), start=np.cos(phi))
    core.add_obj(lambda p: 1 / sqrt((x[p.i] - x[p.j]) ** 2 + (y[p.i] - y[p.j]) ** 2 + (z[p.i] - z[p.j]) ** 2), itr)
    core.add_con(lambda i: x[i] ** 2 + y[i] ** 2 + z[i] ** 2 - 1, rng(1, npts))
    return core
