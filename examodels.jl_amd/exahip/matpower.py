"""MATPOWER case file (.m, the format of the PGLIB-OPF library) -> the ACOPF tables of test/NLPTest/power.jl:31-93.

The reference reads `pglib_opf_case78484_epigrids.m` through PowerModels.jl / ExaPowerIO (test/NLPTest/power.jl:1-16:
`parse_file`, `standardize_cost_terms!(order = 2)`, `calc_thermal_limits!`, `build_ref`), none of which is vendored in
/root/reference and none of which can run here.  This module restates that data path from the MATPOWER case format
(caseformat.m) and the published PowerModels conventions, so that `bench.py --config 4 --case FILE` and
`models.ac_power_model(matpower.load(FILE))` run on a real network when the file is available:

  * per-unit: Pd, Qd, Gs, Bs, Pmax/min, Qmax/min, rateA divided by baseMVA; angles (shift, angmin, angmax) to radians;
  * inactive elements dropped: bus type 4, generators and branches with status <= 0 (or attached to a dropped bus);
  * polynomial generator costs (model 2) brought to order 2 and scaled to per-unit power: c_k * baseMVA^k;
  * branch series admittance y = 1 / (r + jx), tap tr + j ti = tap * exp(j shift) (tap 0 means 1), line charging split
    evenly (b_fr = b_to = b / 2, g_fr = g_to = 0), and the eight coefficients c1..c8 exactly as power.jl:62-78 forms them;
  * a branch without a thermal rating gets the bound implied by its angle limits and the voltage bounds of its buses
    (`calc_thermal_limits!`);
  * arcs = [(l, f, t) for every branch] ++ [(l, t, f) for every branch]: arc k = l is the from side, k = nbr + l the to side.

ONE deliberate difference: buses, generators and branches are numbered in FILE order.  PowerModels keeps them in Julia
`Dict`s and power.jl numbers them in the dictionaries' iteration (hash) order, which cannot be reproduced outside Julia; the
two numberings give the same model up to a permutation of variables and constraints.
"""
from __future__ import annotations

import math
import re

import numpy as np

from .core import Table


def _matrix(text, name):
    m = re.search(r"mpc\." + name + r"\s*=\s*\[(.*?)\]\s*;", text, re.S)
    if not m:
        return np.zeros((0, 0))
    rows = []
    for line in m.group(1).split(";"):
        for piece in line.split("\n"):
            piece = piece.split("%", 1)[0].strip()
            if piece:
                rows.append([float(t) for t in re.split(r"[\s,]+", piece)])
    if not rows:
        return np.zeros((0, 0))
    w = max(len(r) for r in rows)
    return np.array([r + [0.0] * (w - len(r)) for r in rows])


def load(path):
    """-> dict with the keys models.ac_power_model expects (bus, gen, arc, branch tables + bound vectors)."""
    text = open(path).read()
    text = "\n".join(ln.split("%", 1)[0] for ln in text.splitlines())          # MATLAB comments
    base = float(re.search(r"mpc\.baseMVA\s*=\s*([0-9.eE+-]+)", text).group(1))
    bus, gen, branch, cost = (_matrix(text, n) for n in ("bus", "gen", "branch", "gencost"))
    if bus.size == 0 or branch.size == 0:
        raise ValueError(f"{path}: no mpc.bus / mpc.branch matrix found")
    # ---- buses (columns: bus_i type Pd Qd Gs Bs area Vm Va baseKV zone Vmax Vmin)
    keep_bus = bus[:, 1] != 4
    bus = bus[keep_bus]
    busid = {int(b): k + 1 for k, b in enumerate(bus[:, 0])}
    nbus = len(bus)
    vmax, vmin = bus[:, 11].copy(), bus[:, 12].copy()
    bus_t = Table(i=np.arange(1, nbus + 1), pd=bus[:, 2] / base, gs=bus[:, 4] / base, qd=bus[:, 3] / base, bs=bus[:, 5] / base)
    ref_buses = np.array([busid[int(b)] for b, t in zip(bus[:, 0], bus[:, 1]) if t == 3], dtype=np.int64)
    # ---- generators (bus Pg Qg Qmax Qmin Vg mBase status Pmax Pmin ...) + costs (model startup shutdown n c(n-1) ... c0)
    gmask = np.array([g[7] > 0 and int(g[0]) in busid for g in gen], dtype=bool) if gen.size else np.zeros(0, dtype=bool)
    gen_k = gen[gmask]
    ngen = len(gen_k)
    c1, c2, c3 = np.zeros(ngen), np.zeros(ngen), np.zeros(ngen)
    if cost.size:
        cost_k = cost[:len(gen)][gmask]
        for k, row in enumerate(cost_k):
            if int(row[0]) != 2:
                raise ValueError("only polynomial generator costs (model 2) are supported")
            n = int(row[3])
            coef = list(row[4:4 + n])                       # c_{n-1} ... c_0 in $ / MW^k
            coef = [0.0] * max(0, 3 - n) + coef[-3:] if n <= 3 else coef[-3:]
            c1[k], c2[k], c3[k] = coef[0] * base ** 2, coef[1] * base, coef[2]
    gen_t = Table(i=np.arange(1, ngen + 1), cost1=c1, cost2=c2, cost3=c3, bus=np.array([busid[int(b)] for b in gen_k[:, 0]], dtype=np.int64))
    # ---- branches (fbus tbus r x b rateA rateB rateC ratio angle status angmin angmax)
    bmask = np.array([(br[10] != 0) and int(br[0]) in busid and int(br[1]) in busid for br in branch], dtype=bool)
    br = branch[bmask]
    nbr = len(br)
    f_bus = np.array([busid[int(b)] for b in br[:, 0]], dtype=np.int64)
    t_bus = np.array([busid[int(b)] for b in br[:, 1]], dtype=np.int64)
    r, x, bc = br[:, 2], br[:, 3], br[:, 4]
    den = r * r + x * x
    g, b = r / den, -x / den
    tap = np.where(br[:, 8] == 0.0, 1.0, br[:, 8])
    shift = np.deg2rad(br[:, 9])
    tr, ti = tap * np.cos(shift), tap * np.sin(shift)
    ttm = tr * tr + ti * ti
    g_fr = g_to = np.zeros(nbr)
    b_fr = b_to = bc / 2.0
    coef = {
        "c1": (-g * tr - b * ti) / ttm, "c2": (-b * tr + g * ti) / ttm, "c3": (-g * tr + b * ti) / ttm, "c4": (-b * tr - g * ti) / ttm,
        "c5": (g + g_fr) / ttm, "c6": (b + b_fr) / ttm, "c7": (g + g_to), "c8": (b + b_to),
    }
    angmin = np.deg2rad(br[:, 11]) if br.shape[1] > 12 else np.full(nbr, -math.pi / 3)
    angmax = np.deg2rad(br[:, 12]) if br.shape[1] > 12 else np.full(nbr, math.pi / 3)
    rate_a = br[:, 5] / base
    # calc_thermal_limits!: no rating -> what the angle limits and the voltage bounds allow through the series admittance
    missing = rate_a <= 0.0
    if np.any(missing):
        theta = np.maximum(np.abs(angmin), np.abs(angmax))
        ymag = np.sqrt(g * g + b * b)
        fv, tv = vmax[f_bus - 1], vmax[t_bus - 1]
        cmax = np.sqrt(fv * fv + tv * tv - 2.0 * fv * tv * np.cos(theta))
        rate_a = np.where(missing, ymag * np.maximum(fv, tv) * cmax, rate_a)
    bidx = np.arange(1, nbr + 1)
    branch_t = Table(i=bidx, j=np.ones(nbr, dtype=np.int64), f_idx=bidx, t_idx=bidx + nbr, f_bus=f_bus, t_bus=t_bus,
                     rate_a_sq=rate_a ** 2, **coef)
    arc_t = Table(i=np.arange(1, 2 * nbr + 1), rate_a=np.concatenate([rate_a, rate_a]), bus=np.concatenate([f_bus, t_bus]))
    return dict(
        bus=bus_t, gen=gen_t, arc=arc_t, branch=branch_t, ref_buses=ref_buses, vmax=vmax, vmin=vmin,
        pmax=gen_k[:, 8] / base, pmin=gen_k[:, 9] / base, qmax=gen_k[:, 3] / base, qmin=gen_k[:, 4] / base,
        rate_a=np.concatenate([rate_a, rate_a]), rate_a_lo=-np.concatenate([rate_a, rate_a]), angmax=angmax, angmin=angmin,
    )


def write_synthetic_case(path, nbus, nbr, ngen, seed=0):
    """Writes a MATPOWER case file of the SHAPE of a PGLIB transmission case (pglib_opf_case78484_epigrids.m is not in the image):
    buses numbered by area, branches listed by from-bus with ends close in the numbering (models.synthetic_power_data's "bus"
    topology), per-branch r / x / b, taps and phase shifts on 5 % of the branches, ratings missing on 2 % (calc_thermal_limits!),
    one slack bus, quadratic generator costs, a few inactive elements.  Numbers are printed with 17 significant digits, so
    load(path) returns exactly the tables the returned dict describes: the file is what `bench.py --config 4 --case FILE` and the
    round-trip test run the real-case data path on at full scale."""
    from .models import synthetic_power_data
    r = np.random.default_rng(seed)
    topo = synthetic_power_data(nbus, nbr, ngen, seed=seed, topology="bus")
    f_bus, t_bus = topo["branch"].cols["f_bus"], topo["branch"].cols["t_bus"]
    gen_bus = topo["gen"].cols["bus"]
    base = 100.0
    fmt = lambda v: repr(float(v))      # noqa: E731  (shortest round-trip representation)
    with open(path, "w") as fh:
        fh.write("function mpc = synthetic_case\n% written by exahip.matpower.write_synthetic_case (shape of a PGLIB-OPF case; not a real network)\n")
        fh.write("mpc.version = '2';\nmpc.baseMVA = 100.0;\n\n%% bus data\n%	bus_i	type	Pd	Qd	Gs	Bs	area	Vm	Va	baseKV	zone	Vmax	Vmin\nmpc.bus = [\n")
        pd, qd = r.uniform(0.0, 200.0, nbus) * (r.uniform(size=nbus) < 0.6), r.uniform(-50.0, 50.0, nbus)
        gs, bs = r.uniform(0.0, 10.0, nbus) * (r.uniform(size=nbus) < 0.1), r.uniform(0.0, 20.0, nbus) * (r.uniform(size=nbus) < 0.1)
        for k in range(nbus):
            fh.write(f"\t{k + 1}\t{3 if k == 0 else 1}\t{fmt(pd[k])}\t{fmt(qd[k])}\t{fmt(gs[k])}\t{fmt(bs[k])}\t{1 + k * 8 // nbus}\t1.0\t0.0\t230.0\t1\t1.1\t0.9;\n")
        fh.write("];\n\n%% generator data\n%	bus	Pg	Qg	Qmax	Qmin	Vg	mBase	status	Pmax	Pmin\nmpc.gen = [\n")
        pmax, qmax = r.uniform(100.0, 500.0, ngen), r.uniform(100.0, 300.0, ngen)
        for k in range(ngen):
            fh.write(f"\t{gen_bus[k]}\t0.0\t0.0\t{fmt(qmax[k])}\t{fmt(-qmax[k])}\t1.0\t100.0\t1\t{fmt(pmax[k])}\t0.0;\n")
        fh.write("];\n\n%% generator cost data\n%	2	startup	shutdown	n	c(n-1)	...	c0\nmpc.gencost = [\n")
        c2, c1, c0 = r.uniform(0.0, 0.01, ngen), r.uniform(10.0, 50.0, ngen), r.uniform(0.0, 100.0, ngen)
        for k in range(ngen):
            fh.write(f"\t2\t0.0\t0.0\t3\t{fmt(c2[k])}\t{fmt(c1[k])}\t{fmt(c0[k])};\n")
        fh.write("];\n\n%% branch data\n%	fbus	tbus	r	x	b	rateA	rateB	rateC	ratio	angle	status	angmin	angmax\nmpc.branch = [\n")
        rr, xx, bb = r.uniform(0.0005, 0.02, nbr), r.uniform(0.005, 0.2, nbr), r.uniform(0.0, 0.3, nbr)
        rate = np.where(r.uniform(size=nbr) < 0.02, 0.0, r.uniform(100.0, 1000.0, nbr))
        xf = r.uniform(size=nbr) < 0.05
        tap = np.where(xf, r.uniform(0.95, 1.05, nbr), 0.0)
        shift = np.where(xf & (r.uniform(size=nbr) < 0.3), r.uniform(-5.0, 5.0, nbr), 0.0)
        for k in range(nbr):
            fh.write(f"\t{f_bus[k]}\t{t_bus[k]}\t{fmt(rr[k])}\t{fmt(xx[k])}\t{fmt(bb[k])}\t{fmt(rate[k])}\t{fmt(rate[k])}\t{fmt(rate[k])}\t{fmt(tap[k])}\t{fmt(shift[k])}\t1\t-30.0\t30.0;\n")
        fh.write("];\n")
    return dict(nbus=nbus, nbr=nbr, ngen=ngen, baseMVA=base, f_bus=f_bus, t_bus=t_bus, gen_bus=gen_bus, pd=pd / base, rate_a=rate / base)
