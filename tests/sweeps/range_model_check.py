#!/usr/bin/env python
"""Sweep of random RANGE-iterated models (tests/randexpr.build_range_model) through EVERY entry point of the library against
the oracle — the round-3 machinery lives on range models: owner-computes windows of J'v / Hv, the gathered gradient, exa_eval_all,
the staged / chained hess_coord! kernels, owner-sharded outputs.  Per seed:
  * obj, cons_nln!, grad!, jac_coord!, hess_coord!, structures, J v, J'v, H v (default modes) against the oracle (1e-9: deep
    random trees are ill-conditioned);
  * exa_eval_all against the oracle; hess_coord! by all three kernels (EXAHIP_HESS_VARIANT=0/1/2) against each other (1e-12);
  * J'v / H v by atomics and by the sorted gather against the default;
  * the compressed Jacobian / Hessian, densified, against the densified oracle COO;
  * 3 ranks replayed on this GPU: owner pieces into one NaN-poisoned buffer, partial sums added, COO slices tiling the whole.
usage: range_model_check.py FIRST_SEED COUNT [mixed|unit|blocks] [NPTS] [--poison] [--prebuild]   — one line per seed, BAD at the end of a line
that fails; a GPU fault kills the process (run chunks under `timeout`, ONE process at a time).  --poison: every VGPR / AGPR of
the chip is filled with NaN before each model's callbacks (tests/poison.py): a kernel reading a lane it never wrote then fails
every time, not when the stale contents happen to matter."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("examodels.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
import randexpr  # noqa: E402
from conftest import RankReplay  # noqa: E402
from exahip import CompressedExaModel, ExaModel, capi  # noqa: E402

POISON = "--poison" in sys.argv
PREBUILD = "--prebuild" in sys.argv       # on a machine without a GPU: plan + compile every seed's modules into the kernel cache
argv = [a for a in sys.argv if a not in ("--poison", "--prebuild")]
first, count = int(argv[1]), int(argv[2])
flavour = argv[3] if len(argv) > 3 else "mixed"
npts_arg = int(argv[4]) if len(argv) > 4 else 0
poison = (lambda: None)
if POISON:
    import tempfile
    from poison import make_poison
    poison = make_poison(tempfile.mkdtemp())
dev = torch.device("cpu" if PREBUILD else "cuda:0")
TOL = 1e-9


def rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if a.shape != ref.shape:
        return 1.0
    fin = np.isfinite(ref)
    if not fin.any():
        return 0.0
    if not np.all(np.isfinite(a[fin])):
        return 1.0
    scale = np.maximum(np.abs(ref[fin]), 1e-3 * max(1.0, float(np.max(np.abs(ref[fin])))))
    return float(np.max(np.abs(a[fin] - ref[fin]) / scale))


def with_env(env, make):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return make()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def densify(rows, cols, vals, nrow, ncol):
    d = np.zeros(nrow * ncol)
    np.add.at(d, (np.asarray(cols) - 1) * nrow + (np.asarray(rows) - 1), np.where(np.isfinite(vals), vals, 0.0))
    return d


bad = 0
for seed in range(first, first + count):
    npts = npts_arg or (300, 1000, 1037, 4099, 20011)[seed % 5]
    mk = lambda: randexpr.build_range_model(seed, npts=npts, unit=flavour in ("unit", "blocks"), blocks=flavour == "blocks")      # noqa: E731
    if PREBUILD:
        ExaModel(mk(), device=False).compile()
        print(f"seed {seed} {flavour} npts {npts} prebuilt", flush=True)
        continue
    m = ExaModel(mk())
    o = oracle.OracleModel(m.ir)
    nvar, ncon = m.meta.nvar, m.meta.ncon
    x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, nvar)
    y = np.random.default_rng(seed + 1).standard_normal(ncon)
    v = np.random.default_rng(seed + 2).standard_normal(nvar)
    w = np.random.default_rng(seed + 3).standard_normal(ncon)
    xd, yd, vd, wd = (torch.from_numpy(a).to(dev) for a in (x, y, v, w))
    R = {"obj": o.obj(x), "cons": o.cons(x), "grad": o.grad(x), "jac": o.jac_coord(x), "hess": o.hess_coord(x, y, 0.7),
         "jprod": o.jprod(x, v), "jtprod": o.jtprod(x, w), "hprod": o.hprod(x, y, v, 0.7)}
    E = {}
    poison()
    E["obj"] = abs(m.obj(x) - R["obj"]) / max(1.0, abs(R["obj"])) if np.isfinite(R["obj"]) else 0.0
    E["cons"] = rel(m.cons(x), R["cons"]); E["grad"] = rel(m.grad(x), R["grad"])
    E["jac"] = rel(m.jac_coord(x), R["jac"]); E["hess"] = rel(m.hess_coord(x, y, 0.7), R["hess"])
    E["jprod"] = rel(m.jprod(x, v), R["jprod"]); E["jtprod"] = rel(m.jtprod(x, w), R["jtprod"]); E["hprod"] = rel(m.hprod(x, y, v, 0.7), R["hprod"])
    ok_struct = all(np.array_equal(a, b) for a, b in zip(m.jac_structure() + m.hess_structure(), o.jac_structure() + o.hess_structure()))
    E["struct"] = 0.0 if ok_struct else 1.0
    # all five from one sweep
    poison()
    f, g, c, j, h = m.eval_all(xd, yd, 0.7)
    torch.cuda.synchronize()
    E["all"] = max(rel(g.cpu().numpy(), R["grad"]), rel(c.cpu().numpy()[:ncon], R["cons"]), rel(j.cpu().numpy()[:m.meta.nnzj], R["jac"]),
                   rel(h.cpu().numpy()[:m.meta.nnzh], R["hess"]), abs(f.item() - R["obj"]) / max(1.0, abs(R["obj"])) if np.isfinite(R["obj"]) else 0.0)
    # the three hess_coord! kernels
    h0 = m.hess_coord(xd, yd, 0.7).cpu().numpy()
    kinds = []
    for var in (0, 1, 2):
        mv = with_env({"EXAHIP_HESS_VARIANT": str(var)}, lambda: ExaModel(mk()))
        kinds.append(mv._L.exa_hess_variant(mv.id))
        out = torch.full((m.meta.nnzh + 8,), float("nan"), dtype=torch.float64, device=dev)
        poison()
        mv.hess_coord(xd, yd, 0.7, out=out)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        E[f"hv{var}"] = max(rel(got[:m.meta.nnzh], R["hess"]), 0.0 if np.all(np.isnan(got[m.meta.nnzh:])) else 1.0)
        # the three kernels evaluate the same expressions; the compiler may contract a*b+c into an FMA at different places of the
        # differently shaped kernels, which shows in entries that are sums of cancelling terms (round 4, seed 44 blocks: every
        # variant within 1e-9 of the oracle, one such entry 1e-12 apart between two kernels): norm-wise 1e-12 between the kernels
        E[f"hv{var}_vs0"] = rel(got[:m.meta.nnzh], h0) * 1e3           # (TOL is 1e-9)
        del mv
    # products by the other implementations
    info = (m.product_info("jtprod")[1], m.product_info("hprod")[1])
    for mode in (0, 1, 3):
        try:
            m.set_product_mode(mode, mode)
        except capi.ExaHipError:
            continue
        poison()
        E[f"jt{mode}"] = rel(m.jtprod(x, w), R["jtprod"]); E[f"hp{mode}"] = rel(m.hprod(x, y, v, 0.7), R["hprod"])
    m.set_product_mode(-1, -1)
    # compressed COO
    cm = CompressedExaModel(m)
    poison()
    for kind, nrow in (("jac", max(ncon, 1)), ("hess", nvar)):
        r_, c_ = o.jac_structure() if kind == "jac" else o.hess_structure()
        want = densify(r_, c_, R[kind], nrow, nvar)
        cr, cc = cm.jac_structure() if kind == "jac" else cm.hess_structure()
        cv = (cm.jac_coord(xd) if kind == "jac" else cm.hess_coord(xd, yd, 0.7)).cpu().numpy()
        n = cm.meta.nnzj if kind == "jac" else cm.meta.nnzh
        got = densify(np.asarray(cr.cpu() if hasattr(cr, "cpu") else cr)[:n], np.asarray(cc.cpu() if hasattr(cc, "cpu") else cc)[:n], cv[:n], nrow, nvar)
        E["c" + kind] = rel(got, want)
    # 3 ranks on this GPU
    W = 3
    rr = RankReplay(m, [("grad", nvar), ("cons", ncon), ("jprod", ncon), ("jtprod", nvar), ("hprod", nvar)], dev)
    jz = torch.full((max(1, m.meta.nnzj),), float("nan"), dtype=torch.float64, device=dev)
    hz = torch.full((max(1, m.meta.nnzh),), float("nan"), dtype=torch.float64, device=dev)
    objs = 0.0
    try:
        for r in range(W):
            m.set_shard(r, W)
            poison()
            objs += m.obj(x)
            rr.add("grad", lambda out: m.grad(xd, out=out)); rr.add("cons", lambda out: m.cons(xd, out=out))
            rr.add("jprod", lambda out: m.jprod(xd, vd, out=out)); rr.add("jtprod", lambda out: m.jtprod(xd, wd, out=out))
            rr.add("hprod", lambda out: m.hprod(xd, yd, vd, 0.7, out=out))
            m.jac_coord(xd, out=jz); m.hess_coord(xd, yd, 0.7, out=hz)
    finally:
        m.set_shard(0, 1)
    E["shard"] = max([rel(rr.result(k), R[k]) for k in ("grad", "cons", "jprod", "jtprod", "hprod")] +
                     [rel(jz.cpu().numpy()[:m.meta.nnzj], R["jac"]), rel(hz.cpu().numpy()[:m.meta.nnzh], R["hess"]),
                      abs(objs - R["obj"]) / max(1.0, abs(R["obj"])) if np.isfinite(R["obj"]) else 0.0])
    worst = max(E.values())
    flag = "ok" if worst <= TOL else "BAD"
    bad += flag == "BAD"
    fails = " ".join(f"{k}={e:.1e}" for k, e in E.items() if not e <= TOL)
    print(f"seed {seed} {flavour} npts {npts} products {info[0]!r}/{info[1]!r} hess kernels {kinds} maxerr {worst:.2e} {fails} {flag}", flush=True)
    del m, cm
print("BAD seeds:", bad)
