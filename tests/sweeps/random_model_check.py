#!/usr/bin/env python
"""One random model (tests/randexpr.py: `npat` patterns of depth `depth` over 300 data points) on the GPU against the
oracle: H*v and J'v by atomics (twice, interleaved: stale spill space shows on the second kernel), grad! both ways.
Prints one line; a GPU fault kills the process, so sweeps run one process per seed:

    (for s in $(seq 1000 1099); do echo "$s 12 6"; done) | xargs -P 6 -L 1 sh -c \\
        'timeout 300 python tests/sweeps/random_model_check.py $0 $1 $2 2>&1 | grep seed || echo "seed $0 $1 $2 CRASH"'

Entries whose Jacobian / Hessian values themselves differ from the oracle at 1e-6 (magnitudes of 1e20 and more: seeds
2005, 2034 at depth 5) are conditioning of the random expression, not of the kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("examodels.jl_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import randexpr  # noqa: E402
import oracle  # noqa: E402
from exahip import ExaModel  # noqa: E402

seed, npat, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
poison = (lambda: None)
if os.environ.get("POISON"):
    # every VGPR / AGPR of the chip filled with NaN before each group of calls (tests/poison.py): a kernel that reads a lane it
    # never wrote then fails every time
    import tempfile
    from poison import make_poison
    poison = make_poison(tempfile.mkdtemp())
randexpr.NPTS = 300
# USERFN=1: eleven table entries are taken from user-registered twins (randexpr.registered_twins: exa_register_univariate / _fused /
# _bivariate); the oracle, which knows no user function, evaluates the table's model of the same seed
user = randexpr.registered_twins() if os.environ.get("USERFN") else None
if os.environ.get("PREBUILD"):           # without a GPU: plan + compile into the kernel cache (which travels to the GPU box)
    ExaModel(randexpr.build_model(seed, npat, depth, user=user), device=False).compile()
    print(f"seed {seed} {npat} {depth} prebuilt")
    sys.exit(0)
m = ExaModel(randexpr.build_model(seed, npat, depth, user=user))
o = oracle.OracleModel(ExaModel(randexpr.build_model(seed, npat, depth), device=False).ir if user else m.ir)
m.set_product_mode(0, 0)
x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, m.meta.nvar)
y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
v = np.random.default_rng(seed + 2).standard_normal(m.meta.nvar)
w = np.random.default_rng(seed + 3).standard_normal(m.meta.ncon)


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    ok = np.isfinite(b)
    if not ok.any():
        return 0.0
    return float(np.max(np.abs(a[ok] - b[ok]) / (1.0 + np.abs(b[ok]))))


errs = []
for _ in range(2):
    poison()
    errs += [rel(m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7)), rel(m.jtprod(x, w), o.jtprod(x, w))]
poison()
errs += [rel(m.grad(x), o.grad(x))]
m.set_grad_mode(1)
errs += [rel(m.grad(x), o.grad(x))]
if os.environ.get("CHECK_ALL"):
    # the remaining callbacks: values, COO kernels, the fused sweep, the sorted products, the compressed COO
    import torch
    from exahip import CompressedExaModel
    poison()
    errs += [abs(m.obj(x) - o.obj(x)) / (1 + abs(o.obj(x))), rel(m.cons(x), o.cons(x)), rel(m.jac_coord(x), o.jac_coord(x)),
             rel(m.hess_coord(x, y, 0.7), o.hess_coord(x, y, 0.7)), rel(m.jprod(x, v), o.jprod(x, v))]
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    poison()
    f, c, j, h = m.eval_fused(xd, yd, 0.7)
    errs += [abs(float(f[0]) - o.obj(x)) / (1 + abs(o.obj(x))), rel(c.cpu().numpy(), o.cons(x)), rel(j.cpu().numpy(), o.jac_coord(x)),
             rel(h.cpu().numpy(), o.hess_coord(x, y, 0.7))]
    m.set_product_mode(1, 1)
    errs += [rel(m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7)), rel(m.jtprod(x, w), o.jtprod(x, w))]
    try:                                   # owner pull (data-indexed targets), where the model has it
        m.set_product_mode(3, 3)
        poison()
        errs += [rel(m.hprod(x, y, v, 0.7), o.hprod(x, y, v, 0.7)), rel(m.jtprod(x, w), o.jtprod(x, w))]
    except Exception as e:                 # noqa: BLE001
        if "pull" not in str(e) and "mode" not in str(e):
            raise
    m.set_product_mode(1, 1)
    cm = CompressedExaModel(m)
    poison()
    for kind, nrow in (("jac", max(m.meta.ncon, 1)), ("hess", m.meta.nvar)):
        r, cc = (o.jac_structure() if kind == "jac" else o.hess_structure())
        vals = o.jac_coord(x) if kind == "jac" else o.hess_coord(x, y, 0.7)
        dense = np.zeros(nrow * m.meta.nvar)
        np.add.at(dense, (cc - 1) * nrow + (r - 1), np.where(np.isfinite(vals), vals, 0.0))
        # (a matrix entry one of whose duplicates is NaN / Inf is NaN / Inf in the compressed COO — the sum of its duplicates — while `dense` above
        # holds the sum of the finite ones: such entries are not comparable and are left out (seed 8816: 26 of them))
        tainted = np.zeros(nrow * m.meta.nvar, dtype=bool)
        tainted[((cc - 1) * nrow + (r - 1))[~np.isfinite(vals)]] = True
        cr, ccol = (cm.jac_structure() if kind == "jac" else cm.hess_structure())
        cv = (cm.jac_coord(xd) if kind == "jac" else cm.hess_coord(xd, yd, 0.7)).cpu().numpy()
        got = np.zeros_like(dense)
        got[((ccol - 1) * nrow + (cr - 1)).cpu().numpy()] = np.where(np.isfinite(cv), cv, 0.0)
        errs += [rel(got[~tainted], dense[~tainted])]
if os.environ.get("SHARDS"):
    # the same model cut into SHARDS shards (one after the other on this GPU; host-pointer entry points: entries a rank does
    # not own come back as zeros, so owner pieces add up like partial sums): obj / grad! / cons_nln! /
    # products add up to the oracle's, the COO slices written at their global positions tile the unsharded COO
    W = int(os.environ["SHARDS"])
    m.set_product_mode(0, 0)
    m.set_grad_mode(0)
    acc = {k: 0.0 for k in ("obj", "grad", "cons", "jprod", "jtprod", "hprod", "jac", "hess")}
    for r in range(W):
        m.set_shard(r, W)
        acc["obj"] = acc["obj"] + m.obj(x)
        acc["grad"] = acc["grad"] + m.grad(x)
        acc["cons"] = acc["cons"] + m.cons(x)
        acc["jprod"] = acc["jprod"] + m.jprod(x, v)
        acc["jtprod"] = acc["jtprod"] + m.jtprod(x, w)
        acc["hprod"] = acc["hprod"] + m.hprod(x, y, v, 0.7)
        # (COO at global positions: slots of other ranks are left untouched — start from zeros)
        import torch
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        jz = torch.zeros(m.meta.nnzj, dtype=torch.float64, device="cuda"); hz = torch.zeros(m.meta.nnzh, dtype=torch.float64, device="cuda")
        m.jac_coord(xd, out=jz); m.hess_coord(xd, yd, 0.7, out=hz)
        acc["jac"] = acc["jac"] + jz.cpu().numpy(); acc["hess"] = acc["hess"] + hz.cpu().numpy()
    m.set_shard(0, 1)
    errs += [abs(acc["obj"] - o.obj(x)) / (1 + abs(o.obj(x))), rel(acc["grad"], o.grad(x)), rel(acc["cons"], o.cons(x)),
             rel(acc["jprod"], o.jprod(x, v)), rel(acc["jtprod"], o.jtprod(x, w)), rel(acc["hprod"], o.hprod(x, y, v, 0.7)),
             rel(acc["jac"], o.jac_coord(x)), rel(acc["hess"], o.hess_coord(x, y, 0.7))]
print("seed", seed, npat, depth, "maxerr %.2e" % max(errs), "BAD" if max(errs) > 1e-9 else "ok", ["%.1e" % e for e in errs], flush=True)
