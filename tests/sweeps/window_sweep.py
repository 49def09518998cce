#!/usr/bin/env python
"""Sweep of random RANGE-iterated models (tests/randexpr.build_range_model: unit and stepped ranges, literal-index leaves,
several variable blocks) through the owner-computes product windows: J'v and Hv by windows against the oracle (1e-9: deep random
trees are ill-conditioned) and against the atomics, twice the same bits, and 3 ranks replayed into one buffer.
usage: window_sweep.py FIRST_SEED COUNT [unit|blocks]   — one line per seed, BAD at the end of a line that fails."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("examodels.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
import randexpr  # noqa: E402
from exahip import ExaModel, capi  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
flavour = sys.argv[3] if len(sys.argv) > 3 else ""
dev = torch.device("cuda:0")


def rel(a, ref):
    """max relative error over the entries where the reference is finite; where it is not (0 * Inf of a deep random tree) the
    result must be non-finite too"""
    fin = np.isfinite(ref)
    if not np.all(np.isfinite(a[fin])) or np.any(np.isfinite(a[~fin]) & ~np.isfinite(ref[~fin]) & False):
        return 1.0
    if not fin.any():
        return 0.0
    scale = np.maximum(np.abs(ref[fin]), 1e-3 * max(1.0, float(np.max(np.abs(ref[fin])))))
    return float(np.max(np.abs(a[fin] - ref[fin]) / scale))


bad = 0
for seed in range(first, first + count):
    m = ExaModel(randexpr.build_range_model(seed, npts=1000 + (0 if flavour == "blocks" else 37 * (seed % 5)), unit=flavour in ("unit", "blocks"), blocks=flavour == "blocks"))
    o = oracle.OracleModel(m.ir)
    x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(seed + 2).standard_normal(m.meta.nvar)
    w = np.random.default_rng(seed + 3).standard_normal(m.meta.ncon)
    xd, yd, vd, wd = (torch.from_numpy(a).to(dev) for a in (x, y, v, w))
    info = [m.product_info("jtprod"), m.product_info("hprod")]
    errs, notes = [], []
    for which, call, ref in (("jtprod", lambda out: m.jtprod(xd, wd, out=out), o.jtprod(x, w)), ("hprod", lambda out: m.hprod(xd, yd, vd, 0.7, out=out), o.hprod(x, y, v, 0.7))):
        try:
            m.set_product_mode(2 if which == "jtprod" else -1, 2 if which == "hprod" else -1)
        except capi.ExaHipError:
            notes.append(which + ": no windows")
            continue
        outs = []
        for _ in range(2):
            out = torch.full((m.meta.nvar,), float("nan"), dtype=torch.float64, device=dev)
            call(out)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
        errs.append(rel(outs[0], ref))
        if not np.array_equal(outs[0], outs[1], equal_nan=True):
            errs.append(1.0); notes.append(which + ": not reproducible")
        # 3 ranks into one buffer (where the model shards by window owner)
        buf = torch.full((m.meta.nvar,), float("nan"), dtype=torch.float64, device=dev)
        try:
            pieces = True
            for r in range(3):
                m.set_shard(r, 3)
                if m.shard_layout(which) != "pieces":
                    pieces = False
                    break
                call(buf)
            torch.cuda.synchronize()
        finally:
            m.set_shard(0, 1)
        if pieces:
            errs.append(rel(buf.cpu().numpy(), ref))
        m.set_product_mode(0 if which == "jtprod" else -1, 0 if which == "hprod" else -1)
        out = torch.empty(m.meta.nvar, dtype=torch.float64, device=dev)
        call(out)
        errs.append(rel(out.cpu().numpy(), outs[0]))
        m.set_product_mode(-1, -1)
    worst = max(errs) if errs else 0.0
    flag = "BAD" if not (worst <= 1e-9) else "ok"
    bad += flag == "BAD"
    print(f"seed {seed} {flavour or 'mixed'} {info[0][1]!r} / {info[1][1]!r} maxerr {worst:.2e} {' '.join(notes)} {flag}", flush=True)
print("BAD seeds:", bad)
