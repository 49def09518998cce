#!/usr/bin/env python
"""Diagnostic for a random range model whose CHAINED hess_coord! kernel (exa_hessc, EXAHIP_HESS_VARIANT=2) disagrees with exa_hess
(tests/sweeps/range_model_check.py found seed 44, blocks flavour, 20011 points in round 4): per build variant — the fallback flags
of the library, none, other allocator flags — the kernel's resources (exa_build_audit) and how many Hessian entries are off the
oracle / off exa_hess, with and without register poison.  TEST INFRASTRUCTURE.
usage: hessc_seed_diag.py SEED [FLAVOUR=blocks] [NPTS=20011]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("examodels.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
import randexpr  # noqa: E402
from exahip import ExaModel  # noqa: E402
from poison import make_poison  # noqa: E402

seed = int(sys.argv[1])
flavour = sys.argv[2] if len(sys.argv) > 2 else "blocks"
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 20011
poison = make_poison(tempfile.mkdtemp())
mk = lambda: randexpr.build_range_model(seed, npts=npts, unit=flavour in ("unit", "blocks"), blocks=flavour == "blocks")      # noqa: E731
dev = torch.device("cuda:0")
ref_m = None
for label, env in (("variant 0 (exa_hess)", {"EXAHIP_HESS_VARIANT": "0"}),
                   ("variant 2, library default (safe flags where over-sized)", {"EXAHIP_HESS_VARIANT": "2"}),
                   ("variant 2, EXAHIP_SAFE_FLAGS=none", {"EXAHIP_HESS_VARIANT": "2", "EXAHIP_SAFE_FLAGS": "none"}),
                   ("variant 2, safe = -sgpr-regalloc=fast", {"EXAHIP_HESS_VARIANT": "2", "EXAHIP_SAFE_FLAGS": "-mllvm -sgpr-regalloc=fast"}),
                   ("variant 2, safe = -O1", {"EXAHIP_HESS_VARIANT": "2", "EXAHIP_SAFE_FLAGS": "-O1"}),
                   ("variant 2, EXAHIP_GROUP=0", {"EXAHIP_HESS_VARIANT": "2", "EXAHIP_GROUP": "0"})):
    saved = {k: os.environ.get(k) for k in ("EXAHIP_HESS_VARIANT", "EXAHIP_SAFE_FLAGS", "EXAHIP_GROUP", "EXAHIP_CACHE_DIR")}
    os.environ.update(env)
    os.environ["EXAHIP_CACHE_DIR"] = tempfile.mkdtemp()
    try:
        m = ExaModel(mk())
        if ref_m is None:
            o = oracle.OracleModel(m.ir)
            x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, m.meta.nvar)
            y = np.random.default_rng(seed + 1).standard_normal(m.meta.ncon)
            R = o.hess_coord(x, y, 0.7)
            xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
            ref_m = True
        aud = {a["kernel"]: a for a in m.build_audit() if a["module"] == "model"}
        k = aud.get("exa_hessc") or aud.get("exa_hess")
        line = f"{label}: kernel in use {m._L.exa_hess_variant(m.id)}; exa_hess {aud['exa_hess']['vgpr']}v/{aud['exa_hess']['agpr']}a"
        if "exa_hessc" in aud:
            a = aud["exa_hessc"]
            line += f"; exa_hessc {a['vgpr']}v/{a['agpr']}a/{a['scratch']}B scratch/{a['sgpr_spill']} sgpr spills flags {a['flags']}"
        for pz in (False, True):
            out = torch.full((m.meta.nnzh + 8,), float("nan"), dtype=torch.float64, device=dev)
            if pz:
                poison()
            m.hess_coord(xd, yd, 0.7, out=out)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            fin = np.isfinite(R)
            bad = ~np.isfinite(got[:m.meta.nnzh][fin]) | (np.abs(got[:m.meta.nnzh][fin] - R[fin]) > 1e-9 * np.maximum(np.abs(R[fin]), 1e-3 * max(1.0, float(np.max(np.abs(R[fin]))))))
            idx = np.flatnonzero(fin)[bad]
            where = ""
            if idx.size:
                pats = [m.pattern_info(p) for p in range(m.npatterns)]
                hit = {}
                for i in idx[:2000]:
                    for p, pi in enumerate(pats):
                        if pi["o2step"] and pi["o2"] <= i < pi["o2"] + pi["o2step"] * pi["n"]:
                            hit.setdefault(p, set()).add(int((i - pi["o2"]) // pi["o2step"]) // 256)
                where = "; bad by pattern -> tiles: " + ", ".join(f"p{p}: {sorted(t)[:12]}{'...' if len(t) > 12 else ''}" for p, t in sorted(hit.items()))
            line += f"; {'poisoned' if pz else 'plain'}: {int(bad.sum())} of {int(fin.sum())} entries off the oracle (nan {int(np.isnan(got[:m.meta.nnzh][fin]).sum())}), tail {'untouched' if np.all(np.isnan(got[m.meta.nnzh:])) else 'OVERWRITTEN'}{where}"
        print(line, flush=True)
        del m
    finally:
        for k_, v in saved.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
