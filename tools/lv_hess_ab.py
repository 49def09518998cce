#!/usr/bin/env python
"""hess_coord! of LV at N points with each of the three kernels (EXAHIP_HESS_VARIANT=0|1|2: exa_hess / exa_hessc / exa_hesscl), for ONE
package directory (argv[1]: this tree's examodels.jl_amd or a copy of an older one) — run once per (package, variant) in a fresh
process by the caller, so that two libraries never share a process.  usage: lv_hess_ab.py PKGDIR N VARIANT [--compile-only]"""
# An older tree to compare with:  mkdir tools/_old && git archive <commit> examodels.jl_amd include | tar -x -C tools/_old && make -C tools/_old/examodels.jl_amd/csrc
# (round 5 used ccdf976 as tools/_old_r5 and 396dbc3 as tools/_old_r5b; both removed after the measurements)
import os
import sys

pkg, N, var = sys.argv[1], int(float(sys.argv[2])), sys.argv[3]
os.environ["EXAHIP_HESS_VARIANT"] = var
sys.path.insert(0, pkg)
import numpy as np  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

if "--compile-only" in sys.argv:
    m = ExaModel(models.luksan_vlcek_model(1000), device=False)
    m.compile()
    print(pkg, var, m._L.exa_module_name(m.id).decode())
    sys.exit(0)
import torch  # noqa: E402

m = ExaModel(models.luksan_vlcek_model(N))
r = np.random.default_rng(0)
x = torch.from_numpy(m.meta.x0 + 0.1 * r.uniform(-1, 1, N)).cuda()
y = torch.from_numpy(r.standard_normal(m.meta.ncon)).cuda()
h = torch.empty(m.meta.nnzh, dtype=torch.float64, device="cuda")
for _ in range(20):
    m.hess_coord(x, y, 0.5, out=h)
reps = 200 if N <= 2e7 else 40
ts = [m.time_callback("hess", reps, x, y, 0.5, out=h) for _ in range(6)]
audit = {a["kernel"]: a["vgpr"] for a in m.build_audit() if a["kernel"] in ("exa_hess", "exa_hessc", "exa_hesscl")}
print(f"{os.path.relpath(pkg)} N={N:.0e} variant {var}: min {min(ts):.4f} median {sorted(ts)[3]:.4f} ms  ({8 * (m.meta.nnzh + m.meta.nvar + m.meta.ncon) / min(ts) / 1e6 / 8000:.3f} of 8 TB/s)  vgpr {audit}  sum {float(h.sum()):.12e}", flush=True)
