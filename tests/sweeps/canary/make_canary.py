#!/usr/bin/env python
"""Produces the standalone reproducer of an over-sized window kernel's wrong sums, and checks the library's answer to it.
(Round 5: the fault is root-caused — REPORT.md in this directory — and the library's base flags now remove its cause; this script still
records the launch from the build with the round-4 fallback flags and compares the compiler's default against it.)
TEST INFRASTRUCTURE (tests/sweeps/): imports the oracle as the checker.  Runs on a GPU box:

    python tests/sweeps/canary/make_canary.py [OUTDIR]          (default: this directory)

1. the random range model of profiles/NOTES.md (seed 1, blocks flavour, 1000 points): its exa_hprodw inlines 12 pattern
   evaluations and needs 256 architectural VGPRs + AGPRs.  Built by the library AS SHIPPED (every compiled kernel is asked what it
   needs, over-sized modules are compiled again with $EXAHIP_SAFE_FLAGS): Hv by windows must equal the oracle's, register
   poison included; exa_build_audit must say "safe" for the product module.
2. the same with EXAHIP_SAFE_FLAGS=none (the default allocator, what round 3 fenced off by giving the windows up): reported —
   this is the wrong build when the compiler still has the fault.
3. one launch of exa_hprodw recorded into hprodw_canary.dump (+ .hip: the module's source) with the output of the SAFE build as
   the expected values; canary_host.cpp replays it against code objects compiled from that source by hipcc with the default
   flags and with the safe flags (run_canary.sh): default DIFFERENT + safe equal = the fault is there and the fallback works;
   both equal = the compiler no longer has the fault on this kernel.
"""
import ctypes
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
for p in ("examodels.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
import randexpr  # noqa: E402
from exahip import ExaModel  # noqa: E402
from poison import make_poison  # noqa: E402

# keep the FIRST plan of the product windows (all-points sums inside the window kernel: the 12-pass exa_hprodw of round 3 with 256
# VGPRs + 84 AGPRs) instead of re-planning it into something smaller first: the flags alone must make it right
os.environ["EXAHIP_WINDOW_REPLAN"] = "0"
out = sys.argv[1] if len(sys.argv) > 1 else HERE
os.makedirs(out, exist_ok=True)
dev = torch.device("cuda:0")
poison = make_poison(tempfile.mkdtemp())
SEED, NPTS = 1, 1000
mk = lambda: randexpr.build_range_model(SEED, npts=NPTS, unit=True, blocks=True)      # noqa: E731


def check(label):
    m = ExaModel(mk())
    o = oracle.OracleModel(m.ir)
    x = m.meta.x0 + 0.05 * np.random.default_rng(SEED).uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(SEED + 1).standard_normal(m.meta.ncon)
    v = np.random.default_rng(SEED + 2).standard_normal(m.meta.nvar)
    ref = o.hprod(x, y, v, 0.7)
    mode, why = m.product_info("hprod")
    worst = 0.0
    for rep in range(3):
        poison()
        got = m.hprod(x, y, v, 0.7)
        bad = ~np.isfinite(got) | (np.abs(got - ref) > 1e-9 * np.maximum(1.0, np.abs(ref)))
        worst = max(worst, float(bad.sum()))
    aud = [a for a in m.build_audit() if a["module"] == "products"]
    big = [(a["kernel"], a["vgpr"], a["agpr"], a["scratch"]) for a in aud if a["fits"] is False]
    flags = sorted({a["flags"] for a in aud})
    print(f"{label}: Hv mode {mode} ({why}); product module flags {flags}; over-sized kernels {big}; entries off the oracle (worst of 3 poisoned runs): {int(worst)} of {m.meta.nvar}", flush=True)
    return m, (x, y, v), int(worst), flags


m, (x, y, v), nbad, flags = check("library as shipped")
assert nbad == 0, "the shipped build is wrong"
assert flags == ["safe"], "expected the product module to have been rebuilt with the safe flags"
xd, yd, vd = (torch.from_numpy(a).to(dev) for a in (x, y, v))
path = os.path.join(out, "hprodw_canary.dump")
rc = m._L.exa_debug_dump_window_launch(m.id, 1, ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(yd.data_ptr()), ctypes.c_void_p(vd.data_ptr()), 0.7, path.encode())
assert rc == 0, m._L.exa_last_error()
os.replace(path + ".hip", os.path.join(out, "hprodw_canary.hip"))
print("recorded", path, os.path.getsize(path), "bytes; source", os.path.getsize(os.path.join(out, "hprodw_canary.hip")), "bytes", flush=True)
del m
os.environ["EXAHIP_SAFE_FLAGS"] = "none"
os.environ["EXAHIP_CACHE_DIR"] = tempfile.mkdtemp()       # (the note "safe" of the shipped cache would not apply anyway: no fallback)
_, _, nbad_default, flags = check("EXAHIP_SAFE_FLAGS=none (default allocator)")
print("default-allocator build:", "WRONG (the fault is present)" if nbad_default else "right (the fault does not show in the hiprtc build of this process)")
subprocess.call(["bash", os.path.join(HERE, "run_canary.sh"), out])
