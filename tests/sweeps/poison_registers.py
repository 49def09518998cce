#!/usr/bin/env python
"""Diagnostic for "wrong, run-to-run different" results of a kernel: a NaN pattern is left in EVERY architectural VGPR and AGPR of
the chip (a 512-register asm kernel over 4096 workgroups) before the callback under suspicion runs.  If its wrong entries turn
into NaN, the kernel reads registers (lanes) it never wrote — stale contents of whatever ran before.

Written for the reproducer of profiles/NOTES.md (random range model, seed 1, blocks flavour, 1000 points, Hv by windows: a
12-pass kernel with 256 + 74 registers).  The shipped library refuses that window kernel (window_kernels_spill); to reproduce,
lift the rule there.  Result of 2026-09: 995 wrong entries -> 995 NaN after the poison, and they stay NaN in later launches."""
import os, sys, subprocess, tempfile, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("examodels.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import oracle, randexpr
from exahip import ExaModel

# a kernel that leaves a NaN pattern in every architectural VGPR and every AGPR of the SIMDs it runs on
body = ["v_mov_b32 v255, 0x7ff80000"]
for k in range(1, 255):
    body.append(f"v_mov_b32 v{k}, 0x7ff80000")
for k in range(256):
    body.append(f"v_accvgpr_write_b32 a{k}, v255")
clob = ", ".join(f'"v{k}"' for k in range(1, 256)) + ", " + ", ".join(f'"a{k}"' for k in range(256))
src = '#include <hip/hip_runtime.h>\nextern "C" __global__ void __launch_bounds__(256) poison(int* out) {\n  asm volatile("' + "\\n".join(body) + '" ::: ' + clob + ');\n  if (out && threadIdx.x == 999) out[0] = 1;\n}\n'
td = tempfile.mkdtemp()
open(td + "/p.hip", "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O1", "-o", td + "/p.co", td + "/p.hip"])
hip = ctypes.CDLL("libamdhip64.so.7")
mod = ctypes.c_void_p(); assert hip.hipModuleLoadData(ctypes.byref(mod), open(td + "/p.co", "rb").read()) == 0
fn = ctypes.c_void_p(); assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"poison") == 0
def poison():
    nullp = ctypes.c_void_p(0)
    arr = (ctypes.c_void_p * 1)(ctypes.cast(ctypes.pointer(nullp), ctypes.c_void_p))
    st = torch.cuda.current_stream().cuda_stream
    assert hip.hipModuleLaunchKernel(fn, 4096, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(st), arr, None) == 0
    torch.cuda.synchronize()

seed = 1
m = ExaModel(randexpr.build_range_model(seed, npts=1000, unit=True, blocks=True))
o = oracle.OracleModel(m.ir)
nvar, ncon = m.meta.nvar, m.meta.ncon
x = m.meta.x0 + 0.05 * np.random.default_rng(seed).uniform(-1, 1, nvar)
y = np.random.default_rng(seed + 1).standard_normal(ncon)
v = np.random.default_rng(seed + 2).standard_normal(nvar)
ref = o.hprod(x, y, v, 0.7)
m.set_product_mode(-1, 2)
xd, yd, vd = (torch.from_numpy(a).cuda() for a in (x, y, v))
for rep, pz in enumerate([False, False, True, True, False, True]):
    if pz:
        poison()
    got = m.hprod(xd, yd, vd, 0.7).cpu().numpy()
    d = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3 * np.nanmax(np.abs(ref)))
    bad = ~(d <= 1e-9)
    print("RESULT poison" if pz else "RESULT plain ", "wrong entries", int(bad.sum()), "of which NaN", int(np.isnan(got).sum()), np.flatnonzero(bad)[:5], flush=True)
