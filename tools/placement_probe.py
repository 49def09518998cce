#!/usr/bin/env python
"""Does the PLACEMENT of the output buffer change the speed of hess_coord! beyond the MALL?  (the same code
differs by 7-9 % between two output buffers of one process.)  One model (LV, default N = 3e7: 2.16 GB of COO), one kernel;
the output pointer is moved (a) through one large allocation in steps from 64 B to 1 GiB, (b) over separately allocated
buffers whose addresses are printed modulo 4 KiB / 2 MiB / 1 GiB.  A/B/A/B rounds, minimum per placement.
usage: placement_probe.py [N] > profiles/r3_placement_probe.txt"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30_000_000
m = ExaModel(models.luksan_vlcek_model(N))
dev = torch.device("cuda:0")
x = torch.from_numpy(m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)).to(dev)
y = torch.from_numpy(np.random.default_rng(1).standard_normal(N - 2)).to(dev)
nnzh = m.meta.nnzh
L = m._L
L.exa_set_stream(m.id, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
print(f"LV N={N} nnzh={nnzh} ({8 * nnzh / 1e9:.2f} GB), kernel {('exa_hess', 'exa_hesscl', 'exa_hessc')[L.exa_hess_variant(m.id)]}", flush=True)


def t(ptr, reps):
    ms = ctypes.c_float(0.0)
    rc = L.exa_time_callback(m.id, 4, reps, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), 0.5, ctypes.c_void_p(ptr), ctypes.addressof(ms))
    assert rc == 0
    return ms.value


reps = 20 if N <= 4e7 else 6
slack = 1 << 31
big = torch.empty((8 * nnzh + slack) // 8 + 8, dtype=torch.float64, device=dev)
base = big.data_ptr()
print(f"(a) one allocation at {base:#x} (mod 2 MiB = {base % (1 << 21):#x}, mod 1 GiB = {base % (1 << 30):#x})")
offs = [0, 64, 128, 256, 512, 1024, 2048, 3072, 4096, 8192, 65536, 1 << 20, 1 << 21, (1 << 21) + 4096, 1 << 24, 1 << 28, 1 << 30, (1 << 30) + (1 << 21)]
for _ in range(3):
    t(base, reps)
res = {o: [] for o in offs}
for rnd in range(4):
    for o in offs:
        res[o].append(t(base + o, reps))
for o in offs:
    print(f"    offset {o:>12d} B  min {min(res[o]):.4f}  med {float(np.median(res[o])):.4f} ms   ({8 * (nnzh + 2 * N) / min(res[o]) / 1e6:.0f} GB/s algorithmic)", flush=True)
del big
torch.cuda.empty_cache()
print("(b) separately allocated buffers (torch caching allocator emptied in between; hipMalloc places them)")
bufs = []
for k in range(6):
    pad = torch.empty(int((k * 37 + 5) * 1e6), dtype=torch.uint8, device=dev)      # perturb the allocator's layout
    b = torch.empty(nnzh, dtype=torch.float64, device=dev)
    bufs.append((pad, b))
res = {k: [] for k in range(len(bufs))}
for rnd in range(4):
    for k, (_, b) in enumerate(bufs):
        res[k].append(t(b.data_ptr(), reps))
for k, (_, b) in enumerate(bufs):
    p = b.data_ptr()
    print(f"    buffer {k} at {p:#x}  mod 4 KiB {p % 4096:>5d}  mod 2 MiB {p % (1 << 21):>8d}  mod 1 GiB {p % (1 << 30):>11d}   min {min(res[k]):.4f}  med {float(np.median(res[k])):.4f} ms", flush=True)
