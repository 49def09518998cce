#!/usr/bin/env python
"""Config 5 size on ONE GPU: LV N=1e8 (nnzh = 9e8, 7.2 GB of COO): timing + window parity against small models
(window locality of Luksan-Vlcek: the slots of points [a, a+n) depend only on x[a : a+n+2], y[a : a+n])."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "examodels.jl_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from exahip import ExaModel, models  # noqa: E402

N = 100_000_000
m = ExaModel(models.luksan_vlcek_model(N))
dev = torch.device("cuda:0")
x = m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)
y = np.random.default_rng(1).standard_normal(N - 2)
xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
h = torch.empty(m.meta.nnzh, dtype=torch.float64, device=dev)
m.hess_coord(xd, yd, 0.5, out=h)
torch.cuda.synchronize()
for _ in range(5):
    m.time_callback("hess", 10, xd, yd, 0.5, out=h)
ms = m.time_callback("hess", 30, xd, yd, 0.5, out=h)
print(f"LV N=1e8: nnzh={m.meta.nnzh} hess {ms:.4f} ms/eval  {m.meta.nnzh / ms / 1e6:.1f} Gnnz/s  {88.0 * N / ms / 1e6:.0f} GB/s")
worst = 0.0
for a in (0, 12_345_678, 49_999_000, N - 1000 - 2):
    n = 1000
    o = oracle.OracleModel(models.luksan_vlcek_model(n + 2).to_ir())
    ref = o.hess_coord(x[a:a + n + 2], y[a:a + n], 0.5)
    got = h[6 * a:6 * (a + n)].cpu().numpy()
    worst = max(worst, float(np.max(np.abs(got - ref[:6 * n]) / np.maximum(1.0, np.abs(ref[:6 * n])))))
    # objective block: points I = a .. a+n (i = I+2) -> slots 6(N-2) + 3I
    ob = 6 * (N - 2)
    got_o = h[ob + 3 * a: ob + 3 * (a + n)].cpu().numpy()
    ref_o = ref[6 * n: 6 * n + 3 * n]
    worst = max(worst, float(np.max(np.abs(got_o - ref_o) / np.maximum(1.0, np.abs(ref_o)))))
print("window parity max rel err", worst)
assert worst < 1e-10
rows = torch.empty(m.meta.nnzh, dtype=torch.int32, device=dev)
cols = torch.empty(m.meta.nnzh, dtype=torch.int32, device=dev)
m.hess_structure(rows, cols)
torch.cuda.synchronize()
assert bool(torch.all(rows >= cols)) and int(rows.max()) == N and int(cols.min()) == 1
print("structure ok (int32, lower triangle, max row = N)")
