import os, sys
sys.path.insert(0, "examodels.jl_amd")
import numpy as np, torch
from exahip import ExaModel, models
N = 10_000_000
res = {}
for k in ("1", "2"):
    os.environ["EXAHIP_FAST_TRIG"] = k
    m = ExaModel(models.luksan_vlcek_model(N))
    x = torch.from_numpy(m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)).cuda()
    c = torch.empty(m.meta.ncon, dtype=torch.float64, device="cuda")
    for _ in range(50): m.time_callback("cons", 20, x, out=c)
    t = min(m.time_callback("cons", 200, x, out=c) for _ in range(5))
    o = min(m.time_callback("obj", 200, x) for _ in range(5))
    res[k] = c.cpu().numpy().copy()
    print(f"EXAHIP_FAST_TRIG={k}: LV 1e7 cons_nln! {t:.4f} ms  obj {o:.4f} ms", flush=True)
    del m
d = np.abs(res["1"] - res["2"]) / np.maximum(np.abs(res["1"]), 1e-300)
print("max component-wise relative difference of cons between the two:", float(d.max()))
