import sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/examodels.jl_amd')
from exahip import CompressedExaModel, ExaModel, models
from zoo import ZOO
import paramzoo, edgezoo
cases = {k: v for k, v in ZOO.items()}
cases["rocket2000"] = lambda: models.rocket_model(2000)
cases["lv5000"] = lambda: models.luksan_vlcek_model(5000)
for name, f in cases.items():
    try:
        m = ExaModel(f()); cm = CompressedExaModel(m)
        print(name, "J", cm.path("jac"), "H", cm.path("hess"), flush=True)
    except Exception as e:
        print(name, "ERR", e)
import time, numpy as np, torch
for name, f in (("rocket1e6", lambda: models.rocket_model(1_000_000)), ("lv1e7", lambda: models.luksan_vlcek_model(10_000_000))):
    m = ExaModel(f()); t0 = time.time(); cm = CompressedExaModel(m); print(name, "setup", round(time.time() - t0, 3), cm.path("jac"), cm.path("hess"), flush=True)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(m.meta.x0 + 0.01 * np.cos(np.arange(m.meta.nvar))).to(dev); y = torch.from_numpy(1.0 + 0.1*np.sin(np.arange(m.meta.ncon))).to(dev)
    for which in ("hess", "jac"):
        f = (lambda: cm.hess_coord(x, y, 0.7)) if which == "hess" else (lambda: cm.jac_coord(x))
        out = f(); torch.cuda.synchronize()
        for _ in range(20): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(50): f()
        torch.cuda.synchronize(); print(" ", which, "ms", round((time.time() - t0) / 50 * 1e3, 4), flush=True)
