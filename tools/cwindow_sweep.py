"""Window-size sweep of the windowed compressed sweep (exa_chess / exa_cjac): `python tools/cwindow_sweep.py lv|rocket W...`
(W = 0: the size exa_compress picks; EXAHIP_CW_VERBOSE=1 prints the pass table).  Source of the W table in DESIGN.md §5."""
import sys, time, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/examodels.jl_amd')
import numpy as np, torch
from exahip import CompressedExaModel, ExaModel, models
which_model = sys.argv[1]
f = (lambda: models.rocket_model(1_000_000)) if which_model == "rocket" else (lambda: models.luksan_vlcek_model(10_000_000))
m = ExaModel(f())
dev = torch.device("cuda:0")
x = torch.from_numpy(m.meta.x0 + 0.01 * np.cos(np.arange(m.meta.nvar))).to(dev); y = torch.from_numpy(1.0 + 0.1*np.sin(np.arange(m.meta.ncon))).to(dev)
for W in sys.argv[2:]:
    if W != "0": os.environ["EXAHIP_CW_W"] = W
    cm = CompressedExaModel(m)
    ch = torch.empty(cm.meta.nnzh, dtype=torch.float64, device=dev); cj = torch.empty(cm.meta.nnzj, dtype=torch.float64, device=dev)
    r = []
    for which in ("hess", "jac"):
        f = (lambda: cm.hess_coord(x, y, 0.7, out=ch)) if which == "hess" else (lambda: cm.jac_coord(x, out=cj))
        f(); torch.cuda.synchronize()
        for _ in range(30): f()
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(40): f()
            torch.cuda.synchronize(); best = min(best, (time.time() - t0) / 40 * 1e3)
        r.append(round(best, 4))
    print(which_model, "W", W, "hess/jac ms", r, flush=True)
