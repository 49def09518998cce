"""What does a user-registered function cost against the table entry it imitates?  (the rules are pasted where the table's own
templates are, so the kernel STRUCTURE is the same; the arithmetic is whatever the rule text calls.)

LV N = 1e7, every callback, three spellings of the model's sin / exp:
  table     the built-in nodes (sin goes through the library's fused sincos: one range reduction for value and derivative)
  user-ocml registered "sin($1)" / "cos($1)" / "exp($1)": the HIP math library's routines, value and derivative reduced separately
  user-lib  registered with the library's own device routines (exa_sin / exa_cos of the prelude every module carries)
  fused     exa_register_univariate_fused: one statement per argument gives value and both derivatives ("exa_sincos($1, &$2, &$3); $4 = -$2;")
usage (GPU box): python tools/userfn_ab.py [N]"""
import sys
sys.path.insert(0, "examodels.jl_amd")
import numpy as np, torch
from exahip import ExaCore, ExaModel, graph as G, rng
from exahip.models import lv_x0

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000


def lv(sin, exp):
    c = ExaCore()
    x = c.add_var(N, start=lv_x0(N))
    c.add_con(lambda i: 3 * x[i + 1] ** 3 + 2 * x[i + 2] - 5 + sin(x[i + 1] - x[i + 2]) * sin(x[i + 1] + x[i + 2]) + 4 * x[i + 1]
              - x[i] * exp(x[i] - x[i + 1]) - 3, rng(1, N - 2))
    c.add_obj(lambda i: 100 * (x[i - 1] ** 2 - x[i]) ** 2 + (x[i - 1] - 1) ** 2, rng(2, N))
    return c


usin = G.register_univariate("ab_sin_ocml", "sin($1)", "cos($1)", "-$2")
uexp = G.register_univariate("ab_exp", "exp($1)", "$2", "$3")
lsin = G.register_univariate("ab_sin_lib", "exa_sin($1)", "exa_cos($1)", "-$2")
fsin = G.register_univariate("ab_sin_fused", fused="exa_sincos($1, &$2, &$3); $4 = -$2;")
fexp = G.register_univariate("ab_exp_fused", fused="$2 = exp($1); $3 = $2; $4 = $2;")
spell = {"table": (G.sin, G.exp), "user-ocml": (usin, uexp), "user-lib": (lsin, uexp), "fused": (fsin, fexp)}
ref = None
print(f"# LV N = {N:.0e}: ms per call (exa_time_callback, min of 5 x 200 calls), the same inputs for the four spellings")
print(f"{'':10s} {'obj':>8s} {'cons':>8s} {'grad':>8s} {'jac':>8s} {'hess':>8s}   max |difference| / |table| of cons, jac, hess")
for name, (s, e) in spell.items():
    m = ExaModel(lv(s, e))
    x = torch.from_numpy(m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)).cuda()
    y = torch.from_numpy(np.random.default_rng(1).standard_normal(m.meta.ncon)).cuda()
    outs = {"cons": torch.empty(m.meta.ncon, dtype=torch.float64, device="cuda"), "grad": torch.empty(N, dtype=torch.float64, device="cuda"),
            "jac": torch.empty(m.meta.nnzj, dtype=torch.float64, device="cuda"), "hess": torch.empty(m.meta.nnzh, dtype=torch.float64, device="cuda")}
    t = {}
    for cb in ("obj", "cons", "grad", "jac", "hess"):
        kw = dict(out=outs.get(cb))
        if cb == "hess":
            kw.update(y=y, obj_weight=0.5)
        for _ in range(3):
            m.time_callback(cb, 50, x, **kw)
        t[cb] = min(m.time_callback(cb, 200, x, **kw) for _ in range(5))
    got = {k: outs[k].cpu().numpy().copy() for k in ("cons", "jac", "hess")}
    if ref is None:
        ref, d = got, ""
    else:
        d = "  ".join(f"{k} {float(np.max(np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), 1e-300))):.1e}" for k in got)
    print(f"{name:10s} " + " ".join(f"{t[cb]:8.4f}" for cb in ("obj", "cons", "grad", "jac", "hess")) + "   " + d, flush=True)
    del m, outs
    torch.cuda.empty_cache()
