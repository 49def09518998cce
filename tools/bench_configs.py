#!/usr/bin/env python
"""Times all five callbacks on BASELINE.json configs 2-4 (1 GPU) with hipEvents on the launch stream and prints one
JSON object per config: ms per call, algorithmic bytes, achieved GB/s, fraction of the 8 TB/s HBM peak.
Secondary to bench.py (which is the contract line for config 2)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

PEAK = 8000.0


def iterator_bytes(m):
    """bytes of iterator columns read per full pass (8 B per stored column entry)."""
    tot = 0
    for k in range(m.npatterns):
        pat = m.ir.patterns[k]
        for c in range(pat.n_cols):
            if pat.cols[c].type != 2:
                tot += 8 * pat.n
    return tot


def run(name, core, reps=50):
    m = ExaModel(core)
    dev = torch.device("cuda:0")
    r = np.random.default_rng(0)
    x = torch.from_numpy(m.meta.x0 + 0.1 * r.uniform(-1, 1, m.meta.nvar)).to(dev)
    y = torch.from_numpy(np.random.default_rng(1).standard_normal(m.meta.ncon)).to(dev)
    bufs = {"obj": None, "cons": torch.empty(m.meta.ncon, dtype=torch.float64, device=dev),
            "grad": torch.empty(m.meta.nvar, dtype=torch.float64, device=dev),
            "jac": torch.empty(m.meta.nnzj, dtype=torch.float64, device=dev),
            "hess": torch.empty(m.meta.nnzh, dtype=torch.float64, device=dev)}
    itb = iterator_bytes(m)
    alg = {"obj": 8 * m.meta.nvar + itb, "cons": 8 * m.meta.ncon + 8 * m.meta.nvar + itb,
           "grad": 16 * m.meta.nvar + itb, "jac": 8 * m.meta.nnzj + 8 * m.meta.nvar + itb,
           "hess": 8 * m.meta.nnzh + 8 * m.meta.nvar + 8 * m.meta.ncon + itb}
    out = {"config": name, "nvar": m.meta.nvar, "ncon": m.meta.ncon, "nnzj": m.meta.nnzj, "nnzh": m.meta.nnzh,
           "npatterns": m.npatterns, "callbacks": {}}
    for cb in ("obj", "cons", "grad", "jac", "hess"):
        m.time_callback(cb, 5, x, y, 0.5, out=bufs[cb])
        ms = min(m.time_callback(cb, reps, x, y, 0.5, out=bufs[cb]) for _ in range(3))
        gbs = alg[cb] / (ms * 1e-3) / 1e9
        out["callbacks"][cb] = {"ms": ms, "algorithmic_bytes": alg[cb], "GBps": gbs, "frac_of_8TBps": gbs / PEAK,
                                "evals_per_s": 1e3 / ms}
    # fused obj + cons + jac + hess sweep (exa_eval_fused) vs the sum of the four separate callbacks
    c, j, h = bufs["cons"], bufs["jac"], bufs["hess"]
    for _ in range(5):
        m.eval_fused(x, y, 0.5, c=c, jac=j, hess=h)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            m.eval_fused(x, y, 0.5, c=c, jac=j, hess=h)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    sep = sum(out["callbacks"][k]["ms"] for k in ("obj", "cons", "jac", "hess"))
    fb = 8 * (m.meta.nnzh + m.meta.nnzj + m.meta.ncon) + 8 * m.meta.nvar + 8 * m.meta.ncon + itb
    out["fused_obj_cons_jac_hess"] = {"ms": best, "separate_ms": sep, "algorithmic_bytes": fb, "GBps": fb / (best * 1e-3) / 1e9}
    # matrix-free products (secondary)
    v = torch.from_numpy(np.random.default_rng(2).standard_normal(m.meta.nvar)).to(dev)
    w = torch.from_numpy(np.random.default_rng(3).standard_normal(m.meta.ncon)).to(dev)
    prods = {}
    for name, fn in (("jprod", lambda: m.jprod(x, v, out=bufs["cons"])), ("jtprod", lambda: m.jtprod(x, w, out=bufs["grad"])),
                     ("hprod", lambda: m.hprod(x, y, v, 0.5, out=bufs["grad"]))):
        for _ in range(3):
            fn()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        prods[name + "_ms"] = e0.elapsed_time(e1) / 20
    out["products"] = prods
    # one solver iteration's evaluation set (fused sweep + grad!) eagerly vs replayed from ONE hipGraph (torch.cuda.graph)
    fo = torch.zeros(1, dtype=torch.float64, device=dev)

    def evaluate():
        m.eval_fused(x, y, 0.5, c=c, jac=j, hess=h, obj_out=fo)
        m.grad(x, out=bufs["grad"])

    def evaluate_all():
        m.eval_all(x, y, 0.5, g=bufs["grad"], c=c, jac=j, hess=h, obj_out=fo)

    def timed(fn, n=50):
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n)
        return best

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        evaluate()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        evaluate()
    out["iteration_set_fused_plus_grad"] = {"eager_ms": timed(evaluate), "hip_graph_ms": timed(graph.replay), "eval_all_ms": timed(evaluate_all)}
    # duplicate-summed COO (CompressedNLPModel): one-off set-up (device radix sorts) and the per-evaluation cost
    import time
    from exahip import CompressedExaModel
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cm = CompressedExaModel(m)
    torch.cuda.synchronize()
    setup_ms = 1e3 * (time.perf_counter() - t0)
    ch = torch.empty(cm.meta.nnzh, dtype=torch.float64, device=dev)
    cj = torch.empty(cm.meta.nnzj, dtype=torch.float64, device=dev)
    out["compressed"] = {"setup_ms": setup_ms, "cnnzj": cm.meta.nnzj, "cnnzh": cm.meta.nnzh,
                         "path_hess": cm.path("hess"), "path_jac": cm.path("jac"),
                         "chess_ms": timed(lambda: cm.hess_coord(x, y, 0.5, out=ch), 20),
                         "cjac_ms": timed(lambda: cm.jac_coord(x, out=cj), 20)}
    # the reference's scheme on the same model: uncompressed evaluation + sorted gather
    import os
    os.environ["EXAHIP_CWINDOW"] = "0"
    cm0 = CompressedExaModel(m)
    del os.environ["EXAHIP_CWINDOW"]
    out["compressed"]["gather_chess_ms"] = timed(lambda: cm0.hess_coord(x, y, 0.5, out=ch), 20)
    out["compressed"]["gather_cjac_ms"] = timed(lambda: cm0.jac_coord(x, out=cj), 20)
    out["hess_nnz_per_s"] = m.meta.nnzh * out["callbacks"]["hess"]["evals_per_s"]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["lv", "rocket", "acopf"]
    if "lv" in which:
        run("config2: LuksanVlcek N=1e7", models.luksan_vlcek_model(10_000_000))
    if "rocket" in which:
        run("config3: Goddard rocket nh=1e6", models.rocket_model(1_000_000))
    if "acopf" in which:
        run("config4: ACOPF 78484-bus-scale synthetic", models.ac_power_model(models.synthetic_power_data(78_484, 126_015, 6_800, seed=0)))
