#!/usr/bin/env python
"""The evaluation set of a solver iteration (obj, grad!, cons_nln!, jac_coord!, hess_coord! at one x — the call pattern of
test/NLPModelsIpoptLite.jl/src/NLPModelsIpoptLite.jl:28-40) launched eagerly and replayed from ONE captured hipGraph, as five
separate callbacks and as exa_eval_all.  On a model whose kernels take microseconds (ACOPF at case78484 scale) the launches are
most of the time; the callbacks never synchronise, so the host application can capture them (include/exahip.h "streams").

usage (GPU box): python tools/graph_iteration.py [CONFIG=4] > profiles/r3_graph_iteration_config4.txt"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from exahip import ExaModel  # noqa: E402


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    core = bench.build_core(config, {2: int(1e7), 3: int(1e6), 4: 0}[config])
    m = ExaModel(core)
    L = m._L
    x, y = bench.eval_point(config, core, m)
    dev = torch.device("cuda:0")
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    new = lambda n: torch.empty(max(1, n), dtype=torch.float64, device=dev)     # noqa: E731
    f, g, c, j, h = new(1), new(m.meta.nvar), new(m.meta.ncon), new(m.meta.nnzj), new(m.meta.nnzh)
    p = lambda t: ctypes.c_void_p(t.data_ptr())                                    # noqa: E731
    sigma = ctypes.c_double(0.5)

    def five():
        rc = [L.exa_obj_async(m.id, p(xd), p(f)), L.exa_grad(m.id, p(xd), p(g)), L.exa_cons(m.id, p(xd), p(c)),
              L.exa_jac(m.id, p(xd), p(j)), L.exa_hess(m.id, p(xd), p(yd), sigma, p(h))]
        assert not any(rc), L.exa_last_error()

    def all_in_one():
        assert L.exa_eval_all(m.id, p(xd), p(yd), sigma, p(f), p(g), p(c), p(j), p(h)) == 0, L.exa_last_error()

    s = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 500 if config != 2 else 100
    print(f"{bench.CONFIGS[config]}: one solver iteration's evaluations, {reps} back-to-back iterations, ms per iteration (min of 5 rounds)")
    with torch.cuda.stream(s):
        L.exa_set_stream(m.id, ctypes.c_void_p(s.cuda_stream))
        for name, fn in (("five callbacks", five), ("exa_eval_all", all_in_one)):
            fn()
            s.synchronize()
            ref = [t.clone() for t in (f, g, c, j, h)]
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                fn()
            for t in (f, g, c, j, h):
                t.fill_(-7.0)
            graph.replay()
            s.synchronize()
            same = all(torch.allclose(a, b, rtol=1e-12, atol=0.0) for a, b in zip(ref, (f, g, c, j, h)))
            res = {}
            for how, run in (("eager", fn), ("graph replay", graph.replay)):
                best = 1e9
                for _ in range(5):
                    for _ in range(20):
                        run()
                    e0.record(s)
                    for _ in range(reps):
                        run()
                    e1.record(s)
                    s.synchronize()
                    best = min(best, e0.elapsed_time(e1) / reps)
                res[how] = best
            print(f"  {name:16s} eager {res['eager']:.4f}   graph replay {res['graph replay']:.4f}   (replay reproduces the eager outputs: {same})")


if __name__ == "__main__":
    main()
