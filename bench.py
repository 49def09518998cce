#!/usr/bin/env python
"""bench.py — sparse Lagrangian Hessian evaluation throughput on MI355X (BASELINE.json metric).

A "step" is one hess_coord!(m, x, y, H; obj_weight) of the Luksan-Vlcek model (BASELINE.json configs[1],
benchmark/runbenchmark.jl:163-169) with x, y and H already resident in HBM.  Per-GPU work is fixed
(--points data points of each pattern per GPU, default 1e7): with N GPUs the model is LV(N * points) and every
rank evaluates its contiguous shard of each pattern's iterator, writing its disjoint slice of the global COO
vector — no data-path collective => "scaling": "weak", value = aggregate Hessian nonzeros / s.

    python bench.py                       # 1 GPU, LV N=1e7
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus 8 --steps K --warmup W

One JSON line on rank 0.  Extra objects: "roofline" (algorithmic HBM bytes / measured kernel time against the
8 TB/s peak) and "cpu_baseline" (the C restatement of the reference CPU algorithm, oracle/, timed on this host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "examodels.jl_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(sample_n, threads_all):
    """CPU baseline on this host (rank 0, N=1 only), kind = "port": the real reference is Julia and cannot run here.

    value         hand-specialised straight-line C of the SAME algorithm (zero-fill + one `+=` per contribution in
                  hessian.jl order; oracle/exa_oracle.c `ora_lv_hess_compiled`) on the FULL workload, single thread —
                  the closest available proxy for ExaModels `backend = nothing`, whose patterns Julia compiles.
    all_cores     the same with OpenMP over data points (proxy for the KernelAbstractions CPU() backend).
    interpreter   the generic tree-walking test oracle on a 1e6-point sample (what the parity tests use)."""
    import numpy as np
    import oracle
    from exahip import models
    N = sample_n
    x = models.lv_x0(N) + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)
    y = np.random.default_rng(1).standard_normal(N - 2)
    nnzh = 9 * N - 15
    out = np.empty(nnzh)

    def timeit(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    t1 = timeit(lambda: oracle.lv_hess_compiled(N, x, y, 0.5, out=out, threads=1), 3)
    res = {"value": nnzh / t1, "unit": "nnz/s", "cores": 1, "kind": "port",
           "sample": f"LuksanVlcek N={N} hess_coord! (the full bench workload), 3 evals, hand-specialised C port, 1 thread",
           "evals_per_s": 1.0 / t1}
    if threads_all > 1:
        th = min(threads_all, 64)
        tn = timeit(lambda: oracle.lv_hess_compiled(N, x, y, 0.5, out=out, threads=th), 3)
        res["all_cores"] = {"value": nnzh / tn, "cores": th, "evals_per_s": 1.0 / tn,
                            "note": "OpenMP over data points (proxy for the KernelAbstractions CPU() backend)"}
    n2 = min(N, 1_000_000)
    o = oracle.OracleModel(models.luksan_vlcek_model(n2).to_ir(), threads=1)
    o2 = np.empty(o.nnzh)
    ti = timeit(lambda: o.hess_coord(x[:n2], y[:n2 - 2], 0.5, out=o2), 2)
    res["interpreter"] = {"value": o.nnzh / ti, "cores": 1, "sample": f"LuksanVlcek N={n2}, generic test oracle"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--preheat-ms", type=float, default=200.0,
                    help="untimed: keep the GPU busy with the same call this long before the W warmup steps, so that the "
                         "clock governor has left its idle state (sclk idles at ~570 MHz and takes tens of ms to ramp)")
    ap.add_argument("--points", type=float, default=1e7, help="LV size per GPU (N)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling instead: --points is the GLOBAL LV size, split over the ranks (BASELINE.json configs[4]: "
                         "--points 1e8 --gpus 8); value is then evals/s-proportional nnz/s of the fixed model")
    ap.add_argument("--cpu-sample", type=float, default=1e7)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--all-callbacks", action="store_true", help="also time obj/cons/grad/jac (secondary)")
    ap.add_argument("--with-collectives", action="store_true",
                    help="N>1 only, secondary: also time sharded grad! + RCCL all_reduce (off by default so that nothing "
                         "optional can stall the contract run)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    ndev = torch.cuda.device_count()
    # one rank per GPU.  (EXAHIP_DIST_BACKEND=gloo lets the launch path be exercised on a box with fewer GPUs than
    # ranks — ranks then share GPUs and the number is meaningless; never set by the driver.)
    backend = os.environ.get("EXAHIP_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks need {world} GPUs, found {ndev}")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from exahip import ExaModel, models
    per_gpu = int(args.points) // world if args.strong else int(args.points)
    N = int(args.points) if args.strong else per_gpu * world
    core = models.luksan_vlcek_model(N)
    m = ExaModel(core)
    m.set_shard(rank, world)

    r = np.random.default_rng(0)
    x = m.meta.x0 + 0.1 * r.uniform(-1, 1, m.meta.nvar)
    y = np.random.default_rng(1).standard_normal(m.meta.ncon)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.from_numpy(y).to(dev)
    del x, y
    h = torch.empty(m.meta.nnzh, dtype=torch.float64, device=dev)
    sigma = 0.5

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:
        m.time_callback("hess", 20, xd, yd, sigma, out=h)
    for _ in range(args.warmup):
        m.hess_coord(xd, yd, sigma, out=h)
    barrier()
    t0 = time.perf_counter()
    # exactly K steps; the same K launches are bracketed by hipEvents on the launch stream inside libexahip
    kernel_ms = m.time_callback("hess", args.steps, xd, yd, sigma, out=h)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = t[0].item(), t[1].item()

    nnzh = m.meta.nnzh
    ms_per_step = 1e3 * elapsed / args.steps
    value = nnzh * args.steps / elapsed
    # algorithmic HBM bytes of one launch on one GPU: write each COO slot once, read x and y once;
    # the LV iterators are UnitRanges (no iterator bytes)
    shard_nnzh = nnzh / world
    alg_bytes = 8.0 * shard_nnzh + 8.0 * (m.meta.nvar / world) + 8.0 * (m.meta.ncon / world)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC counters: measured in separate rocprofv3 passes (cannot run inside the timed
    # process), committed under profiles/ and attached only when the workload is the one that was profiled
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r1_traffic_lv1e7.json")
    if world == 1 and per_gpu == 10_000_000 and os.path.exists(tpath):
        with open(tpath) as fh:
            traffic = json.load(fh)["hbm_bytes_per_launch"]
    out = {
        "metric": "sparse Lagrangian Hessian throughput (hess_coord!), nonzeros/s; evals/s in evals_per_s",
        "value": value, "unit": "nnz/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preheat_ms": args.preheat_ms,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"LuksanVlcek N={per_gpu:.0e} per GPU (global N={N:.0e}), hess_coord! sharded-output",
                   "nvar": m.meta.nvar, "ncon": m.meta.ncon, "nnzh": nnzh, "obj_weight": sigma,
                   "parallelism": f"iterator-shard x{world}, no data-path collective"},
        "evals_per_s": args.steps / elapsed,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "exa_hess", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "block_order": {0: "sequential", 1: "interleaved-128"}.get(m._L.exa_block_order(m.id, 4), "n/a")},
    }
    # SURVEY §8d protocol: min and median over >= 30 individually event-bracketed calls (the reference harness reports
    # the BenchmarkTools minimum, benchmark/runbenchmark.jl:94); outside the contract timing above
    per_call = sorted(m.time_callback("hess", 1, xd, yd, sigma, out=h) for _ in range(50))
    out["per_call_ms"] = {"min": per_call[0], "median": per_call[len(per_call) // 2], "n": len(per_call)}
    if world > 1 and args.with_collectives:
        # secondary (never part of `value`): the callbacks that DO need a collective, completed with RCCL all_reduce
        # over xGMI through exahip.dist — sharded grad! + all_reduce(SUM) of the dense nvar vector, and obj.
        try:
            from exahip.dist import ShardedEvaluator
            ev = ShardedEvaluator(m)
            g = torch.empty(m.meta.nvar, dtype=torch.float64, device=dev)
            if backend != "nccl":
                raise RuntimeError("collective timing needs the nccl (RCCL) backend")
            for _ in range(3):
                ev.grad(xd, out=g)
            barrier()
            t1 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                ev.grad(xd, out=g)
            barrier()
            out["collectives"] = {"grad_plus_allreduce_ms": 1e3 * (time.perf_counter() - t1) / reps,
                                  "allreduce_bytes": 8 * m.meta.nvar, "backend": "rccl"}
        except Exception as e:  # keep the contract line alive whatever happens here
            out["collectives"] = {"error": repr(e)}
    if args.all_callbacks:
        g = torch.empty(m.meta.nvar, dtype=torch.float64, device=dev)
        c = torch.empty(m.meta.ncon, dtype=torch.float64, device=dev)
        j = torch.empty(m.meta.nnzj, dtype=torch.float64, device=dev)
        sec = {}
        for name, buf in (("obj", None), ("cons", c), ("grad", g), ("jac", j)):
            m.time_callback(name, 3, xd, out=buf)
            sec[name + "_ms"] = m.time_callback(name, 20, xd, out=buf)
        out["secondary_callbacks"] = sec
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(int(args.cpu_sample), os.cpu_count() or 1)
        out["cpu_baseline"]["gpu_over_cpu_1thread"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
