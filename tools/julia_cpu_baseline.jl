# julia_cpu_baseline.jl CONFIG POINTS — BASELINE.md §4 step 1: time the REAL reference (ExaModels.jl, backend = nothing, the
# thread count julia was started with) on the bench workload and print ONE JSON line for bench.py's "cpu_baseline".
# bench.py runs it when `julia` is on the box (it is not on the benchmark pool's image); nothing else uses it.
#   CONFIG 2 / 5: Luksan-Vlcek N = POINTS (benchmark/runbenchmark.jl:163-169);  3 / 4 are not restated here (the COPS and
#   PGLIB model packages are separate Julia packages) and make the script exit non-zero, so bench.py keeps the C port.
using ExaModels, NLPModels, Printf
config, N = parse(Int, ARGS[1]), parse(Int, ARGS[2])
config in (2, 5) || exit(2)
N = min(N, 10_000_000)
lv_x0(i) = mod(i, 2) == 1 ? -1.2 : 1.0
c = ExaCore(concrete = Val(true))
@add_var(c, x, N; start = (lv_x0(i) for i = 1:N))
@add_con(c, s, 3x[i+1]^3 + 2 * x[i+2] - 5 + sin(x[i+1] - x[i+2])sin(x[i+1] + x[i+2]) + 4x[i+1] - x[i]exp(x[i] - x[i+1]) - 3 for i = 1:(N-2))
@add_obj(c, 100 * (x[i-1]^2 - x[i])^2 + (x[i-1] - 1)^2 for i = 2:N)
m = ExaModel(c)
x0 = copy(m.meta.x0) .+ 0.05
y0 = ones(m.meta.ncon)
h = zeros(m.meta.nnzh)
NLPModels.hess_coord!(m, x0, y0, h; obj_weight = 0.5)              # compile
t = minimum(@elapsed(NLPModels.hess_coord!(m, x0, y0, h; obj_weight = 0.5)) for _ = 1:5)
@printf("{\"value\": %.6e, \"unit\": \"nnz/s\", \"cores\": %d, \"evals_per_s\": %.6e, \"sample\": \"ExaModels.jl backend=nothing, LuksanVlcek N=%d hess_coord!, min of 5\"}\n",
        m.meta.nnzh / t, Threads.nthreads(), 1 / t, N)
