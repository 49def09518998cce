# reference_check.jl — compares the REAL reference (ExaModels.jl v0.12, backend = nothing) with the golden fixtures this
# repository's oracle and HIP path are tested against (tests/golden/zoo_fixtures/, tests/test_golden_zoo.py).
#
#     julia --project=<env with ExaModels 0.12 and NLPModels> tools/reference_check.jl [fixture dir] [--dump OUTDIR]
#
# --dump OUTDIR additionally writes, for every model, a fixture in the SAME format whose inputs (x, y, sigma, u, v, the ACOPF
# tables) are the committed fixture's and whose outputs — structures and values of all callbacks — are the REFERENCE's:
# commit that directory as tests/golden/zoo_fixtures_reference/ and tests/test_golden_zoo.py checks the oracle and the HIP
# path against what a Julia process computed (structure ==, values 1e-10): "parity unpinned on the Hessian slot order"
# becomes a committed fact.
#
# It cannot run in the build container (no Julia there); it is the one command that, on any machine with Julia, turns
# "the COO slot order is pinned by two independent re-readings of src/hessian.jl" into "a Julia process agrees with every
# slot": for each model it rebuilds the model with the reference's own macros, evaluates the seven NLPModels callbacks and
# the three products at the fixture's (x, y, sigma, u, v) and requires
#     jac_structure! / hess_structure!   ==      (NLPTest.jl:103-112 asks the same of two backends)
#     obj, cons, grad, jac_coord, hess_coord, jprod, jtprod, hprod   within 1e-10 relative (BASELINE.json north_star).
# Models: the Luksan-Vlcek family (test/NLPTest/luksan.jl, benchmark/runbenchmark.jl:163-169, N = 3 ... 1e4 = BASELINE
# config 1), the 2-D split variant with a tuple-target augmentation, the ACOPF of test/NLPTest/power.jl on the synthetic
# 30-bus tables stored in the fixture, the Goddard rocket as this repository states it, trivialmax, and the 2-D
# augmentation model of test/NLPTest/conaug_test.jl:200-213, the two COPS models of benchmark/runbenchmark.jl:239-282 (hanging
# chain, electrons on a sphere) and this repository's two feature models (`mixed`: parameters with an offset index range,
# table iterators with Int and Float columns, a 1-D augmentation, literal and data exponents; `stepped`: StepRange iterators,
# exa_sum / exa_prod, Constant algebra, a parameterised power) — 18 of the 18 fixture models (round 6: + the parametric LV-10 of docs/src/parameters.md).
using ExaModels, NLPModels, Printf
using SpecialFunctions   # loads ExaModelsSpecialFunctions (ext/): the `specialfn` model
import JSON   # any JSON reader will do; JSON.jl is what ExaModels' own test environment has

const DUMP = (k = findfirst(==("--dump"), ARGS); k === nothing ? nothing : ARGS[k+1])
const POS = [a for (i, a) in enumerate(ARGS) if a != "--dump" && (i == 1 || ARGS[i-1] != "--dump")]
const DIR = length(POS) >= 1 ? POS[1] : joinpath(@__DIR__, "..", "tests", "golden", "zoo_fixtures")

function load_fixture(name)
    man = JSON.parsefile(joinpath(DIR, name * ".json"))
    raw = read(joinpath(DIR, name * ".bin"))
    arrays = Dict{String,Any}()
    for (k, (dt, n, off)) in man["arrays"]
        T = dt == "f8" ? Float64 : dt == "i4" ? Int32 : Int64
        arrays[k] = collect(reinterpret(T, raw[off+1:off+n*sizeof(T)]))
    end
    return man["scalars"], arrays
end

# ---- the models, with the reference's macros ---------------------------------------------------------------------------
lv_x0(i) = mod(i, 2) == 1 ? -1.2 : 1.0

function lv_model(N; obj_first = false)            # benchmark/runbenchmark.jl:163-169; docs/src/performance.jl:13-17 puts the objective first
    c = ExaCore(concrete = Val(true))
    @add_var(c, x, N; start = (lv_x0(i) for i = 1:N))
    if obj_first
        @add_obj(c, 100 * (x[i-1]^2 - x[i])^2 + (x[i-1] - 1)^2 for i = 2:N)
        @add_con(c, s, 3x[i+1]^3 + 2 * x[i+2] - 5 + sin(x[i+1] - x[i+2])sin(x[i+1] + x[i+2]) + 4x[i+1] - x[i]exp(x[i] - x[i+1]) - 3 for i = 1:(N-2))
    else
        @add_con(c, s, 3x[i+1]^3 + 2 * x[i+2] - 5 + sin(x[i+1] - x[i+2])sin(x[i+1] + x[i+2]) + 4x[i+1] - x[i]exp(x[i] - x[i+1]) - 3 for i = 1:(N-2))
        @add_obj(c, 100 * (x[i-1]^2 - x[i])^2 + (x[i-1] - 1)^2 for i = 2:N)
    end
    return ExaModel(c; prod = true)
end

function docparam_model(N = 10)                    # docs/src/parameters.md:24-82, as documented: theta = [100, 1], the objective FIRST
    c = ExaCore(concrete = Val(true))
    @add_par(c, θ, [100.0, 1.0])
    @add_var(c, x, N; start = (lv_x0(i) for i = 1:N))
    @add_obj(c, θ[1] * (x[i-1]^2 - x[i])^2 + (x[i-1] - θ[2])^2 for i = 2:N)
    @add_con(c, s, 3x[i+1]^3 + 2 * x[i+2] - 5 + sin(x[i+1] - x[i+2])sin(x[i+1] + x[i+2]) + 4x[i+1] - x[i]exp(x[i] - x[i+1]) - 3 for i = 1:(N-2))
    return ExaModel(c; prod = true)
end

function lv_split_model(N, M)                      # test/NLPTest/luksan.jl:17-26
    c = ExaCore(concrete = Val(true))
    @add_var(c, x, N, M; start = [lv_x0(i) for i = 1:N, j = 1:M])
    @add_con(c, s, 3x[i+1, j]^3 + 2 * x[i+2, j] - 5 for i = 1:(N-2), j = 1:M)
    @add_con!(c, s, (i, j) => sin(x[i+1, j] - x[i+2, j])sin(x[i+1, j] + x[i+2, j]) + 4x[i+1, j] - x[i, j]exp(x[i, j] - x[i+1, j]) - 3 for i = 1:(N-2), j = 1:M)
    @add_obj(c, 100 * (x[i-1, j]^2 - x[i, j])^2 + (x[i-1, j] - 1)^2 for i = 2:N, j = 1:M)
    return ExaModel(c; prod = true)
end

struct I2; i::Tuple{Int, Int}; end                 # test/NLPTest/luksan_struct.jl:1-20: the split model over a Matrix of NESTED structs,
struct I1; i::I2; end                              # fields reached through the access path i.i.i[1] / i.i.i[2]
function lv_struct_model(N, M)
    c = ExaCore(concrete = Val(true))
    data = [I1(I2((i, j))) for i = 1:N, j = 1:M]
    @add_var(c, x, N, M; start = [lv_x0(i.i.i[1]) for i in data])
    @add_con(c, s, 3x[i.i.i[1]+1, i.i.i[2]]^3 + 2 * x[i.i.i[1]+2, i.i.i[2]] - 5 for i in data[1:end-2, :])
    @add_con!(c, s, (i.i.i[1], i.i.i[2]) => sin(x[i.i.i[1]+1, i.i.i[2]] - x[i.i.i[1]+2, i.i.i[2]])sin(x[i.i.i[1]+1, i.i.i[2]] + x[i.i.i[1]+2, i.i.i[2]]) + 4x[i.i.i[1]+1, i.i.i[2]] - x[i.i.i[1], i.i.i[2]]exp(x[i.i.i[1], i.i.i[2]] - x[i.i.i[1]+1, i.i.i[2]]) - 3 for i in data[1:end-2, :])
    @add_obj(c, 100 * (x[i-1, j]^2 - x[i, j])^2 + (x[i-1, j] - 1)^2 for i = 2:N, j = 1:M)
    return ExaModel(c; prod = true)
end

function trivialmax_model(n)                       # test/NLPTest/trivialmax.jl:3-10 (start 0.3 as in tests/zoo.py)
    c = ExaCore(; minimize = false, concrete = Val(true))
    @add_var(c, x, n; start = fill(0.3, n))
    @add_con(c, s, x[1]; lcon = 0, ucon = 1)
    @add_obj(c, x[1]^2)
    return ExaModel(c; prod = true)
end

function conaug2d_model()                          # tests/zoo.py conaug2d_model; API as in test/NLPTest/conaug_test.jl:185-191
    N, M = 4, 5
    c = ExaCore(concrete = Val(true))
    c, x = add_var(c, N, M; start = reshape(collect(1.0:(N*M)), N, M))
    c, g = add_con(c, N, M; lcon = -Inf, ucon = Inf)
    fwd = [(i, j) for j = 1:M for i = 1:(N-1)]
    bwd = [(i, j) for j = 1:M for i = 2:N]
    c, _ = add_con!(c, g[i, j] += x[i, j] * x[i+1, j] for (i, j) in fwd)
    c, _ = add_con!(c, g[i, j] += x[i-1, j] - x[i, j]^2 for (i, j) in bwd)
    c, _ = add_obj(c, x[i, j]^2 for i = 1:N, j = 1:M)
    return ExaModel(c; prod = true)
end

function rocket_model(nh)                          # examodels.jl_amd/exahip/models.py rocket_model; velocity row = README.md:20-26
    h_0, v_0, m_0, g_0 = 1.0, 0.0, 1.0, 1.0
    T_c, h_c, v_c, m_c = 3.5, 500.0, 620.0, 0.6
    c_ = 0.5 * sqrt(g_0 * h_0); m_f = m_c * m_0; D_c = 0.5 * v_c * (m_0 / g_0); T_max = T_c * m_0 * g_0
    core = ExaCore(; minimize = false, concrete = Val(true))
    @add_var(core, h, 0:nh; start = ones(nh + 1), lvar = fill(h_0, nh + 1))
    @add_var(core, v, 0:nh; start = [(k / nh) * (1.0 - k / nh) for k = 0:nh], lvar = zeros(nh + 1))
    @add_var(core, m, 0:nh; start = [(m_f - m_0) * (k / nh) + m_0 for k = 0:nh], lvar = fill(m_f, nh + 1), uvar = fill(m_0, nh + 1))
    @add_var(core, tau, 0:nh; start = fill(T_max / 2.0, nh + 1), lvar = zeros(nh + 1), uvar = fill(T_max, nh + 1))
    @add_var(core, dt, 1; start = [1.0 / nh], lvar = [0.0])
    @add_obj(core, h[i] for i = nh:nh)
    @add_con(core, c1, -h[i] + h[i-1] + 0.5 * dt[1] * (v[i] + v[i-1]) for i = 1:nh)
    @add_con(core, c2, -v[i] + v[i-1] + 0.5 * dt[1] * (
        (tau[i] - D_c * v[i]^2 * exp(-h_c * (h[i] - h_0) / h_0) - m[i] * g_0 * (h_0 / h[i])^2) / m[i] +
        (tau[i-1] - D_c * v[i-1]^2 * exp(-h_c * (h[i-1] - h_0) / h_0) - m[i-1] * g_0 * (h_0 / h[i-1])^2) / m[i-1]) for i = 1:nh)
    @add_con(core, c3, -m[i] + m[i-1] - 0.5 * dt[1] * (tau[i] + tau[i-1]) / c_ for i = 1:nh)
    @add_con(core, b1, h[i] - h_0 for i = 0:0)
    @add_con(core, b2, v[i] - v_0 for i = 0:0)
    @add_con(core, b3, m[i] - m_0 for i = 0:0)
    @add_con(core, b4, m[i] - m_f for i = nh:nh)
    return ExaModel(core; prod = true)
end

function acopf_model(a)                            # test/NLPTest/power.jl:112-213 on the tables stored in the fixture
    nt(tab, cols) = [NamedTuple{Tuple(Symbol.(cols))}(Tuple(a["data_$(tab)_$(c)"][k] for c in cols)) for k = 1:length(a["data_$(tab)_$(cols[1])"])]
    bus = nt("bus", ["i", "pd", "gs", "qd", "bs"])
    gen = nt("gen", ["i", "cost1", "cost2", "cost3", "bus"])
    arc = nt("arc", ["i", "rate_a", "bus"])
    branch = nt("branch", ["i", "j", "f_idx", "t_idx", "f_bus", "t_bus", "c1", "c2", "c3", "c4", "c5", "c6", "c7", "c8", "rate_a_sq"])
    w = ExaCore(concrete = Val(true))
    @add_var(w, va, length(bus);)
    @add_var(w, vm, length(bus); start = ones(length(bus)), lvar = a["data_vmin"], uvar = a["data_vmax"])
    @add_var(w, pg, length(gen); lvar = a["data_pmin"], uvar = a["data_pmax"])
    @add_var(w, qg, length(gen); lvar = a["data_qmin"], uvar = a["data_qmax"])
    @add_var(w, p, length(arc); lvar = -a["data_rate_a"], uvar = a["data_rate_a"])
    @add_var(w, q, length(arc); lvar = -a["data_rate_a"], uvar = a["data_rate_a"])
    @add_obj(w, g.cost1 * pg[g.i]^2 + g.cost2 * pg[g.i] + g.cost3 for g in gen)
    @add_con(w, c1, va[i] for i in Int.(a["data_ref_buses"]))
    @add_con(w, c2, p[b.f_idx] - b.c5 * vm[b.f_bus]^2 - b.c3 * (vm[b.f_bus] * vm[b.t_bus] * cos(va[b.f_bus] - va[b.t_bus])) -
                    b.c4 * (vm[b.f_bus] * vm[b.t_bus] * sin(va[b.f_bus] - va[b.t_bus])) for b in branch)
    @add_con(w, c3, q[b.f_idx] + b.c6 * vm[b.f_bus]^2 + b.c4 * (vm[b.f_bus] * vm[b.t_bus] * cos(va[b.f_bus] - va[b.t_bus])) -
                    b.c3 * (vm[b.f_bus] * vm[b.t_bus] * sin(va[b.f_bus] - va[b.t_bus])) for b in branch)
    @add_con(w, c4, p[b.t_idx] - b.c7 * vm[b.t_bus]^2 - b.c1 * (vm[b.t_bus] * vm[b.f_bus] * cos(va[b.t_bus] - va[b.f_bus])) -
                    b.c2 * (vm[b.t_bus] * vm[b.f_bus] * sin(va[b.t_bus] - va[b.f_bus])) for b in branch)
    @add_con(w, c5, q[b.t_idx] + b.c8 * vm[b.t_bus]^2 + b.c2 * (vm[b.t_bus] * vm[b.f_bus] * cos(va[b.t_bus] - va[b.f_bus])) -
                    b.c1 * (vm[b.t_bus] * vm[b.f_bus] * sin(va[b.t_bus] - va[b.f_bus])) for b in branch)
    @add_con(w, c6, va[b.f_bus] - va[b.t_bus] for b in branch; lcon = a["data_angmin"], ucon = a["data_angmax"])
    @add_con(w, c7, p[b.f_idx]^2 + q[b.f_idx]^2 - b.rate_a_sq for b in branch; lcon = fill(-Inf, length(branch)))
    @add_con(w, c8, p[b.t_idx]^2 + q[b.t_idx]^2 - b.rate_a_sq for b in branch; lcon = fill(-Inf, length(branch)))
    @add_con(w, c9, b.pd + b.gs * vm[b.i]^2 for b in bus)
    @add_con(w, c10, b.qd - b.bs * vm[b.i]^2 for b in bus)
    @add_con!(w, c9, x.bus => p[x.i] for x in arc)
    @add_con!(w, c10, x.bus => q[x.i] for x in arc)
    @add_con!(w, c9, g.bus => -pg[g.i] for g in gen)
    @add_con!(w, c10, g.bus => -qg[g.i] for g in gen)
    return ExaModel(w; prod = true)
end

function cops_chain_model(n)                       # benchmark/runbenchmark.jl:239-264, operand for operand
    nh = max(2, div(n - 4, 4))
    L = 4; a = 1; b = 3
    tmin = b > a ? 1/4 : 3/4
    tf = 1.0; h = tf / nh
    c = ExaCore(concrete = Val(true))
    @add_var(c, u,  nh+1; start = [4*abs(b-a)*(k/nh - tmin) for k in 1:nh+1])
    @add_var(c, x1, nh+1; start = [4*abs(b-a)*k/nh*(1/2*k/nh - tmin) + a for k in 1:nh+1])
    @add_var(c, x2, nh+1; start = [(4*abs(b-a)*k/nh*(1/2*k/nh - tmin) + a) * (4*abs(b-a)*(k/nh - tmin)) for k in 1:nh+1])
    @add_var(c, x3, nh+1; start = [4*abs(b-a)*(k/nh - tmin) for k in 1:nh+1])
    @add_obj(c, x2[nh+1])
    @add_con(c, c1, x1[j+1] - x1[j] - 1/2*h*(u[j] + u[j+1]) for j in 1:nh)
    @add_con(c, c2, x1[1] - a)
    @add_con(c, c3, x1[nh+1] - b)
    @add_con(c, c4, x2[1])
    @add_con(c, c5, x3[1])
    @add_con(c, c6, x3[nh+1] - L)
    @add_con(c, c7, x2[j+1] - x2[j] - 1/2*h*(x1[j]*sqrt(1+u[j]^2) + x1[j+1]*sqrt(1+u[j+1]^2)) for j in 1:nh)
    @add_con(c, c8, x3[j+1] - x3[j] - 1/2*h*(sqrt(1+u[j]^2) + sqrt(1+u[j+1]^2)) for j in 1:nh)
    return ExaModel(c; prod = true)
end

function cops_elec_model(np)                       # benchmark/runbenchmark.jl:267-282 (the start point is not compared: x comes from the fixture)
    itr = [(i, j) for i in 1:np-1 for j in i+1:np]
    core = ExaCore(concrete = Val(true))
    @add_var(core, x, 1:np; start = zeros(np))
    @add_var(core, y, 1:np; start = zeros(np))
    @add_var(core, z, 1:np; start = ones(np))
    @add_obj(core, 1 / sqrt((x[i]-x[j])^2 + (y[i]-y[j])^2 + (z[i]-z[j])^2) for (i, j) in itr)
    @add_con(core, c1, x[i]^2 + y[i]^2 + z[i]^2 - 1 for i = 1:np)
    return ExaModel(core; prod = true)
end

function mixed_model()                             # tests/zoo.py mixed_model, statement for statement
    c = ExaCore(concrete = Val(true))
    @add_var(c, x, 12; start = collect(range(0.5, 1.5; length = 12)))
    @add_var(c, z, 0:5; start = fill(0.7, 6))
    c, th = add_par(c, 2:4; value = [10.0, 20.0, 30.0])
    tab = [(i = i, j = j, w = w, e = e) for (i, j, w, e) in zip([1, 3, 5, 7], [2, 4, 6, 8], [0.5, 1.5, -2.0, 3.0], [2, 3, 4, 5])]
    @add_obj(c, d.w * (x[d.i] - x[d.j])^2 + sin(x[d.i] * x[d.j]) for d in tab)
    @add_obj(c, exp(-z[i]) * z[i]^3 + th[2] * z[i] for i = 0:5)
    @add_con(c, g1, th[j] * x[1] * x[j] + log(x[j+1]) for j = 2:4)
    @add_con(c, g, x[d.i] / x[d.j] - d.w * sqrt(x[d.j]) + x[d.i]^d.e for d in tab; lcon = fill(-1.0, 4), ucon = fill(1.0, 4))
    @add_con!(c, g, k => tanh(x[k] * x[k+4]) - 3 * x[k+8] for k = 1:4)
    @add_con(c, g3, z[i] * z[i-1] - cos(z[i] - x[12]) + 2.0^z[i] for i = 1:5)
    return ExaModel(c; prod = true)
end

function specialfn_model(n = 40)                    # tests/zoo.py specialfn_model: every entry of ext/functionlist.jl (needs `using SpecialFunctions`)
    c = ExaCore(concrete = Val(true))
    @add_var(c, x, n; start = collect(range(0.45, 1.35; length = n)))
    c, th = add_par(c, 2; value = [1.25, 0.5])
    is = collect(1:3:(n-2)); js = collect(3:3:n)[1:length(is)]; ws = collect(range(0.5, 2.0; length = length(is)))
    tab = [(i = i, j = j, w = w) for (i, j, w) in zip(is, js, ws)]
    @add_obj(c, erf(x[i] - x[i+1]) * gamma(x[i] + 1.0) + beta(x[i] + 0.5, x[i+1] + th[1]) + airyai(2.0 * x[i] - 3.0 * x[i+1])
                + besselj0(4.0 * x[i]) + dawson(x[i] * x[i+1]) + erfcx(x[i] - 2.0) + digamma(x[i] + 0.2) for i = 1:(n-1))
    @add_con(c, g, erfinv(x[i] * 0.5) + invdigamma(x[i] - x[i+1]) + logbeta(x[i] + 0.1, 2.0) + airybiprime(x[i+1] - 2.0)
                + bessely1(x[i] + 0.5) + erfi(x[i]) + trigamma(x[i] + 0.3) + erfcinv(x[i]) + erfc(x[i] * x[i+2]) for i = 1:(n-2))
    @add_con(c, g2, d.w * besselj1(x[d.i] * 3.0) * bessely0(x[d.j] + 1.0) + airybi(-x[d.i] * x[d.j]) + airyaiprime(x[d.j])
                + beta(th[2] + 1.0, x[d.i]) * logbeta(x[d.i], x[d.j]) for d in tab; lcon = fill(-5.0, length(tab)), ucon = fill(5.0, length(tab)))
    @add_con!(c, g, k => erf(x[k] * x[k+5]) + gamma(x[k+2]) for k = 1:6)
    return ExaModel(c; prod = true)
end

function stepped_model()                           # tests/zoo.py stepped_model: StepRange iterators, exa_sum / exa_prod, Constant
    N = 50
    c = ExaCore(concrete = Val(true))
    @add_var(c, x, N; start = collect(range(0.4, 1.6; length = N)))
    c, th = add_par(c, 2; value = [1.5, 0.25])
    @add_obj(c, (x[i] - x[i+1])^2 + Constant(1) * x[i] * Constant(0) + th[1] * x[i+1] for i = 1:2:(N-1))
    @add_obj(c, exa_sum(x[i+k]^2 for k in 0:2) * 0.5 for i = 2:3:(N-2))
    @add_con(c, s1, exa_prod(1 + x[i+k] for k in 0:2) - x[i]^th[2] + Constant(2)^x[i+1] for i = 3:3:(N-2))
    @add_con(c, s2, (x[i] + (-x[i-1]) + 2 * x[i-2]) / (1 + x[i]^2) for i = N:-4:3)
    return ExaModel(c; prod = true)
end

const MODELS = [
    ("lv10_docparam", a -> docparam_model(10)), ("lv3", a -> lv_model(3)), ("lv20", a -> lv_model(20)), ("lv20_objfirst", a -> lv_model(20; obj_first = true)),
    ("lv1000", a -> lv_model(1000)), ("lv10000", a -> lv_model(10_000)),
    ("lv_split_20x1", a -> lv_split_model(20, 1)), ("lv_split_20x2", a -> lv_split_model(20, 2)), ("lv_struct_20x2", a -> lv_struct_model(20, 2)),
    ("trivialmax", a -> trivialmax_model(6)), ("conaug2d", a -> conaug2d_model()),
    ("rocket50", a -> rocket_model(50)), ("acopf30", a -> acopf_model(a)),
    ("cops_chain", a -> cops_chain_model(200)), ("cops_elec", a -> cops_elec_model(25)),
    ("mixed", a -> mixed_model()), ("stepped", a -> stepped_model()), ("specialfn", a -> specialfn_model()),
]
# `stepped`: tests/zoo.py builds exa_sum / exa_prod as the reference's SumNode / ProdNode over the literal offsets 0:2 and the last
# constraint's sum from a list — if the first run shows a structure mismatch on this model only, compare that row's
# statement with tests/zoo.py:stepped_model before suspecting the evaluator.

relerr(a, ref) = isempty(ref) ? 0.0 : maximum(abs.(a .- ref) ./ max.(abs.(ref), 1e-3 * max(1.0, maximum(abs.(ref)))))

function check(name, build)
    sc, a = load_fixture(name)
    m = build(a)
    ok = true
    say(what, good) = (good || (ok = false); @printf("  %-18s %s\n", what, good ? "ok" : "MISMATCH"))
    say("sizes", (m.meta.nvar, m.meta.ncon, m.meta.nnzj, m.meta.nnzh) == (sc["nvar"], sc["ncon"], sc["nnzj"], sc["nnzh"]))
    x, y, u, v, s = a["x"], a["y"], a["u"], a["v"], sc["sigma"]
    jr, jc = zeros(Int, m.meta.nnzj), zeros(Int, m.meta.nnzj)
    hr, hc = zeros(Int, m.meta.nnzh), zeros(Int, m.meta.nnzh)
    NLPModels.jac_structure!(m, jr, jc); NLPModels.hess_structure!(m, hr, hc)
    say("jac_structure ==", jr == a["jac_rows"] && jc == a["jac_cols"])
    say("hess_structure ==", hr == a["hess_rows"] && hc == a["hess_cols"])
    say("obj", abs(NLPModels.obj(m, x) - sc["obj"]) <= 1e-10 * max(1.0, abs(sc["obj"])))
    say("cons", relerr(NLPModels.cons(m, x), a["cons"]) <= 1e-10)
    say("grad", relerr(NLPModels.grad(m, x), a["grad"]) <= 1e-10)
    jv = zeros(m.meta.nnzj); NLPModels.jac_coord!(m, x, jv)
    hv = zeros(m.meta.nnzh); NLPModels.hess_coord!(m, x, y, hv; obj_weight = s)
    say("jac_coord", relerr(jv, a["jac_vals"]) <= 1e-10)
    say("hess_coord", relerr(hv, a["hess_vals"]) <= 1e-10)
    say("jprod", relerr(NLPModels.jprod(m, x, u), a["jprod"]) <= 1e-10)
    say("jtprod", relerr(NLPModels.jtprod(m, x, v), a["jtprod"]) <= 1e-10)
    say("hprod", relerr(NLPModels.hprod(m, x, y, u; obj_weight = s), a["hprod"]) <= 1e-10)
    if DUMP !== nothing
        out = Pair{String,Any}["x" => x, "y" => y, "u" => u, "v" => v,
            "cons" => NLPModels.cons(m, x), "grad" => NLPModels.grad(m, x),
            "jac_rows" => Int32.(jr), "jac_cols" => Int32.(jc), "jac_vals" => jv,
            "hess_rows" => Int32.(hr), "hess_cols" => Int32.(hc), "hess_vals" => hv,
            "jprod" => NLPModels.jprod(m, x, u), "jtprod" => NLPModels.jtprod(m, x, v), "hprod" => NLPModels.hprod(m, x, y, u; obj_weight = s)]
        for (k, val) in a                      # the model's data tables travel along unchanged
            startswith(k, "data_") && push!(out, k => val)
        end
        scal = Dict("nvar" => m.meta.nvar, "ncon" => m.meta.ncon, "nnzj" => m.meta.nnzj, "nnzh" => m.meta.nnzh, "sigma" => s,
                    "obj" => NLPModels.obj(m, x), "minimize" => m.meta.minimize, "generated_by" => "tools/reference_check.jl --dump (ExaModels.jl, backend = nothing)")
        dump_fixture(DUMP, name, scal, out)
    end
    return ok
end

# the format of tests/golden/make_zoo_fixtures.py: <name>.bin = the arrays back to back (little endian), <name>.json = scalars +
# {array name: [dtype, count, byte offset]}
function dump_fixture(dir, name, scalars, arrays)
    mkpath(dir)
    man = Dict{String,Any}("scalars" => scalars, "arrays" => Dict{String,Any}())
    off = 0
    open(joinpath(dir, name * ".bin"), "w") do io
        for (k, val) in arrays
            v = val isa AbstractVector ? collect(val) : [val]
            dt = eltype(v) == Float64 ? "f8" : eltype(v) == Int32 ? "i4" : "i8"
            v = dt == "i8" ? Int64.(v) : v
            write(io, htol.(v))
            man["arrays"][k] = Any[dt, length(v), off]
            off += sizeof(v)
        end
    end
    open(joinpath(dir, name * ".json"), "w") do io
        JSON.print(io, man, 1)
    end
end

function main()
    bad = String[]
    for (name, build) in MODELS
        println(name)
        check(name, build) || push!(bad, name)
    end
    println(isempty(bad) ? "ALL MODELS AGREE WITH THE FIXTURES" : "MISMATCH in: " * join(bad, ", "))
    exit(isempty(bad) ? 0 : 1)
end

main()
