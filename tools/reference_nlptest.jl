# reference_nlptest.jl — the reference's own backend-equivalence test (test/NLPTest/NLPTest.jl:48-114, `full = true`)
# with libexahip as the backend under test:  m1 = ExaModel built with backend = nothing (the reference's CPU loops),
# m2 = the same model built with backend = ExaModelsHIP.HIPNativeBackend() (every callback forwarded to the C ABI of
# libexahip.so through examodels.jl_amd/julia/ExaModelsHIP.jl).  Structure arrays must be ==, values ≈.
#
#     EXAHIP_LIB=/path/to/libexahip.so julia --project=<env: ExaModels 0.12, NLPModels, AMDGPU> tools/reference_nlptest.jl
#
# Needs Julia + an MI355X: it cannot run in the build container, and it is the first thing to run where both exist.
using Test, ExaModels, NLPModels, AMDGPU
include(joinpath(@__DIR__, "..", "examodels.jl_amd", "julia", "ExaModelsHIP.jl"))
using .ExaModelsHIP: HIPNativeBackend

lv_x0(i) = mod(i, 2) == 1 ? -1.2 : 1.0
function lv(backend, N; M = 1)                     # test/NLPTest/luksan.jl:17-26
    c = ExaCore(backend = backend, concrete = Val(true))
    @add_var(c, x, N, M; start = [lv_x0(i) for i = 1:N, j = 1:M])
    @add_con(c, s, 3x[i+1, j]^3 + 2 * x[i+2, j] - 5 for i = 1:(N-2), j = 1:M)
    @add_con!(c, s, (i, j) => sin(x[i+1, j] - x[i+2, j])sin(x[i+1, j] + x[i+2, j]) + 4x[i+1, j] - x[i, j]exp(x[i, j] - x[i+1, j]) - 3 for i = 1:(N-2), j = 1:M)
    @add_obj(c, 100 * (x[i-1, j]^2 - x[i, j])^2 + (x[i-1, j] - 1)^2 for i = 2:N, j = 1:M)
    return ExaModel(c; prod = true)
end
function rosenrock(backend, N)                     # benchmark/runbenchmark.jl:163-169
    c = ExaCore(concrete = Val(true); backend)
    @add_var(c, x, N; start = (lv_x0(i) for i = 1:N))
    @add_con(c, s, 3x[i+1]^3 + 2 * x[i+2] - 5 + sin(x[i+1] - x[i+2])sin(x[i+1] + x[i+2]) + 4x[i+1] - x[i]exp(x[i] - x[i+1]) - 3 for i = 1:(N-2))
    @add_obj(c, 100 * (x[i-1]^2 - x[i])^2 + (x[i-1] - 1)^2 for i = 2:N)
    return ExaModel(c)
end

host(v) = Array(v)
function test_nlp(m1, m2; tol = 1e-10)             # NLPTest.jl:48-114 with the device arrays brought back for comparison
    @testset "meta" begin
        for f in (:nvar, :ncon, :nnzj, :nnzh)
            @test getfield(m1.meta, f) == getfield(m2.meta, f)
        end
        for f in (:x0, :lvar, :uvar, :lcon, :ucon)
            @test host(getfield(m1.meta, f)) ≈ host(getfield(m2.meta, f)) atol = tol rtol = tol
        end
    end
    @testset "callbacks" begin
        x0 = host(copy(m1.meta.x0)) .+ 0.01
        y0 = randn(m1.meta.ncon); u = randn(m1.meta.nvar); v = randn(m1.meta.ncon)
        d(a) = ROCArray(a)
        @test NLPModels.obj(m1, x0) ≈ NLPModels.obj(m2, d(x0)) atol = tol rtol = tol
        @test NLPModels.cons(m1, x0) ≈ host(NLPModels.cons(m2, d(x0))) atol = tol rtol = tol
        @test NLPModels.grad(m1, x0) ≈ host(NLPModels.grad(m2, d(x0))) atol = tol rtol = tol
        @test NLPModels.jprod(m1, x0, u) ≈ host(NLPModels.jprod(m2, d(x0), d(u))) atol = tol rtol = tol
        @test NLPModels.jtprod(m1, x0, v) ≈ host(NLPModels.jtprod(m2, d(x0), d(v))) atol = tol rtol = tol
        @test NLPModels.hprod(m1, x0, y0, u) ≈ host(NLPModels.hprod(m2, d(x0), d(y0), d(u))) atol = tol rtol = tol
        j1, j2 = zeros(m1.meta.nnzj), ROCArray(zeros(m2.meta.nnzj))
        h1, h2 = zeros(m1.meta.nnzh), ROCArray(zeros(m2.meta.nnzh))
        NLPModels.jac_coord!(m1, x0, j1); NLPModels.jac_coord!(m2, d(x0), j2)
        NLPModels.hess_coord!(m1, x0, y0, h1); NLPModels.hess_coord!(m2, d(x0), d(y0), h2)
        @test j1 ≈ host(j2) atol = tol rtol = tol
        @test h1 ≈ host(h2) atol = tol rtol = tol
        # the objective-only forms (nlp.jl:1906-1915, :1942-1952): exa_hess / exa_hprod with y == NULL
        NLPModels.hess_coord!(m1, x0, h1); NLPModels.hess_coord!(m2, d(x0), h2)
        @test h1 ≈ host(h2) atol = tol rtol = tol
        @test NLPModels.hprod(m1, x0, u) ≈ host(NLPModels.hprod(m2, d(x0), d(u))) atol = tol rtol = tol
        for (st!, n) in ((NLPModels.jac_structure!, m1.meta.nnzj), (NLPModels.hess_structure!, m1.meta.nnzh))
            r1, c1, r2, c2 = zeros(Int, n), zeros(Int, n), zeros(Int, n), zeros(Int, n)
            st!(m1, r1, c1); st!(m2, r2, c2)
            @test r1 == r2
            @test c1 == c2
        end
    end
end

@testset "libexahip behind ExaModels' NLPModels surface" begin
    for (name, build) in (("luksan N=3", b -> lv(b, 3)), ("luksan N=20", b -> lv(b, 20)), ("luksan N=20 M=2", b -> lv(b, 20; M = 2)),
                          ("rosenrock N=1e4", b -> rosenrock(b, 10_000)), ("rosenrock N=1e6", b -> rosenrock(b, 1_000_000)))
        @testset "$name" begin
            test_nlp(build(nothing), build(HIPNativeBackend()))
        end
    end
end
