# ExaModelsHIP.jl — reference-side binding of libexahip.so (the maintainer-facing shim of INTEGRATION.md).
#
# STATUS: written against ExaModels v0.12.0 sources, NOT executed in the build container (no julia there).
# It is deliberately thin: (1) lower a built ExaCore's blocks to the pattern table of include/exahip_ir.h,
# (2) hand it to exa_new_from_table, (3) forward the seven NLPModels callbacks to the C ABI with device pointers.
#
# Plugs into the same seam the KernelAbstractions extension uses:
#   ExaModels.build_extension(c::ExaCore; prod)            src/nlp.jl:898, KA ext :33-37
#   callbacks on AbstractExaModel{T,VT,E<:HIPExtension}     KA ext :253-256, :515-520
module ExaModelsHIP

import ExaModels
import ExaModels: ExaCore, AbstractExaModel, Var, ParameterNode, DataSource, DataIndexed, Node1, Node2, Constant,
    Null, SumNode, ProdNode, Objective, Constraint, ConstraintAugmentation
import NLPModels
import AMDGPU
using AMDGPU: ROCArray, ROCBackend

const LIB = get(ENV, "EXAHIP_LIB", "libexahip.so")

struct HIPNativeBackend end                     # ExaCore(; backend = HIPNativeBackend())
ExaModels.convert_array(v, ::HIPNativeBackend) = ROCArray(v)

mutable struct HIPExtension
    id::Cint
    keep::Vector{Any}                           # host buffers referenced by the table during the build call
end

# ---- wire structs (must match include/exahip_ir.h) ---------------------------------------------------------
struct CNode;    op::Int32; fn::Int32; a::Int32; b::Int32; fval::Float64; ival::Int64; end
struct CColumn;  type::Int32; pad::Int32; data::Ptr{Cvoid}; start::Int64; step::Int64; end
struct CPattern; kind::Int32; n_nodes::Int32; nodes::Ptr{CNode}; root::Int32; target::Int32; base::Int32;
                 n_cols::Int32; cols::Ptr{CColumn}; n::Int64; end
struct CModelDesc; nvar::Int64; npar::Int64; x0::Ptr{Float64}; lvar::Ptr{Float64}; uvar::Ptr{Float64};
                 theta0::Ptr{Float64}; n_patterns::Int32; minimize::Int32; patterns::Ptr{CPattern};
                 y0::Ptr{Float64}; lcon::Ptr{Float64}; ucon::Ptr{Float64}; end

const OP_CONST_F, OP_CONST_I, OP_DATA, OP_PAR, OP_VAR, OP_UN, OP_BIN, OP_NULLV = Int32.(0:7)
const UN = Dict(f => Int32(i - 1) for (i, f) in enumerate((+, -, inv, sqrt, cbrt, abs, abs2, sign, exp, exp2, exp10,
    expm1, log, log2, log1p, log10, sin, cos, tan, asin, acos, atan, acot, csc, sec, cot, sinh, cosh, tanh, asinh,
    acosh, csch, sech, coth, sind, cosd, tand, cscd, secd, cotd, atand, acotd, sinpi, cospi, sinc, deg2rad, rad2deg,
    signbit, floor, ceil, atanh, acoth)))
const BIN = Dict(f => Int32(i - 1) for (i, f) in enumerate((+, -, *, /, ^, atan, hypot, max, min)))
# the SpecialFunctions extension (ext/functionlist.jl; include/exahip_ir.h EXA_U_ERF.., EXA_B_BETA..): SpecialFunctions.jl is a weak
# dependency of ExaModels, so these are matched by NAME (nameof of the node's function) and need no import here
const UN_SPECIAL = (:erf, :erfc, :erfi, :erfcx, :digamma, :trigamma, :invdigamma, :gamma, :airyai, :airybi, :airyaiprime,
    :airybiprime, :besselj0, :bessely0, :besselj1, :bessely1, :dawson, :erfinv, :erfcinv)
const BIN_SPECIAL = (:beta, :logbeta)

# ---- lowering of one expression tree (graph.jl:37-300) ------------------------------------------------------
mutable struct Lower
    nodes::Vector{CNode}
    paths::Vector{Any}                          # distinct DataIndexed access paths -> column ids
end
push_node!(l, op, fn, a, b, f, i) = (push!(l.nodes, CNode(op, fn, a, b, f, i)); Int32(length(l.nodes) - 1))
path(::DataSource) = ()
path(d::DataIndexed{I,J}) where {I,J} = (path(getfield(d, :inner))..., J)
function col!(l, p)
    k = findfirst(==(p), l.paths)
    k === nothing && (push!(l.paths, p); k = length(l.paths))
    return Int32(k - 1)
end
lower!(l, v::Integer) = push_node!(l, OP_CONST_I, 0, -1, -1, 0.0, Int64(v))
lower!(l, v::Real) = push_node!(l, OP_CONST_F, 0, -1, -1, Float64(v), 0)
lower!(l, ::Constant{v}) where {v} = lower!(l, v)
lower!(l, ::Val{v}) where {v} = lower!(l, v)
lower!(l, n::Null) = push_node!(l, OP_NULLV, 0, -1, -1, n.value === nothing ? 0.0 : Float64(n.value), 0)
lower!(l, d::Union{DataSource,DataIndexed}) = push_node!(l, OP_DATA, 0, col!(l, path(d)), -1, 0.0, 0)
lower!(l, v::Var) = push_node!(l, OP_VAR, 0, lower!(l, v.i), -1, 0.0, 0)
lower!(l, v::ParameterNode) = push_node!(l, OP_PAR, 0, lower!(l, v.i), -1, 0.0, 0)
# functions registered by the user (@register_univariate / @register_bivariate outside src/functionlist.jl) have no
# entry in the library's derivative table: say so instead of failing with a bare KeyError
# User functions: in the reference `@register_univariate(f, df, ddf)` / `@register_bivariate(f, d1, d2, d11, d12, d22)` (src/register.jl) take
# Julia lambdas; libexahip compiles the rules into its kernels, so here they are given as HIP device expressions (include/exahip.h:
# exa_register_univariate / exa_register_bivariate — placeholders $1, $2, $3).  Call these once per process, after the reference's macro:
#     @register_univariate(softplus, x -> 1 / (1 + exp(-x)), x -> ...)            # the reference's registration (CPU path, tracing)
#     ExaModelsHIP.register_univariate(softplus, "log1p(exp(\$1))", "1.0 / (1.0 + exp(-\$1))", "\$3 * (1.0 - \$3)")
const USER_UN = Dict{Any,Int32}()
const USER_BIN = Dict{Any,Int32}()
function register_univariate(f, fx::String, dfx::String, ddfx::String; helpers::String = "")
    id = ccall((:exa_register_univariate, LIB), Cint, (Cstring, Cstring, Cstring, Cstring, Cstring), string(nameof(f)), fx, dfx, ddfx, helpers)
    id < 0 && error(unsafe_string(ccall((:exa_last_error, LIB), Cstring, ())))
    USER_UN[f] = Int32(id)
end
# one statement for value and both derivatives (exa_register_univariate_fused: $1 argument, $2 $3 $4 receive f, f', f''), for functions
# whose derivatives share work with the value:  register_univariate_fused(mysin, "exa_sincos(\$1, &\$2, &\$3); \$4 = -\$2;")
function register_univariate_fused(f, stmt::String; helpers::String = "")
    id = ccall((:exa_register_univariate_fused, LIB), Cint, (Cstring, Cstring, Cstring), string(nameof(f)), stmt, helpers)
    id < 0 && error(unsafe_string(ccall((:exa_last_error, LIB), Cstring, ())))
    USER_UN[f] = Int32(id)
end
function register_bivariate(f, fx::String, d1::String, d2::String, d11::String, d12::String, d22::String; helpers::String = "")
    id = ccall((:exa_register_bivariate, LIB), Cint, (Cstring, Cstring, Cstring, Cstring, Cstring, Cstring, Cstring, Cstring),
               string(nameof(f)), fx, d1, d2, d11, d12, d22, helpers)
    id < 0 && error(unsafe_string(ccall((:exa_last_error, LIB), Cstring, ())))
    USER_BIN[f] = Int32(id)
end
fncode(tab, f, arity) = get(tab, f) do
    user = tab === UN ? USER_UN : USER_BIN
    haskey(user, f) && return user[f]
    special = tab === UN ? UN_SPECIAL : BIN_SPECIAL
    k = findfirst(==(nameof(f)), special)
    k === nothing && error("ExaModelsHIP: the $arity function `$f` is not in libexahip's derivative table (src/functionlist.jl, ext/functionlist.jl) and was not given to ExaModelsHIP.register_univariate / register_bivariate")
    Int32(length(tab) + k - 1)
end
lower!(l, n::Node1{F}) where {F} = push_node!(l, OP_UN, fncode(UN, F.instance, "univariate"), lower!(l, n.inner), -1, 0.0, 0)
function lower!(l, n::Node2{F}) where {F}
    a = lower!(l, n.inner1); b = lower!(l, n.inner2)
    push_node!(l, OP_BIN, fncode(BIN, F.instance, "bivariate"), a, b, 0.0, 0)
end
lower!(l, n::SumNode) = foldl_nodes!(l, n.inners, BIN[+], 0.0)      # reduce(+, ...) (graph.jl:549-567)
lower!(l, n::ProdNode) = foldl_nodes!(l, n.inners, BIN[*], 1.0)
function foldl_nodes!(l, inners, fn, unit)
    isempty(inners) && return push_node!(l, OP_NULLV, 0, -1, -1, unit, 0)
    acc = lower!(l, inners[1])
    for k in 2:length(inners)
        acc = push_node!(l, OP_BIN, fn, acc, lower!(l, inners[k]), 0.0, 0)
    end
    return acc
end

getpath(e, p::Tuple{}) = e
getpath(e, p::Tuple) = getpath(getfield(e, p[1]), Base.tail(p))

# column of one access path: UnitRange iterators need no storage
function column(itr, p, keep)
    if itr isa AbstractRange && isempty(p)
        return CColumn(2, 0, C_NULL, first(itr), step(itr))
    end
    host = [getpath(e, p) for e in Array(itr)]            # AoS -> SoA transpose, once, on the host
    if eltype(host) <: Integer
        v = Int64.(vec(host)); push!(keep, v); return CColumn(0, 0, pointer(v), 0, 0)
    else
        v = Float64.(vec(host)); push!(keep, v); return CColumn(1, 0, pointer(v), 0, 0)
    end
end

# multi-index augmentation target -> one integer expression: 1 + sum stride_d (idx_d - 1)  (nlp.jl:2012-2015)
function lower_target!(l, first, dims)
    first isa Tuple || return lower!(l, first)
    acc = lower!(l, 1); stride = 1
    for (d, ix) in enumerate(first)
        t = push_node!(l, OP_BIN, BIN[-], lower!(l, ix), lower!(l, 1), 0.0, 0)
        t = push_node!(l, OP_BIN, BIN[*], lower!(l, stride), t, 0.0, 0)
        acc = push_node!(l, OP_BIN, BIN[+], acc, t, 0.0, 0)
        stride *= dims[d]
    end
    return acc
end

flatten(t::Tuple{}) = ()
flatten(t::Tuple) = (flatten(Base.tail(t))..., first(t))           # blocks are prepended (nlp.jl:536)

function ExaModels.build_extension(c::ExaCore{T,VT,B}; prod = false) where {T,VT,B<:HIPNativeBackend}
    T === Float64 || error("libexahip evaluates in Float64")
    objs, cons = collect(flatten(c.obj)), collect(flatten(c.cons))
    # merge the two insertion-ordered lists by the shared running nnzh counter (f.o2)
    blocks = Any[]; i = j = 1
    while i <= length(objs) || j <= length(cons)
        takeobj = j > length(cons) || (i <= length(objs) && objs[i].f.o2 <= cons[j].f.o2 &&
                                       !(objs[i].f.o2 == cons[j].f.o2 && cons[j].f.o2step == 0))
        push!(blocks, takeobj ? objs[i] : cons[j]); takeobj ? (i += 1) : (j += 1)
    end
    keep = Any[]; pats = CPattern[]; basepos = Dict{Int,Int32}()
    for (k, blk) in enumerate(blocks)
        l = Lower(CNode[], Any[])
        f = blk.f.f
        expr = f isa Pair ? f.second : f
        root = lower!(l, expr)
        kind, target, base = Int32(0), Int32(-1), Int32(-1)
        if blk isa Constraint
            kind = Int32(1)
            # an augmentation names its base block by its first row (f.o0, nlp.jl:1683).  A zero-length Constraint shares
            # that number with the block after it: it never takes the entry of a block that has rows, and a block with rows
            # always does — the rows an augmentation can target belong to the latter.
            (length(blk.itr) > 0 || !haskey(basepos, blk.f.o0)) && (basepos[blk.f.o0] = Int32(k - 1))
        elseif blk isa ConstraintAugmentation
            kind = Int32(2); target = lower_target!(l, f.first, blk.dims)
            base = get(basepos, blk.f.o0) do
                error("ExaModelsHIP: augmentation block $k has no base constraint starting at row $(blk.f.o0 + 1)")
            end
        end
        cols = [column(blk.itr, p, keep) for p in l.paths]
        push!(keep, l.nodes); push!(keep, cols)
        push!(pats, CPattern(kind, length(l.nodes), pointer(l.nodes), root, target, base, length(cols),
                             isempty(cols) ? C_NULL : pointer(cols), length(blk.itr)))
    end
    h = x -> (v = Array{Float64}(x); push!(keep, v); isempty(v) ? Ptr{Float64}(C_NULL) : pointer(v))
    desc = Ref(CModelDesc(c.nvar, c.npar, h(c.x0), h(c.lvar), h(c.uvar), h(c.θ), length(pats), c.minimize ? 1 : 0,
                          pointer(pats), h(c.y0), h(c.lcon), h(c.ucon)))
    id = Ref{Cint}(0)
    GC.@preserve keep pats desc begin
        st = ccall((:exa_new_from_table, LIB), Cint, (Ref{CModelDesc}, Ref{Cint}), desc, id)
    end
    st == 0 || error("exa_new_from_table: status $st: " * unsafe_string(ccall((:exa_last_error, LIB), Cstring, ())))
    # the library recomputes the running offsets; they must equal the core's (layout contract) — a hard error, not an
    # @assert that an optimised build would drop
    for (sym, want) in ((:exa_nnzh64, c.nnzh), (:exa_nnzj64, c.nnzj), (:exa_ncon64, c.ncon), (:exa_nvar64, c.nvar))
        got = sym === :exa_nnzh64 ? ccall((:exa_nnzh64, LIB), Int64, (Cint,), id[]) :
              sym === :exa_nnzj64 ? ccall((:exa_nnzj64, LIB), Int64, (Cint,), id[]) :
              sym === :exa_ncon64 ? ccall((:exa_ncon64, LIB), Int64, (Cint,), id[]) : ccall((:exa_nvar64, LIB), Int64, (Cint,), id[])
        got == want || (ccall((:exa_free, LIB), Cint, (Cint,), id[]); error("ExaModelsHIP: layout mismatch, $sym = $got, ExaCore has $want"))
    end
    ext = HIPExtension(id[], Any[])
    finalizer(e -> ccall((:exa_free, LIB), Cint, (Cint,), e.id), ext)
    return ext
end

# Every callback runs on the CALLING TASK's AMDGPU stream (AMDGPU.jl streams are task-local and non-blocking, so the
# library's default null stream would not be ordered against the arrays' producers and consumers).
usestream(m) = chk(ccall((:exa_set_stream, LIB), Cint, (Cint, Ptr{Cvoid}), m.ext.id, AMDGPU.stream().stream), "exa_set_stream")

# No tuning call is needed (the reference has none, ext/ExaModelsKernelAbstractions.jl:526-537): exa_new_from_table above decides the block
# order and the hess_coord! kernel at plan time (interleaved where the heaviest patterns walk the same stretch of x and the call streams
# < 1.5 GB; the chained kernel at three workgroups per CU beyond) and picks up any decision an earlier tune! persisted next to the cached
# module for this module / device / sizes.  tune! is a refinement:
# explicit, blocking tuning (block orders, hess_coord! kernel, product implementations); persisted by the library.
tune!(m; what = 7) = (usestream(m); chk(ccall((:exa_tune, LIB), Cint, (Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, what, C_NULL, C_NULL), "exa_tune"))
# grad!: 0 = gathered + FP64 atomics, 1 = gradient COO + sorted gather (the reference's scheme, deterministic), -1 = what tune! persisted
deterministic!(m, on::Bool = true) = chk(ccall((:exa_set_deterministic, LIB), Cint, (Cint, Cint), m.ext.id, on), "exa_set_deterministic")
grad_mode!(m, mode::Integer) = chk(ccall((:exa_set_grad_mode, LIB), Cint, (Cint, Cint), m.ext.id, mode), "exa_set_grad_mode")

# ---- multi-GPU: one Julia process per GPU (include/exahip.h "multi-GPU behind the ABI") ---------------------------------
# rank 0:  uid = comm_unique_id();  uid = MPI.bcast(uid, 0, comm)       every rank:  comm_init!(model, rank, world, uid)
# afterwards obj / grad! / cons_nln! / jprod / jtprod / hprod! return the COMPLETE result on every rank (ncclAllReduce over
# xGMI on the model's stream, inside libexahip); jac_coord! / hess_coord! fill this rank's slice (coo_local!(model) packs it).
function comm_unique_id()
    uid = Vector{UInt8}(undef, 128)
    chk(ccall((:exa_comm_unique_id, LIB), Cint, (Ptr{UInt8},), uid), "exa_comm_unique_id")
    return uid
end
comm_init!(m, rank::Integer, world::Integer, uid::Vector{UInt8}) =
    chk(ccall((:exa_comm_init, LIB), Cint, (Cint, Cint, Cint, Ptr{UInt8}), m.ext.id, rank, world, uid), "exa_comm_init")
comm_free!(m) = chk(ccall((:exa_comm_free, LIB), Cint, (Cint,), m.ext.id), "exa_comm_free")
# deferred completion (after `exa_set_reduce(id, 0)`): the collective plan of callback `which` issued on `buf` through the attached communicator
comm_complete!(m, which::Integer, buf) = (usestream(m); chk(ccall((:exa_comm_complete, LIB), Cint, (Cint, Cint, Ptr{Cdouble}), m.ext.id, which, pointer(buf)), "exa_comm_complete"))
coo_local!(m, on::Bool = true) = chk(ccall((:exa_set_coo_local, LIB), Cint, (Cint, Cint), m.ext.id, on), "exa_set_coo_local")
local_nnzj(m) = ccall((:exa_local_nnzj64, LIB), Int64, (Cint,), m.ext.id)
local_nnzh(m) = ccall((:exa_local_nnzh64, LIB), Int64, (Cint,), m.ext.id)
function coo_slices(m; hess::Bool = true)          # rows of (first global slot, first local position, length), 0-based
    np = ccall((:exa_npatterns, LIB), Cint, (Cint,), m.ext.id)
    out = Vector{Int64}(undef, 3 * max(np, 1))
    chk(ccall((:exa_coo_slices, LIB), Cint, (Cint, Cint, Ptr{Int64}), m.ext.id, hess, out), "exa_coo_slices")
    return permutedims(reshape(out[1:3np], 3, np))
end

# a sharded Jacobian (hess = false) / Hessian COO vector whole on every rank: all-gather-v of the ranks' slot ranges
function allgather_coo!(m, dest::AbstractVector, loc::AbstractVector; hess::Bool = true)
    chk(ccall((:exa_allgather_coo, LIB), Cint, (Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, hess, pointer(loc), pointer(dest)), "exa_allgather_coo"); dest
end
# :pieces (complete values in disjoint pieces: an all-gather completes) or :partial (partial sums: an all-reduce completes):
# how this rank leaves the output of callback `which` (0 obj 1 grad 2 cons 3 jac 4 hess 5 jprod 6 jtprod 7 hprod) on its own
shard_layout(m, which::Integer) = ccall((:exa_shard_layout, LIB), Cint, (Cint, Cint), m.ext.id, which) == 1 ? :pieces : :partial
# jtprod / hprod implementation: 0 atomics in the sweep, 1 COO + sorted gather, 2 owner-computes windows, -1 undecided
product_mode!(m, jtprod::Integer, hprod::Integer) = chk(ccall((:exa_set_product_mode, LIB), Cint, (Cint, Cint, Cint), m.ext.id, jtprod, hprod), "exa_set_product_mode")
# all five callbacks of a solver iteration from one sweep (NLPModelsIpoptLite.jl:28-40); returns the objective as a 1-element device array
function eval_all!(m, x::AbstractVector, y::AbstractVector, g::AbstractVector, c::AbstractVector, jac::AbstractVector, hess::AbstractVector; obj_weight = one(eltype(x)))
    usestream(m)
    f = ROCArray{Float64}(undef, 1)
    chk(ccall((:exa_eval_all, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
              m.ext.id, pointer(x), pointer(y), Float64(obj_weight), pointer(f), pointer(g), pointer(c), pointer(jac), pointer(hess)), "exa_eval_all")
    return f
end

# ---- the seven callbacks (device pointers; asynchronous on the task's stream except obj) ------------------------
const HM{T,VT} = AbstractExaModel{T,VT,E} where {E<:HIPExtension}
chk(st, what) = st == 0 || error("$what: status $st: " * unsafe_string(ccall((:exa_last_error, LIB), Cstring, ())))

function ExaModels.obj(m::HM, x::AbstractVector)
    usestream(m)
    out = Ref{Cdouble}(0)
    chk(ccall((:exa_obj, LIB), Cint, (Cint, Ptr{Cdouble}, Ref{Cdouble}), m.ext.id, pointer(x), out), "exa_obj")
    return out[]
end
function ExaModels.cons_nln!(m::HM, x::AbstractVector, c::AbstractVector)
    usestream(m)
    chk(ccall((:exa_cons, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(c)), "exa_cons"); c
end
function ExaModels.grad!(m::HM, x::AbstractVector, g::AbstractVector)
    usestream(m)
    chk(ccall((:exa_grad, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(g)), "exa_grad"); g
end
function ExaModels.jac_coord!(m::HM, x::AbstractVector, v::AbstractVector)
    usestream(m)
    chk(ccall((:exa_jac, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(v)), "exa_jac"); v
end
function ExaModels.hess_coord!(m::HM, x::AbstractVector, y::AbstractVector, v::AbstractVector; obj_weight = one(eltype(x)))
    usestream(m)
    chk(ccall((:exa_hess, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}),
              m.ext.id, pointer(x), pointer(y), Float64(obj_weight), pointer(v)), "exa_hess"); v
end
# objective-only form (nlp.jl:1906-1915): y == NULL, the constraint slots come back as zeros
function ExaModels.hess_coord!(m::HM, x::AbstractVector, v::AbstractVector; obj_weight = one(eltype(x)))
    usestream(m)
    chk(ccall((:exa_hess, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}),
              m.ext.id, pointer(x), Ptr{Cdouble}(C_NULL), Float64(obj_weight), pointer(v)), "exa_hess"); v
end
function ExaModels.jprod_nln!(m::HM, x::AbstractVector, v::AbstractVector, Jv::AbstractVector)
    usestream(m)
    chk(ccall((:exa_jprod, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(v), pointer(Jv)), "exa_jprod"); Jv
end
function ExaModels.jtprod_nln!(m::HM, x::AbstractVector, v::AbstractVector, Jtv::AbstractVector)
    usestream(m)
    chk(ccall((:exa_jtprod, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(v), pointer(Jtv)), "exa_jtprod"); Jtv
end
function ExaModels.hprod!(m::HM, x::AbstractVector, y::AbstractVector, v::AbstractVector, Hv::AbstractVector; obj_weight = one(eltype(x)))
    usestream(m)
    chk(ccall((:exa_hprod, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}),
              m.ext.id, pointer(x), pointer(y), pointer(v), Float64(obj_weight), pointer(Hv)), "exa_hprod"); Hv
end
# objective-only form (nlp.jl:1942-1952)
function ExaModels.hprod!(m::HM, x::AbstractVector, v::AbstractVector, Hv::AbstractVector; obj_weight = one(eltype(x)))
    usestream(m)
    chk(ccall((:exa_hprod, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}),
              m.ext.id, pointer(x), Ptr{Cdouble}(C_NULL), pointer(v), Float64(obj_weight), pointer(Hv)), "exa_hprod"); Hv
end
# set_value!(model, param, values) (nlp.jl:1279-1287) writes model.θ; the library keeps its own device copy of θ, so the
# update is forwarded — no rebuild, exactly as in the reference.  Values that already live on the device (a ROCArray: model.θ of a
# device model is one) go device-to-device on the model's stream (exa_set_value_dev: no PCIe hop, no synchronisation, capturable);
# host values through exa_set_value.
function ExaModels.set_value!(m::ExaModels.ExaModel{T,VT,E}, param::ExaModels.Parameter, values) where {T,VT,E<:HIPExtension}
    length(values) == param.length || throw(DimensionMismatch("expected $(param.length) elements, got $(length(values))"))
    dst = view(m.θ, param.offset+1:param.offset+param.length)
    copyto!(dst, values)
    if m.θ isa Array
        v = Array{Float64}(values)
        chk(ccall((:exa_set_value, LIB), Cint, (Cint, Int64, Ptr{Cdouble}, Int64), m.ext.id, param.offset, v, length(v)), "exa_set_value")
    else
        usestream(m)
        chk(ccall((:exa_set_value_dev, LIB), Cint, (Cint, Int64, Ptr{Cdouble}, Int64), m.ext.id, param.offset, pointer(dst), param.length), "exa_set_value_dev")
    end
    return nothing
end
# the library's own device-resident parameter vector (npar doubles; get_value's view, nlp.jl:1270-1277): wrap it with
# unsafe_wrap(ROCArray, convert(ROCPtr{Float64}, theta_ptr(m)), npar) to write parameters in place
theta_ptr(m) = ccall((:exa_theta_ptr, LIB), Ptr{Cdouble}, (Cint,), m.ext.id)
# how exa_eval_all produces grad! on this model (exa_eval_all_mode): 1 / 2 = inside the sweep's one launch, 0 / 3 = a launch in front, 4 = sorted gather
eval_all_mode(m) = ccall((:exa_eval_all_mode, LIB), Cint, (Cint,), m.ext.id)
# which hess_coord! kernel tune! chose (0 exa_hess, 1 exa_hesscl, 2 exa_hessc) and the unused dynamic LDS its launches carry (an occupancy throttle)
hess_kernel(m) = (ccall((:exa_hess_variant, LIB), Cint, (Cint,), m.ext.id), ccall((:exa_hess_throttle, LIB), Cint, (Cint,), m.ext.id))
# may model files bring device code of their own (exa_recipe_trust_code)?  Returns the setting in force before the call.
trust_model_code!(on::Bool) = ccall((:exa_recipe_trust_code, LIB), Cint, (Cint,), on) != 0
# what the compiled kernels of the model need and which flags their modules were built with (exa_build_audit): one line per kernel
function build_audit(m)
    n = ccall((:exa_build_audit, LIB), Cint, (Cint, Ptr{UInt8}, Cint), m.ext.id, C_NULL, 0)
    buf = Vector{UInt8}(undef, n + 1)
    ccall((:exa_build_audit, LIB), Cint, (Cint, Ptr{UInt8}, Cint), m.ext.id, buf, n + 1)
    return split(unsafe_string(pointer(buf)), '\n'; keepempty = false)
end
# the collective operations that complete callback `which` of a sharded model (exa_collective_plan): rows of (kind, offset, count, root),
# kind 0 in-place all-gather, 1 broadcast, 2 all-reduce(sum) — for a host that issues them itself (MPI.jl) after exa_set_reduce(id, 0)
function collective_plan(m, which::Integer)
    ops = Matrix{Int64}(undef, 4, 64)
    n = ccall((:exa_collective_plan, LIB), Cint, (Cint, Cint, Ptr{Int64}, Cint), m.ext.id, which, ops, 64)
    return permutedims(ops[:, 1:max(0, min(n, 64))])
end
function ExaModels.jac_structure!(m::HM, rows::AbstractVector, cols::AbstractVector)
    usestream(m)
    r, c = ROCArray{Int64}(undef, length(rows)), ROCArray{Int64}(undef, length(cols))
    chk(ccall((:exa_jac_structure64, LIB), Cint, (Cint, Ptr{Int64}, Ptr{Int64}), m.ext.id, pointer(r), pointer(c)), "exa_jac_structure64")
    copyto!(rows, r); copyto!(cols, c); rows, cols
end
function ExaModels.hess_structure!(m::HM, rows::AbstractVector, cols::AbstractVector)
    usestream(m)
    r, c = ROCArray{Int64}(undef, length(rows)), ROCArray{Int64}(undef, length(cols))
    chk(ccall((:exa_hess_structure64, LIB), Cint, (Cint, Ptr{Int64}, Ptr{Int64}), m.ext.id, pointer(r), pointer(c)), "exa_hess_structure64")
    copyto!(rows, r); copyto!(cols, c); rows, cols
end

end # module
