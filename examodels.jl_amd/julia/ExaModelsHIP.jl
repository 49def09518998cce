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
lower!(l, n::Node1{F}) where {F} = push_node!(l, OP_UN, UN[F.instance], lower!(l, n.inner), -1, 0.0, 0)
function lower!(l, n::Node2{F}) where {F}
    a = lower!(l, n.inner1); b = lower!(l, n.inner2)
    push_node!(l, OP_BIN, BIN[F.instance], a, b, 0.0, 0)
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
            kind = Int32(1); basepos[blk.f.o0] = Int32(k - 1)
        elseif blk isa ConstraintAugmentation
            kind = Int32(2); target = lower_target!(l, f.first, blk.dims); base = basepos[blk.f.o0]
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
    # the library recomputes the running offsets; they must equal the core's (layout contract)
    @assert ccall((:exa_nnzh64, LIB), Int64, (Cint,), id[]) == c.nnzh
    @assert ccall((:exa_nnzj64, LIB), Int64, (Cint,), id[]) == c.nnzj
    ext = HIPExtension(id[], Any[])
    finalizer(e -> ccall((:exa_free, LIB), Cint, (Cint,), e.id), ext)
    return ext
end

# ---- the seven callbacks (device pointers; asynchronous on the null stream except obj) ------------------------
const HM{T,VT} = AbstractExaModel{T,VT,E} where {E<:HIPExtension}
chk(st, what) = st == 0 || error("$what: status $st: " * unsafe_string(ccall((:exa_last_error, LIB), Cstring, ())))

function ExaModels.obj(m::HM, x::AbstractVector)
    out = Ref{Cdouble}(0)
    chk(ccall((:exa_obj, LIB), Cint, (Cint, Ptr{Cdouble}, Ref{Cdouble}), m.ext.id, pointer(x), out), "exa_obj")
    return out[]
end
function ExaModels.cons_nln!(m::HM, x::AbstractVector, c::AbstractVector)
    chk(ccall((:exa_cons, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(c)), "exa_cons"); c
end
function ExaModels.grad!(m::HM, x::AbstractVector, g::AbstractVector)
    chk(ccall((:exa_grad, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(g)), "exa_grad"); g
end
function ExaModels.jac_coord!(m::HM, x::AbstractVector, v::AbstractVector)
    chk(ccall((:exa_jac, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(v)), "exa_jac"); v
end
function ExaModels.hess_coord!(m::HM, x::AbstractVector, y::AbstractVector, v::AbstractVector; obj_weight = one(eltype(x)))
    chk(ccall((:exa_hess, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}),
              m.ext.id, pointer(x), pointer(y), Float64(obj_weight), pointer(v)), "exa_hess"); v
end
function ExaModels.jprod_nln!(m::HM, x::AbstractVector, v::AbstractVector, Jv::AbstractVector)
    chk(ccall((:exa_jprod, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(v), pointer(Jv)), "exa_jprod"); Jv
end
function ExaModels.jtprod_nln!(m::HM, x::AbstractVector, v::AbstractVector, Jtv::AbstractVector)
    chk(ccall((:exa_jtprod, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), m.ext.id, pointer(x), pointer(v), pointer(Jtv)), "exa_jtprod"); Jtv
end
function ExaModels.hprod!(m::HM, x::AbstractVector, y::AbstractVector, v::AbstractVector, Hv::AbstractVector; obj_weight = one(eltype(x)))
    chk(ccall((:exa_hprod, LIB), Cint, (Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}),
              m.ext.id, pointer(x), pointer(y), pointer(v), Float64(obj_weight), pointer(Hv)), "exa_hprod"); Hv
end
# set_value!(model, param, values) (nlp.jl:1279-1287) writes model.θ; the library keeps its own device copy of θ, so the
# update is forwarded (host values) — no rebuild, exactly as in the reference.
function ExaModels.set_value!(m::ExaModels.ExaModel{T,VT,E}, param::ExaModels.Parameter, values) where {T,VT,E<:HIPExtension}
    length(values) == param.length || throw(DimensionMismatch("expected $(param.length) elements, got $(length(values))"))
    copyto!(view(m.θ, param.offset+1:param.offset+param.length), values)
    v = Array{Float64}(values)
    chk(ccall((:exa_set_value, LIB), Cint, (Cint, Int64, Ptr{Cdouble}, Int64), m.ext.id, param.offset, v, length(v)), "exa_set_value")
    return nothing
end
function ExaModels.jac_structure!(m::HM, rows::AbstractVector, cols::AbstractVector)
    r, c = ROCArray{Int64}(undef, length(rows)), ROCArray{Int64}(undef, length(cols))
    chk(ccall((:exa_jac_structure64, LIB), Cint, (Cint, Ptr{Int64}, Ptr{Int64}), m.ext.id, pointer(r), pointer(c)), "exa_jac_structure64")
    copyto!(rows, r); copyto!(cols, c); rows, cols
end
function ExaModels.hess_structure!(m::HM, rows::AbstractVector, cols::AbstractVector)
    r, c = ROCArray{Int64}(undef, length(rows)), ROCArray{Int64}(undef, length(cols))
    chk(ccall((:exa_hess_structure64, LIB), Cint, (Cint, Ptr{Int64}, Ptr{Int64}), m.ext.id, pointer(r), pointer(c)), "exa_hess_structure64")
    copyto!(rows, r); copyto!(cols, c); rows, cols
end

end # module
