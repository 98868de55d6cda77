# ProximalAlgorithmsHIP.jl -- Julia-side binding of libproxgrad_hip.so (include/proxgrad_hip.h).
#
# STATUS: UNEXECUTED.  The build image has no Julia toolchain; this file is the reference-side glue a
# maintainer would add (INTEGRATION.md).  It defines device array / operator types whose methods `ccall`
# the C ABI, and iterator types that plug into `ProximalAlgorithms.IterativeAlgorithm` unchanged:
#
#     using ProximalAlgorithms, ProximalAlgorithmsHIP
#     f = HIPLeastSquares(A, b); g = HIPNormL1(lam)
#     x, it = HIPFastForwardBackward(tol = 1e-6)(x0 = zeros(Float32, n), f = f, g = g)
#     x, it = HIPFastForwardBackward(tol = 1e-6)(x0 = x0, f = f, g = g, Lf = Lf, extrapolation_sequence = FixedNesterovSequence(Float32))
#
module ProximalAlgorithmsHIP

using ProximalAlgorithms
using ProximalCore
using LinearAlgebra
using Printf

const libpg = get(ENV, "PROXGRAD_HIP_LIB", "libproxgrad_hip.so")

const PG_F32, PG_F64 = Int32(0), Int32(1)
pg_dtype(::Type{Float32}) = PG_F32
pg_dtype(::Type{Float64}) = PG_F64

struct ProxGradError <: Exception
    code::Int32
    msg::String
end

function check(status::Int32)
    status == 0 && return nothing
    throw(ProxGradError(status, unsafe_string(ccall((:pg_last_error, libpg), Cstring, ()))))
end

const PG_ABI_VERSION = Int32(4)   # include/proxgrad_hip.h: the version this file was written against

function __init__()   # a stale or mismatched build is refused at load, not at the first missing symbol
    found = ccall((:pg_abi_version, libpg), Int32, ())
    found == PG_ABI_VERSION || error("libproxgrad_hip ($libpg) has ABI version $found, ProximalAlgorithmsHIP.jl was written against $PG_ABI_VERSION: rebuild the library or point PROXGRAD_HIP_LIB at a matching build")
end

# ---------------------------------------------------------------- context ------------------------------------
mutable struct HIPContext
    handle::Ptr{Cvoid}
    function HIPContext(device::Integer = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:pg_ctx_create, libpg), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, h))
        ctx = new(h[])
        finalizer(c -> ccall((:pg_ctx_destroy, libpg), Int32, (Ptr{Cvoid},), c.handle), ctx)
    end
end
const DEFAULT_CTX = Ref{Union{Nothing,HIPContext}}(nothing)
default_ctx() = (DEFAULT_CTX[] === nothing && (DEFAULT_CTX[] = HIPContext()); DEFAULT_CTX[])

# ---------------------------------------------------------------- device vector ------------------------------
# The `Tx` of the iterators.  Owns its buffer unless `owner !== nothing` (views of library-owned state).
mutable struct HIPVector{T} <: AbstractVector{T}
    ctx::HIPContext
    ptr::Ptr{Cvoid}
    n::Int
    owner::Any
end
function HIPVector{T}(::UndefInitializer, n::Integer; ctx = default_ctx()) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pg_malloc, libpg), Int32, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx.handle, n * sizeof(T), p))
    v = HIPVector{T}(ctx, p[], n, nothing)
    finalizer(v -> v.owner === nothing && ccall((:pg_free, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), v.ctx.handle, v.ptr), v)
end
function HIPVector(x::Vector{T}; ctx = default_ctx()) where {T}
    v = HIPVector{T}(undef, length(x); ctx)
    GC.@preserve x check(ccall((:pg_memcpy_h2d, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                               ctx.handle, v.ptr, pointer(x), sizeof(x)))
    v
end
Base.size(v::HIPVector) = (v.n,)
Base.similar(v::HIPVector{T}) where {T} = HIPVector{T}(undef, v.n; ctx = v.ctx)
function Base.Array(v::HIPVector{T}) where {T}
    out = Vector{T}(undef, v.n)
    GC.@preserve out check(ccall((:pg_memcpy_d2h, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                                 v.ctx.handle, pointer(out), v.ptr, sizeof(out)))
    out
end
function Base.copyto!(dst::HIPVector{T}, src::HIPVector{T}) where {T}
    check(ccall((:pg_memcpy_d2d, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                dst.ctx.handle, dst.ptr, src.ptr, dst.n * sizeof(T)))
    dst
end
Base.copy(v::HIPVector) = copyto!(similar(v), v)
function LinearAlgebra.dot(x::HIPVector{T}, y::HIPVector{T}) where {T}
    out = Ref{Float64}(0)
    check(ccall((:pg_dot, libpg), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}),
                x.ctx.handle, pg_dtype(T), x.n, x.ptr, y.ptr, out))
    T(out[])
end
function LinearAlgebra.norm(x::HIPVector{T}, p::Real = 2) where {T}
    out = Ref{Float64}(0)
    if p == Inf
        check(ccall((:pg_nrminf, libpg), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ref{Float64}),
                    x.ctx.handle, pg_dtype(T), x.n, x.ptr, out))
        return T(out[])
    end
    check(ccall((:pg_nrm2sq, libpg), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ref{Float64}),
                x.ctx.handle, pg_dtype(T), x.n, x.ptr, out))
    sqrt(T(out[]))
end
# out .= a .* x .+ b .* y  (the broadcasts of the iteration bodies lower to this)
function axpby!(out::HIPVector{T}, a, x::HIPVector{T}, b, y::HIPVector{T}) where {T}
    check(ccall((:pg_axpby, libpg), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64, Ptr{Cvoid}),
                out.ctx.handle, pg_dtype(T), out.n, out.ptr, a, x.ptr, b, y.ptr))
    out
end

# ---------------------------------------------------------------- matrix + LeastSquares ----------------------
mutable struct HIPMatrix{T}
    ctx::HIPContext
    handle::Ptr{Cvoid}
    m::Int
    n::Int
end
function HIPMatrix(A::Matrix{T}; ctx = default_ctx()) where {T}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    m, n = size(A)
    check(ccall((:pg_mat_create, libpg), Int32, (Ptr{Cvoid}, Int32, Int64, Int64, Ref{Ptr{Cvoid}}), ctx.handle, pg_dtype(T), m, n, h))
    M = HIPMatrix{T}(ctx, h[], m, n)
    finalizer(M -> ccall((:pg_mat_destroy, libpg), Int32, (Ptr{Cvoid},), M.handle), M)
    GC.@preserve A check(ccall((:pg_mat_upload, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), M.handle, pointer(A), max(m, 1)))
    M
end

# mul!(y, A, x) / mul!(g, A', r) on the device matrix (panoc.jl:180,186 ...)
function LinearAlgebra.mul!(y::HIPVector{T}, A::HIPMatrix{T}, x::HIPVector{T}) where {T}
    check(ccall((:pg_mat_mul, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), A.handle, x.ptr, y.ptr)); y
end
function LinearAlgebra.mul!(g::HIPVector{T}, At::Adjoint{T,HIPMatrix{T}}, r::HIPVector{T}) where {T}
    check(ccall((:pg_mat_mul_adjoint, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), parent(At).handle, r.ptr, g.ptr)); g
end

# ys[k] = A xs[k] for up to three vectors on ONE read of A (pg_mat_mul_multi; each bit-identical to mul!'s): the candidates of the
# step-size search (fb_tools.jl:46-55) evaluated ahead
function mul_multi!(ys::Vector{HIPVector{T}}, A::HIPMatrix{T}, xs::Vector{HIPVector{T}}) where {T}
    xp = Ptr{Cvoid}[x.ptr for x in xs]; yp = Ptr{Cvoid}[y.ptr for y in ys]
    GC.@preserve xp yp check(ccall((:pg_mat_mul_multi, libpg), Int32, (Ptr{Cvoid}, Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}), A.handle, length(xs), xp, yp))
    ys
end

# The single sweep for x -> loss(A x) terms: At_r = A'r, y = x - gamma At_r, z = prox_{gamma g}(y), res = x - z and Az = A z in
# ONE read of A (g_kind: 0 Zero, 1 NormL1(p0 = lam), 2 IndBox(p0 = lo, p1 = hi)); returns (g(z), norm(res, Inf), dot(At_r, res),
# norm(res)^2).  This is what lets FB / FFB / Vu-Condat / LiLin on such terms run at one read of A per iteration.
# image_of_res = true: the last output is A (x - z), the image of the forward-backward residual as a product of the residual
# itself (pg_mat_fused_tn_res) -- what PANOC with the L-BFGS image slab runs on (HIPLBFGSOperator below).
function fused_tn!(A::HIPMatrix{T}, r, x, gamma, g_kind, p0, p1, At_r, y, z, res, Az; image_of_res::Bool = false) where {T}
    sc = zeros(Float64, 4)
    if image_of_res
        check(ccall((:pg_mat_fused_tn_res, libpg), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int32, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}),
                    A.handle, r.ptr, x.ptr, gamma, g_kind, p0, p1, At_r.ptr, y.ptr, z.ptr, res.ptr, Az.ptr, sc))
    else
        check(ccall((:pg_mat_fused_tn, libpg), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int32, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}),
                    A.handle, r.ptr, x.ptr, gamma, g_kind, p0, p1, At_r.ptr, y.ptr, z.ptr, res.ptr, Az.ptr, sc))
    end
    (sc[1], sc[2], sc[3], sc[4])
end

# TWO instances of fused_tn! on ONE read of A (pg_mat_fused_tn_pair): the trial points of tau and tau / 2 of ZeroFPR's line search
# (zerofpr.jl:200-217).  out1 / out2 = (At_r, y, z, res, Az); returns the two scalar quadruples.  Columns of 33 .. 64 KiB only
# (config 4's 16384 Float32 rows); a ProxGradError with code PG_ERR_UNSUPPORTED means: one trial point per sweep.
function fused_tn_pair!(A::HIPMatrix{T}, r1, x1, r2, x2, gamma, g_kind, p0, p1, out1, out2; image_of_res::Bool = false) where {T}
    sc = zeros(Float64, 8)
    if image_of_res   # the last output of each instance is A (x - z): PANOCplus' speculative first sweep (pg_mat_fused_tn_pair_res)
        check(ccall((:pg_mat_fused_tn_pair_res, libpg), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int32, Float64, Float64,
                     Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}),
                    A.handle, r1.ptr, x1.ptr, r2.ptr, x2.ptr, gamma, g_kind, p0, p1,
                    out1[1].ptr, out1[2].ptr, out1[3].ptr, out1[4].ptr, out1[5].ptr, out2[1].ptr, out2[2].ptr, out2[3].ptr, out2[4].ptr, out2[5].ptr, sc))
        return ((sc[1], sc[2], sc[3], sc[4]), (sc[5], sc[6], sc[7], sc[8]))
    end
    check(ccall((:pg_mat_fused_tn_pair, libpg), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int32, Float64, Float64,
                 Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}),
                A.handle, r1.ptr, x1.ptr, r2.ptr, x2.ptr, gamma, g_kind, p0, p1,
                out1[1].ptr, out1[2].ptr, out1[3].ptr, out1[4].ptr, out1[5].ptr, out2[1].ptr, out2[2].ptr, out2[3].ptr, out2[4].ptr, out2[5].ptr, sc))
    ((sc[1], sc[2], sc[3], sc[4]), (sc[5], sc[6], sc[7], sc[8]))
end

# THREE instances on ONE read of A (pg_mat_fused_tn_trio): tau, tau / 2 and tau / 4 of the same line search.  rs, xs: three vectors
# each; outs[k] = (At_r, y, z, res, Az) of instance k; returns the three scalar quadruples.  Same column lengths as fused_tn_pair!.
function fused_tn_trio!(A::HIPMatrix{T}, rs, xs, gamma, g_kind, p0, p1, outs; image_of_res::Bool = false) where {T}
    sc = zeros(Float64, 12)
    ptrs(vs) = Ptr{Cvoid}[v.ptr for v in vs]
    cols = [ptrs([outs[k][i] for k in 1:3]) for i in 1:5]
    rp, xp = ptrs(rs), ptrs(xs)
    GC.@preserve rp xp cols check(ccall((:pg_mat_fused_tn_trio, libpg), Int32,
                (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Float64, Int32, Float64, Float64,
                 Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Int32, Ptr{Float64}),
                A.handle, rp, xp, gamma, g_kind, p0, p1, cols[1], cols[2], cols[3], cols[4], cols[5], Int32(image_of_res), sc))
    ((sc[1], sc[2], sc[3], sc[4]), (sc[5], sc[6], sc[7], sc[8]), (sc[9], sc[10], sc[11], sc[12]))
end

# One Davis-Yin iteration (davis_yin.jl:73-83) in ONE read of A; prox kinds as above plus 3 = SqrNormL2(p0 = lam)
function fused_dys!(A::HIPMatrix{T}, r, xg, z, gamma, relax, g_spec, h_spec, grad, z_half, xh, res, z_next, xg_next, A_xg_next) where {T}
    sc = zeros(Float64, 4)
    check(ccall((:pg_mat_fused_dys, libpg), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int32, Float64, Float64, Int32, Float64, Float64,
                 Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}),
                A.handle, r.ptr, xg.ptr, z.ptr, gamma, relax, g_spec[1], g_spec[2], g_spec[3], h_spec[1], h_spec[2], h_spec[3],
                grad.ptr, z_half.ptr, xh.ptr, res.ptr, z_next.ptr, xg_next.ptr, A_xg_next.ptr, sc))
    (sc[2], sc[3], sc[4])
end

# f(x) = lam/2 ||A x - b||^2 : replaces ProximalOperators.LeastSquares + benchmark/benchmarks.jl:11-17
mutable struct HIPLeastSquares{T}
    A::HIPMatrix{T}
    b::HIPVector{T}
    handle::Ptr{Cvoid}
end
function HIPLeastSquares(A::Matrix{T}, b::Vector{T}, lam::Real = 1) where {T}
    Ad = HIPMatrix(A)
    bd = HIPVector(b; ctx = Ad.ctx)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pg_ls_create, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ref{Ptr{Cvoid}}),
                Ad.ctx.handle, Ad.handle, bd.ptr, lam, h))
    f = HIPLeastSquares{T}(Ad, bd, h[])
    finalizer(f -> ccall((:pg_ls_destroy, libpg), Int32, (Ptr{Cvoid},), f.handle), f)
end
function ProximalAlgorithms.value_and_gradient(f::HIPLeastSquares{T}, x::HIPVector{T}) where {T}
    grad = similar(x)
    fx = Ref{Float64}(0)
    check(ccall((:pg_ls_value_and_gradient, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}),
                f.handle, x.ptr, grad.ptr, fx))
    T(fx[]), grad
end
function (f::HIPLeastSquares{T})(x::HIPVector{T}) where {T}
    fx = Ref{Float64}(0)
    check(ccall((:pg_ls_value, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), f.handle, x.ptr, fx))
    T(fx[])
end

# ---------------------------------------------------------------- prox operators -----------------------------
# NormL1(lambda): a scalar, or per-element weights (a HIPVector), like ProximalOperators.NormL1(lambda::AbstractArray)
struct HIPNormL1{R}
    lambda::R
end
function ProximalCore.prox!(y::HIPVector{T}, g::HIPNormL1, x::HIPVector{T}, gamma) where {T}
    gy = Ref{Float64}(0)
    check(ccall((:pg_prox_norml1, libpg), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ref{Float64}),
                x.ctx.handle, pg_dtype(T), x.n, y.ptr, x.ptr, g.lambda, gamma, gy))
    T(gy[])
end
function ProximalCore.prox!(y::HIPVector{T}, g::HIPNormL1{<:HIPVector}, x::HIPVector{T}, gamma) where {T}
    gy = Ref{Float64}(0)
    check(ccall((:pg_prox_norml1w, libpg), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ref{Float64}),
                x.ctx.handle, pg_dtype(T), x.n, y.ptr, x.ptr, g.lambda.ptr, gamma, gy))
    T(gy[])
end
# IndBox(lo, hi): scalar bounds or per-element bounds (HIPVectors), like ProximalOperators.IndBox
struct HIPIndBox{R}
    lo::R
    hi::R
end
box_scalar(v) = v isa HIPVector ? 0.0 : Float64(v)
box_vector(v) = v isa HIPVector ? v.ptr : C_NULL
function ProximalCore.prox!(y::HIPVector{T}, g::HIPIndBox, x::HIPVector{T}, gamma) where {T}
    gy = Ref{Float64}(0)
    check(ccall((:pg_prox_indbox, libpg), Int32,
                (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}),
                x.ctx.handle, pg_dtype(T), x.n, y.ptr, x.ptr, box_scalar(g.lo), box_scalar(g.hi), box_vector(g.lo), box_vector(g.hi), gy))
    T(0)
end
ProximalCore.prox(g::Union{HIPNormL1,HIPIndBox}, x::HIPVector, gamma) = (y = similar(x); (y, ProximalCore.prox!(y, g, x, gamma)))

# ---------------------------------------------------------------- operator-level drop-in: broadcasts ---------
# With the methods above AND this section the reference's OWN ForwardBackwardIteration / FastForwardBackwardIteration
# (src/algorithms/forward_backward.jl, fast_forward_backward.jl, src/utilities/fb_tools.jl) run unmodified on HIPVector:
#
#     x, it = ProximalAlgorithms.FastForwardBackward(tol = 1e-6)(x0 = HIPVector(x0), f = HIPLeastSquares(A, b), g = HIPNormL1(lam))
#
# Every vector statement of those bodies is either a ProximalCore / LinearAlgebra call bound above (prox!, value_and_gradient,
# dot, norm, copy, similar) or a broadcast.  A broadcast over HIPVectors carries the style HIPStyle; `copyto!(dest, bc)` matches
# the tree of the EXACT shapes the path uses -- BROADCAST_TABLE below lists every such statement of the reference with the shape
# it lowers to and the C entry point that runs it -- and anything else is an error (never a scalar getindex loop).
# UNEXECUTED like the rest of this file; tests/test_julia_glue.py checks the table against the reference's source text, that every
# shape named there has a `lower!` method here, and that every broadcast statement of the path's files is in the table.
using Base.Broadcast: Broadcasted, AbstractArrayStyle

struct HIPStyle <: AbstractArrayStyle{1} end
HIPStyle(::Val{N}) where {N} = HIPStyle()
Base.BroadcastStyle(::Type{<:HIPVector}) = HIPStyle()

Base.getindex(::HIPVector, ::Int) =
    error("scalar indexing of a HIPVector is not supported (it would be one device read per element): copy it to the host with Array(v)")
Base.setindex!(::HIPVector, ::Any, ::Int) =
    error("scalar indexing of a HIPVector is not supported: build the vector on the host and upload it with HIPVector(x)")
Base.show(io::IO, ::MIME"text/plain", v::HIPVector{T}) where {T} = print(io, v.n, "-element HIPVector{", T, "} (device memory)")
Base.show(io::IO, v::HIPVector{T}) where {T} = print(io, "HIPVector{", T, "}(n = ", v.n, ")")

first_vector(v::HIPVector) = v
first_vector(::Any) = nothing
first_vector(bc::Broadcasted) = first_vector(bc.args)
first_vector(args::Tuple) = (v = first_vector(args[1]); v === nothing ? first_vector(Base.tail(args)) : v)
first_vector(::Tuple{}) = nothing
function Base.similar(bc::Broadcasted{HIPStyle}, ::Type{T}) where {T}
    v = first_vector(bc)
    HIPVector{T}(undef, v.n; ctx = v.ctx)
end

function add_scalar!(out::HIPVector{T}, x::HIPVector{T}, c) where {T}
    check(ccall((:pg_add_scalar, libpg), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Float64),
                out.ctx.handle, pg_dtype(T), out.n, out.ptr, x.ptr, c))
    out
end
# x .= z .+ beta .* (z .- z_prev)
function extrapolate!(x::HIPVector{T}, z::HIPVector{T}, z_prev::HIPVector{T}, beta) where {T}
    check(ccall((:pg_extrapolate, libpg), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64),
                x.ctx.handle, pg_dtype(T), x.n, x.ptr, z.ptr, z_prev.ptr, beta))
    x
end

# the shapes, as types of the Broadcasted tree (V: a HIPVector leaf, S: a scalar leaf)
const BcNode{F,Args} = Broadcasted{HIPStyle,<:Any,F,Args}
const V = HIPVector
const S = Number
const ScaledV = BcNode{typeof(*),<:Tuple{S,V}}                       # a .* x
const DiffVV = BcNode{typeof(-),<:Tuple{V,V}}                        # x .- y

shape_string(x::HIPVector) = "V"
shape_string(x::Number) = "S"
shape_string(x) = string(typeof(x))
shape_string(bc::Broadcasted) = "(" * join(map(shape_string, bc.args), " ." * string(bc.f) * " ") * ")"

lower!(dest, bc) =
    error("unsupported broadcast on HIPVector: ", shape_string(bc), " -- the shapes of the proximal-gradient path are listed in BROADCAST_TABLE; ",
          "compute anything else on the host (Array(v)) or add a kernel for it")
lower!(dest::V, bc::BcNode{typeof(identity),<:Tuple{V}}) = copyto!(dest, bc.args[1])                          # :copy         x
lower!(dest::V, bc::BcNode{typeof(*),<:Tuple{S,V}}) = axpby!(dest, bc.args[1], bc.args[2], 0, bc.args[2])     # :scale        a .* x
lower!(dest::V, bc::BcNode{typeof(*),<:Tuple{V,S}}) = axpby!(dest, bc.args[2], bc.args[1], 0, bc.args[1])     # :scale_right  x .* a
lower!(dest::V, bc::BcNode{typeof(-),<:Tuple{V,V}}) = axpby!(dest, 1, bc.args[1], -1, bc.args[2])             # :sub          x .- y
lower!(dest::V, bc::BcNode{typeof(+),<:Tuple{V,S}}) = add_scalar!(dest, bc.args[1], bc.args[2])               # :add_scalar   x .+ c
lower!(dest::V, bc::BcNode{typeof(-),<:Tuple{V,<:ScaledV}}) =                                                  # :axmy         x .- a .* y
    axpby!(dest, 1, bc.args[1], -bc.args[2].args[1], bc.args[2].args[2])
lower!(dest::V, bc::BcNode{typeof(+),<:Tuple{V,<:ScaledV}}) =                                                  # :axpy         x .+ a .* y
    axpby!(dest, 1, bc.args[1], bc.args[2].args[1], bc.args[2].args[2])
function lower!(dest::V, bc::BcNode{typeof(+),<:Tuple{V,<:BcNode{typeof(*),<:Tuple{S,<:DiffVV}}}})             # :extrapolate  z .+ b .* (z .- w)
    z, beta, d = bc.args[1], bc.args[2].args[1], bc.args[2].args[2]
    d.args[1] === z || error("unsupported broadcast on HIPVector: x .+ b .* (y .- w) with x !== y (the path has z .+ beta .* (z .- z_prev))")
    extrapolate!(dest, z, d.args[2], beta)
end
Base.copyto!(dest::HIPVector, bc::Broadcasted{HIPStyle}) = (length(dest) == length(first_vector(bc)) || throw(DimensionMismatch()); lower!(dest, bc))

# Every broadcast statement of the reference's files on the path: (file, line, statement, shape(s) it lowers to, C entry point).
# A statement with an un-dotted operator materialises its dotted part first (`x - gamma .* g` is `t = gamma .* g; x - t`).
const BROADCAST_TABLE = [
    ("src/algorithms/forward_backward.jl", 71, "y = x - gamma .* grad_f_x", (:scale, :sub), :pg_axpby),
    ("src/algorithms/forward_backward.jl", 114, "state.grad_f_x .= grad_f_x", (:copy,), :pg_memcpy_d2d),
    ("src/algorithms/forward_backward.jl", 117, "state.y .= state.x .- state.gamma .* state.grad_f_x", (:axmy,), :pg_axpby),
    ("src/algorithms/forward_backward.jl", 120, "state.res .= state.x .- state.z", (:sub,), :pg_axpby),
    ("src/algorithms/fast_forward_backward.jl", 79, "y = x - gamma .* grad_f_x", (:scale, :sub), :pg_axpby),
    ("src/algorithms/fast_forward_backward.jl", 135, "state.x .= state.z .+ beta .* (state.z .- state.z_prev)", (:extrapolate,), :pg_extrapolate),
    ("src/algorithms/fast_forward_backward.jl", 139, "state.grad_f_x .= grad_f_x", (:copy,), :pg_memcpy_d2d),
    ("src/algorithms/fast_forward_backward.jl", 140, "state.y .= state.x .- state.gamma .* state.grad_f_x", (:axmy,), :pg_axpby),
    ("src/algorithms/fast_forward_backward.jl", 142, "state.res .= state.x .- state.z", (:sub,), :pg_axpby),
    ("src/utilities/fb_tools.jl", 9, "xeps = x .+ 1", (:add_scalar,), :pg_add_scalar),
    ("src/utilities/fb_tools.jl", 48, "y .= x .- gamma .* At_grad_f_Ax", (:axmy,), :pg_axpby),
    ("src/utilities/fb_tools.jl", 50, "res .= x .- z", (:sub,), :pg_axpby),
    ("src/utilities/fb_tools.jl", 57, "grad_f_Az .= grad_f_Az_tmp", (:copy,), :pg_memcpy_d2d),
    ("src/utilities/fb_tools.jl", 78, "y = x - gamma .* At_grad_f_Ax", (:scale, :sub), :pg_axpby),
    ("src/accel/lbfgs.jl", 31, "L.s .= s", (:copy,), :pg_memcpy_d2d),
    ("src/accel/lbfgs.jl", 32, "L.y .= y", (:copy,), :pg_memcpy_d2d),
    ("src/accel/lbfgs.jl", 65, "d .= v", (:copy,), :pg_memcpy_d2d),
    ("src/accel/lbfgs.jl", 67, "d .*= L.H", (:scale_right,), :pg_axpby),
    ("src/accel/lbfgs.jl", 76, "d .-= L.alphas[idx] .* L.y_M[idx]", (:axmy,), :pg_axpby),
    ("src/accel/lbfgs.jl", 92, "d .+= (L.alphas[idx] - beta) .* L.s_M[idx]", (:axpy,), :pg_axpby),
]
# Un-dotted array arithmetic of the same files goes through Base's `-(::AbstractArray, ::AbstractArray)` and
# `*(::UniformScaling, ::AbstractVector)`, i.e. through the same broadcasts: `res = x - z` (forward_backward.jl:81,
# fast_forward_backward.jl:89) is :sub; `A * xeps` and `A' * (grad_f_Axeps - grad_f_Ax)` with `A === I` (fb_tools.jl:10-11) are
# :scale with a = true and :sub.

# ---------------------------------------------------------------- fused iterators ----------------------------
# C structs of include/proxgrad_hip.h
struct PgIterOpts
    fast::Int32; adaptive::Int32; Lf::Float64; gamma::Float64; minimum_gamma::Float64; reduce_gamma::Float64
    increase_gamma::Float64; mf::Float64; seq_kind::Int32; seq_p0::Float64; seq_p1::Float64
    g_kind::Int32; g_p0::Float64; g_p1::Float64; reuse_residual::Int32; single_sweep::Int32
end
struct PgIterScalars
    gamma::Float64; f_x::Float64; g_z::Float64; res_inf::Float64; beta::Float64; f_z::Float64; f_z_upp::Float64
    n_backtracks::Int32; flags::Int32; a_passes::Int64
end
struct PgIterState
    x::Ptr{Cvoid}; grad_f_x::Ptr{Cvoid}; y::Ptr{Cvoid}; z::Ptr{Cvoid}; res::Ptr{Cvoid}; z_prev::Ptr{Cvoid}; grad_f_z::Ptr{Cvoid}
end
g_spec(g::HIPNormL1) = (Int32(1), Float64(g.lambda), 0.0)
g_spec(g::HIPNormL1{<:HIPVector}) = (Int32(1), 0.0, 0.0)
g_spec(g::HIPIndBox) = (Int32(2), box_scalar(g.lo), box_scalar(g.hi))
# per-element bounds reach the fused iteration through pg_iter_set_g_vectors (after create, before init)
set_g_vectors!(h, g) = nothing
set_g_vectors!(h, g::HIPIndBox{<:HIPVector}) =
    check(ccall((:pg_iter_set_g_vectors, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, g.lo.ptr, g.hi.ptr))
set_g_vectors!(h, g::HIPNormL1{<:HIPVector}) =   # per-element weights: the first vector, the second stays null
    check(ccall((:pg_iter_set_g_vectors, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, g.lambda.ptr, C_NULL))

# extrapolation sequences (src/accel/nesterov.jl) -> (seq_kind, seq_p0, seq_p1, host-side iterator or nothing)
const PG_SEQ_ADAPTIVE, PG_SEQ_FIXED, PG_SEQ_SIMPLE, PG_SEQ_CONSTANT, PG_SEQ_HOST, PG_SEQ_REPEATED =
    Int32(0), Int32(1), Int32(2), Int32(3), Int32(4), Int32(5)
# (R = real(eltype(x0)): the library evaluates the recurrences in the working precision, like a sequence of type R)
seq_spec(::Nothing, R) = (PG_SEQ_ADAPTIVE, 0.0, 0.0, nothing)   # AdaptiveNesterovSequence(mf), fast_forward_backward.jl:94-95
seq_spec(::ProximalAlgorithms.FixedNesterovSequence{R}, ::Type{R}) where {R} = (PG_SEQ_FIXED, 0.0, 0.0, nothing)    # nesterov.jl:14-17
seq_spec(::ProximalAlgorithms.SimpleNesterovSequence{R}, ::Type{R}) where {R} = (PG_SEQ_SIMPLE, 0.0, 0.0, nothing)  # nesterov.jl:36
# ConstantNesterovSequence(m, stepsize) IS `repeated(beta)` (nesterov.jl:51-54): the library repeats the same value
seq_spec(s::Base.Iterators.Repeated{R}, ::Type{R}) where {R} = (PG_SEQ_REPEATED, Float64(s.x), 0.0, nothing)
# any other iterator (or a sequence of another precision): the coefficient of each step is drawn on the host
# (fast_forward_backward.jl:91-93,99-101) and passed to pg_iter_step; the library then iterates with two sweeps over A
# (the single sweep needs the NEXT coefficient one step early)
seq_spec(s, R) = (PG_SEQ_HOST, 0.0, 0.0, Iterators.Stateful(s))

# keyword constructors mirror ForwardBackwardIteration (forward_backward.jl:38-48) and FastForwardBackwardIteration
# (fast_forward_backward.jl:44-56): same names, same defaults
Base.@kwdef struct HIPForwardBackwardIteration{R,Tf,Tg,Tx}
    f::Tf
    g::Tg
    x0::Tx
    Lf::Union{Nothing,R} = nothing
    gamma::Union{Nothing,R} = Lf === nothing ? nothing : (1 / Lf)
    adaptive::Bool = gamma === nothing
    minimum_gamma::R = real(eltype(x0))(1e-7)
    reduce_gamma::R = real(eltype(x0))(0.5)
    increase_gamma::R = real(eltype(x0))(1.0)
end
Base.@kwdef struct HIPFastForwardBackwardIteration{R,Tf,Tg,Tx,Textr}
    f::Tf
    g::Tg
    x0::Tx
    mf::R = real(eltype(x0))(0)
    Lf::Union{Nothing,R} = nothing
    gamma::Union{Nothing,R} = Lf === nothing ? nothing : (1 / Lf)
    adaptive::Bool = gamma === nothing
    minimum_gamma::R = real(eltype(x0))(1e-7)
    reduce_gamma::R = real(eltype(x0))(0.5)
    increase_gamma::R = real(eltype(x0))(1.0)
    extrapolation_sequence::Textr = nothing
end
const HIPIteration = Union{HIPForwardBackwardIteration,HIPFastForwardBackwardIteration}
Base.IteratorSize(::Type{<:HIPForwardBackwardIteration}) = Base.IsInfinite()       # forward_backward.jl:50
Base.IteratorSize(::Type{<:HIPFastForwardBackwardIteration}) = Base.IsInfinite()   # fast_forward_backward.jl:58
is_fast(::HIPForwardBackwardIteration) = false
is_fast(::HIPFastForwardBackwardIteration) = true

# ForwardBackwardState (forward_backward.jl:52-63) / FastForwardBackwardState (fast_forward_backward.jl:60-71): the same
# field names; the vectors are non-owning views of the library's state slab and follow its pointer swaps.  z_prev and
# extrapolation_sequence exist for the fast iteration only (z_prev === nothing, extrapolation_sequence === nothing else).
mutable struct HIPIterState{R,T}
    handle::Ptr{Cvoid}
    x::HIPVector{T}; f_x::R; grad_f_x::HIPVector{T}; gamma::R; y::HIPVector{T}; z::HIPVector{T}; g_z::R; res::HIPVector{T}
    z_prev::Union{Nothing,HIPVector{T}}
    extrapolation_sequence::Any   # Iterators.Stateful for host-drawn coefficients, else the (kind, p0, p1) the library runs
    res_inf::R
    keepalive::Any                # iter.g: the library BORROWS its per-element vectors (pg_iter_set_g_vectors) for the life of the handle
end

function refresh!(st::HIPIterState{R,T}, sc::PgIterScalars, ctx, n) where {R,T}
    v = Ref(PgIterState(C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
    check(ccall((:pg_iter_state_view, libpg), Int32, (Ptr{Cvoid}, Ref{PgIterState}), st.handle, v))
    mk(p) = HIPVector{T}(ctx, p, n, st)   # non-owning views, pointers follow the library's swaps
    st.x, st.grad_f_x, st.y, st.z, st.res = mk(v[].x), mk(v[].grad_f_x), mk(v[].y), mk(v[].z), mk(v[].res)
    st.z_prev = v[].z_prev == C_NULL ? nothing : mk(v[].z_prev)
    st.f_x, st.gamma, st.g_z, st.res_inf = R(sc.f_x), R(sc.gamma), R(sc.g_z), R(sc.res_inf)
    st
end

function Base.iterate(iter::HIPIteration)
    T = eltype(iter.x0)
    R = real(T)
    x0 = iter.x0 isa HIPVector ? iter.x0 : HIPVector(iter.x0; ctx = iter.f.A.ctx)   # x0 is copied, never mutated
    kind, p0, p1 = g_spec(iter.g)
    fast = is_fast(iter)
    skind, sp0, sp1, host_seq = fast ? seq_spec(iter.extrapolation_sequence, R) : (PG_SEQ_ADAPTIVE, 0.0, 0.0, nothing)
    opts = Ref(PgIterOpts(fast, iter.adaptive, something(iter.Lf, -1.0), something(iter.gamma, -1.0),
                          iter.minimum_gamma, iter.reduce_gamma, iter.increase_gamma, fast ? iter.mf : 0.0,
                          skind, sp0, sp1, kind, p0, p1, 1, 1))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pg_iter_create, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{PgIterOpts}, Ref{Ptr{Cvoid}}),
                iter.f.A.ctx.handle, iter.f.handle, opts, h))
    set_g_vectors!(h[], iter.g)
    sc = Ref{PgIterScalars}()
    check(ccall((:pg_iter_init, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{PgIterScalars}), h[], x0.ptr, sc))
    st = HIPIterState{R,T}(h[], x0, R(0), x0, R(0), x0, x0, R(0), x0, nothing,
                           host_seq === nothing ? (skind, sp0, sp1) : host_seq, R(0), iter.g)
    finalizer(s -> ccall((:pg_iter_destroy, libpg), Int32, (Ptr{Cvoid},), s.handle), st)
    refresh!(st, sc[], iter.f.A.ctx, length(x0))
    return st, st
end
function Base.iterate(iter::HIPIteration, st::HIPIterState)
    # get_next_extrapolation_coefficient! (fast_forward_backward.jl:99-104) for host-side sequences
    beta = st.extrapolation_sequence isa Iterators.Stateful ? Float64(first(st.extrapolation_sequence)) : 0.0
    sc = Ref{PgIterScalars}()
    check(ccall((:pg_iter_step, libpg), Int32, (Ptr{Cvoid}, Float64, Ref{PgIterScalars}), st.handle, beta, sc))
    (sc[].flags & 1) != 0 && @warn "stepsize `gamma` became too small ($(sc[].gamma))"   # fb_tools.jl:59-61
    refresh!(st, sc[], iter.f.A.ctx, st.x.n)
    return st, st
end

default_stopping_criterion(tol, ::HIPIteration, st::HIPIterState) = st.res_inf / st.gamma <= tol
# fast_forward_backward.jl:152 / forward_backward.jl:128 return `state.z` itself (aliased, not copied): so does the glue -- a
# HIPVector view of the library-owned state, valid until the next step.  `host_solution` is the explicit host copy.
default_solution(::HIPIteration, st::HIPIterState) = st.z
host_solution(iter::HIPIteration, st::HIPIterState) = Array(default_solution(iter, st))
default_display(it, ::HIPIteration, st::HIPIterState) =
    @printf("%5d | %.3e | %.3e\n", it, st.gamma, st.res_inf / st.gamma)

HIPForwardBackward(; maxit = 10_000, tol = 1e-8, stop = (iter, state) -> default_stopping_criterion(tol, iter, state),
                   solution = default_solution, verbose = false, freq = 100, display = default_display, kwargs...) =
    ProximalAlgorithms.IterativeAlgorithm(HIPForwardBackwardIteration; maxit, stop, solution, verbose, freq, display, kwargs...)
HIPFastForwardBackward(; maxit = 10_000, tol = 1e-8, stop = (iter, state) -> default_stopping_criterion(tol, iter, state),
                       solution = default_solution, verbose = false, freq = 100, display = default_display, kwargs...) =
    ProximalAlgorithms.IterativeAlgorithm(HIPFastForwardBackwardIteration; maxit, stop, solution, verbose, freq, display, kwargs...)

# The driver loop of ProximalAlgorithms.jl:114-123 with the default stopping rule, inside the library: no host round
# trip per iteration.  Launch-bound sizes run as ONE kernel launch (pg_iter_run_small: one workgroup;
# pg_iter_run_coop: cooperating workgroups with grid barriers while A is cache-resident); larger problems use the
# streaming kernels (pg_iter_run).  Returns (solution, k) like IterativeAlgorithm.
function hip_solve(iter::HIPIteration; maxit = 10_000, tol = 1e-8)
    st, _ = iterate(iter)                                  # k = 1 (the state after init)
    A = iter.f.A
    T = eltype(iter.x0)
    nbytes = A.m * A.n * sizeof(T)
    k = Ref{Int64}(0)
    sc = Ref{PgIterScalars}()
    # the one-launch solvers take scalar parameters of g only (they return PG_ERR_UNSUPPORTED once pg_iter_set_g_vectors was
    # called): per-element bounds / weights run through pg_iter_run
    g_has_vectors = (iter.g isa HIPIndBox{<:HIPVector}) || (iter.g isa HIPNormL1{<:HIPVector})
    if !g_has_vectors && A.m * A.n <= 8192
        check(ccall((:pg_iter_run_small, libpg), Int32, (Ptr{Cvoid}, Int64, Int64, Float64, Ref{Int64}, Ref{PgIterScalars}),
                    st.handle, 1, maxit, tol, k, sc))
    elseif !g_has_vectors && 3 * cld(A.m, 64) * 64 * sizeof(T) <= 96 * 1024 && nbytes <= (iter.adaptive ? 10 : 6) << 20
        check(ccall((:pg_iter_run_coop, libpg), Int32, (Ptr{Cvoid}, Int64, Int64, Float64, Int32, Ref{Int64}, Ref{PgIterScalars}),
                    st.handle, 1, maxit, tol, 0, k, sc))
    else
        check(ccall((:pg_iter_run, libpg), Int32, (Ptr{Cvoid}, Int64, Int64, Float64, Ref{Int64}, Ref{PgIterScalars}),
                    st.handle, 1, maxit, tol, k, sc))
    end
    refresh!(st, sc[], A.ctx, st.x.n)
    return st.z, Int(k[])   # the aliased device vector, like IterativeAlgorithm's `solution(iter, state)`; Array(z) copies to the host
end

# Checkpoint / resume.  In the reference `iterate(iter, saved_state)` continues from any saved state (all algorithm memory is
# in the state struct: fast_forward_backward.jl:60-71, nesterov.jl:56-60).  The library's state lives on the device, so the
# saved form is a host blob: save_state(st) :: Vector{UInt8}; resume(iter, blob) creates a fresh library iterator with
# `iter`'s options, uploads the blob (pg_iter_state_upload; pg_iter_init is not needed) and returns the state object from
# which `iterate(iter, state)` continues bit-identically.  Host-drawn extrapolation sequences are not part of the blob.
function save_state(st::HIPIterState)
    nbytes = Ref{Int64}(0)
    check(ccall((:pg_iter_state_bytes, libpg), Int32, (Ptr{Cvoid}, Ref{Int64}), st.handle, nbytes))
    blob = Vector{UInt8}(undef, nbytes[])
    check(ccall((:pg_iter_state_download, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), st.handle, blob, nbytes[]))
    blob
end
function resume(iter::HIPIteration, blob::Vector{UInt8})
    T = eltype(iter.x0)
    R = real(T)
    x0 = iter.x0 isa HIPVector ? iter.x0 : HIPVector(iter.x0; ctx = iter.f.A.ctx)
    kind, p0, p1 = g_spec(iter.g)
    fast = is_fast(iter)
    skind, sp0, sp1, host_seq = fast ? seq_spec(iter.extrapolation_sequence, R) : (PG_SEQ_ADAPTIVE, 0.0, 0.0, nothing)
    host_seq === nothing || error("a host-drawn extrapolation sequence is not part of a saved state")
    opts = Ref(PgIterOpts(fast, iter.adaptive, something(iter.Lf, -1.0), something(iter.gamma, -1.0),
                          iter.minimum_gamma, iter.reduce_gamma, iter.increase_gamma, fast ? iter.mf : 0.0,
                          skind, sp0, sp1, kind, p0, p1, 1, 1))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pg_iter_create, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{PgIterOpts}, Ref{Ptr{Cvoid}}),
                iter.f.A.ctx.handle, iter.f.handle, opts, h))
    set_g_vectors!(h[], iter.g)
    sc = Ref{PgIterScalars}()
    check(ccall((:pg_iter_state_upload, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{PgIterScalars}), h[], blob, length(blob), sc))
    st = HIPIterState{R,T}(h[], x0, R(0), x0, R(0), x0, x0, R(0), x0, nothing, (skind, sp0, sp1), R(0), iter.g)
    finalizer(s -> ccall((:pg_iter_destroy, libpg), Int32, (Ptr{Cvoid},), s.handle), st)
    refresh!(st, sc[], iter.f.A.ctx, length(x0))
    return st
end

# LBFGSOperator on the device (src/accel/lbfgs.jl:5-95) with the image slab: `ProximalAlgorithms.initialize(LBFGS(M), x)` for a
# HIPVector x gives this operator; update! / reset! / mul! are the reference's methods.  With images enabled (enable_images!)
# the image A (H v) of a direction follows from A v without reading A (images_mul!), which removes the
# `mul!(state.Ad, iter.A, state.d)` of panoc.jl:180.
mutable struct HIPLBFGSOperator{T}
    handle::Ptr{Cvoid}
    ctx::HIPContext
    n::Int
end
function HIPLBFGSOperator(M::Integer, x::HIPVector{T}) where {T}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pg_lbfgs_create, libpg), Int32, (Ptr{Cvoid}, Int32, Int32, Int64, Ref{Ptr{Cvoid}}), x.ctx.handle, pg_dtype(T), M, x.n, h))
    L = HIPLBFGSOperator{T}(h[], x.ctx, x.n)
    finalizer(l -> ccall((:pg_lbfgs_destroy, libpg), Int32, (Ptr{Cvoid},), l.handle), L)
end
ProximalAlgorithms.initialize(::ProximalAlgorithms.LBFGS{M}, x::HIPVector) where {M} = HIPLBFGSOperator(M, x)   # lbfgs.jl:103-105
ProximalAlgorithms.update!(L::HIPLBFGSOperator, s::HIPVector, y::HIPVector) =   # lbfgs.jl:30-50
    (check(ccall((:pg_lbfgs_update, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), L.handle, s.ptr, y.ptr)); L)
ProximalAlgorithms.reset!(L::HIPLBFGSOperator) = (check(ccall((:pg_lbfgs_reset, libpg), Int32, (Ptr{Cvoid},), L.handle)); L)   # :52-55
LinearAlgebra.mul!(d::HIPVector{T}, L::HIPLBFGSOperator{T}, v::HIPVector{T}) where {T} =   # :64-95
    (check(ccall((:pg_lbfgs_apply, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), L.handle, d.ptr, v.ptr)); d)
enable_images!(L::HIPLBFGSOperator, m::Integer) = (check(ccall((:pg_lbfgs_images_enable, libpg), Int32, (Ptr{Cvoid}, Int64), L.handle, m)); L)
images_update!(L::HIPLBFGSOperator, As::HIPVector, Ay::HIPVector) =   # right after update!(L, s, y)
    (check(ccall((:pg_lbfgs_images_update, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), L.handle, As.ptr, Ay.ptr)); L)
images_mul!(Ad::HIPVector, L::HIPLBFGSOperator, Av::HIPVector) =       # right after mul!(d, L, v): Ad = A d from A v
    (check(ccall((:pg_lbfgs_images_apply, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), L.handle, Ad.ptr, Av.ptr)); Ad)
function images_ready(L::HIPLBFGSOperator)
    r = Ref{Int32}(0)
    check(ccall((:pg_lbfgs_images_ready, libpg), Int32, (Ptr{Cvoid}, Ref{Int32}), L.handle, r))
    r[] != 0
end

# Row teams (INTEGRATION.md section 3): north_star's row layout at one read of A per iteration.  Each process allocates its
# inbox, ships the 64-byte IPC handle to its peers (MPI.jl / Distributed), opens theirs and registers the list.
function row_team_inbox(ctx::HIPContext)
    p = Ref{Ptr{Cvoid}}(C_NULL); nb = Ref{Int64}(0)
    check(ccall((:pg_ctx_row_team_alloc, libpg), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int64}), ctx.handle, p, nb))
    p[]
end
function row_team_handle(ctx::HIPContext)
    h = Vector{UInt8}(undef, 64)
    check(ccall((:pg_ctx_row_team_export, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.handle, h))
    h
end
function row_team_open(ctx::HIPContext, handle::Vector{UInt8})
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pg_ctx_row_team_import, libpg), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx.handle, handle, p))
    p[]
end
function row_team_selftest(ctx::HIPContext)   # every rank, right after set_row_team! and a barrier: must return N (N + 1) / 2
    s = Ref{Float64}(0.0)
    check(ccall((:pg_ctx_row_team_selftest, libpg), Int32, (Ptr{Cvoid}, Ref{Float64}), ctx.handle, s))
    s[]
end
function row_team_stats(ctx::HIPContext)   # (sweeps, late wave-steps, polls spent waiting) since set_row_team!
    a = Ref{Int64}(0); b = Ref{Int64}(0); c = Ref{Int64}(0)
    check(ccall((:pg_ctx_row_team_stats, libpg), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}), ctx.handle, a, b, c))
    (sweeps = a[], late_waves = b[], wait_polls = c[])
end
# the sweep's geometry at run time (every rank the same values): row_team_tune!(ctx, "PAIR", 1) = one post per two steps
row_team_tune!(ctx::HIPContext, key::AbstractString, value::Integer) =
    check(ccall((:pg_ctx_row_team_tune, libpg), Int32, (Ptr{Cvoid}, Cstring, Int64), ctx.handle, key, value))
function row_team_geometry(ctx::HIPContext)   # what the last row-team sweep ran with ("W=1 U=8 C=2 ... PAIR=0 ...")
    buf = zeros(UInt8, 160)
    check(ccall((:pg_ctx_row_team_geometry, libpg), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int64), ctx.handle, buf, length(buf)))
    unsafe_string(pointer(buf))
end
set_row_team!(ctx::HIPContext, rank::Integer, inboxes::Vector{Ptr{Cvoid}}; max_workgroups::Integer = 0) =
    check(ccall((:pg_ctx_set_row_team, libpg), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Ptr{Cvoid}}, Int32),
                ctx.handle, length(inboxes), rank, inboxes, max_workgroups))

# DouglasRachford on a separable quadratic + box / L1 (douglas_rachford.jl:53-70): the whole loop in the library,
# 16 iterations per HBM sweep (pg_dr_run); d, q: scalars or HIPVectors; g: HIPIndBox / HIPNormL1.
function hip_douglas_rachford(d, q, g, x0::Vector{T}; gamma, maxit = 1_000, tol = 1e-8, ctx = default_ctx()) where {T}
    x, x_alt, y = HIPVector(x0; ctx), HIPVector{T}(undef, length(x0); ctx), HIPVector{T}(undef, length(x0); ctx)
    vecptr(v) = v isa HIPVector ? v.ptr : C_NULL
    scal(v) = v isa HIPVector ? 0.0 : Float64(v)
    kind, p0, p1 = g_spec(g)
    k = Ref{Int64}(0)
    sc = zeros(Float64, 3)
    check(ccall((:pg_dr_run, libpg), Int32,
                (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64,
                 Ptr{Cvoid}, Float64, Int32, Float64, Float64, Float64, Float64, Int64, Int32, Ref{Int64}, Ptr{Float64}),
                ctx.handle, pg_dtype(T), length(x0), x.ptr, x_alt.ptr, y.ptr, C_NULL, C_NULL, C_NULL, vecptr(d), scal(d),
                vecptr(q), scal(q), kind, p0, p1, Float64(gamma), Float64(tol), maxit, 16, k, sc))
    return Array(y), Int(k[])                              # solution = state.y (douglas_rachford.jl:70)
end

export HIPStyle, BROADCAST_TABLE, HIPContext, HIPVector, HIPMatrix, HIPLeastSquares, HIPNormL1, HIPIndBox,
       HIPForwardBackwardIteration, HIPFastForwardBackwardIteration, HIPForwardBackward, HIPFastForwardBackward, hip_solve,
       hip_douglas_rachford, save_state, resume, HIPLBFGSOperator, enable_images!, images_update!, images_mul!, images_ready,
       row_team_inbox, row_team_handle, row_team_open, set_row_team!, row_team_stats, row_team_selftest, row_team_tune!, row_team_geometry,
       fused_tn!, fused_dys!

end # module
