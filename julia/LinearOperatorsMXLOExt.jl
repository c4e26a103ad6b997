# LinearOperatorsMXLOExt — the reference-side binding of libmxlo.so (include/mxlo.h).
#
# STATUS: NOT EXECUTED. The build image has no Julia; this file is the glue a maintainer would drop
# next to ext/LinearOperatorsAMDGPUExt.jl. Every `ccall` below repeats, argument for argument, a call
# that IS executed and tested from Python/ctypes (linearoperators.jl_amd/{leaves,qn,sharded}.py) and
# from plain C (tests/abi_client.c). Signatures are those of include/mxlo.h.
#
# Design: the reference's storage type `S` is a KEYWORD of its constructors (src/special-operators.jl:53,95,118,187,249)
# and keywords do not take part in dispatch, so the extension hooks in one level lower, where the reference
# dispatches on the VECTORS: it adds methods on the kernel functions the reference's closures call —
# `mulOpEye!`, `mulOpOnes!`, `mulOpZeros!`, `mulSquareOpDiagonal!`, `mulOpDiagonal!`, `mulHouseholder!`,
# `mulRestrict!`, `multRestrict!` — for `MXVector` arguments. Every existing call site
# (`opEye(T, n; S = MXVector{T})`, `opDiagonal(d)`, `opRestriction(I, n; S = ...)`, and the internal
# `opOnes(op.nrow, op.ncol)` of `op + x`, src/operations.jl:222-223) then reaches the device unchanged: whatever `S`
# says, the closure is handed device vectors and dispatch does the rest. Constructors get their own methods only
# where the reference's closure would allocate or copy per call (`conj.(d)` in opDiagonal's ctprod!,
# src/special-operators.jl:139-141; `tril(A,-1)` in opHermitian; kron's `convert(Vector{S}, x)`, src/kron.jl:16)
# or hard-codes `Vector{T}` (LBFGSData / LSR1Data, src/lbfgs.jl:26-57 — those need the storage type as a THIRD
# POSITIONAL argument: `LBFGSOperator(T, n, MXVector{T}; mem = 5)`; see INTEGRATION.md).
# Each body is one `ccall`; flags, counters, `mul!` dispatch, adjoint/transpose wrappers, combinators, cat,
# `Matrix(op)` stay the reference's code.
module LinearOperatorsMXLOExt

using LinearOperators, LinearAlgebra
import LinearOperators: storage_type, LinearOperator, LinearOperatorException, AbstractQuasiNewtonOperator,
  opDiagonal, opHouseholder, opHermitian, opRestriction, opEye, opOnes, opZeros, BlockDiagonalOperator,
  LBFGSOperator, InverseLBFGSOperator, LSR1Operator, reset!, diag!, solve_shifted_system!, has_args5,
  isallocated5, mulOpEye!, mulOpOnes!, mulOpZeros!, mulSquareOpDiagonal!, mulOpDiagonal!, mulHouseholder!,
  mulRestrict!, multRestrict!
import Base: kron, push!, size, getindex, view, fill!, copyto!, similar, length, unsafe_convert
import LinearAlgebra: ldiv!

const lib = "libmxlo"            # linearoperators.jl_amd/csrc/libmxlo.so on LD_LIBRARY_PATH
const rccl = "libmxlo_rccl"      # optional: native RCCL transport of the all-reduce hook

# ---- status codes -> the exception types the reference throws -----------------------------------
lasterr() = unsafe_string(ccall((:mxlo_last_error, lib), Cstring, ()))
@inline function check(st::Int32)
  st == 0 && return nothing
  st == 2 && throw(LinearOperatorException(lasterr()))     # MXLO_ESHAPE  (operations.jl:23-24)
  st == 6 && throw(ArgumentError(lasterr()))               # MXLO_EDOMAIN (utilities.jl:213-215)
  error(lasterr())                                         # ErrorException (lbfgs.jl:295-299, HIP errors)
end

# ---- context ---------------------------------------------------------------------------------------
mutable struct Ctx
  h::Ptr{Cvoid}
end
function Ctx(device::Integer = 0; stream::Ptr{Cvoid} = C_NULL)
  r = Ref{Ptr{Cvoid}}()
  check(ccall((:mxlo_ctx_create, lib), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), device, stream, r))
  finalizer(c -> ccall((:mxlo_ctx_destroy, lib), Int32, (Ptr{Cvoid},), c.h), Ctx(r[]))
end
const CTX = Ref{Ctx}()
ctx() = (isassigned(CTX) || (CTX[] = Ctx(parse(Int, get(ENV, "LOCAL_RANK", "0")))); CTX[].h)
synchronize() = check(ccall((:mxlo_ctx_sync, lib), Int32, (Ptr{Cvoid},), ctx()))

# ---- device vector / matrix types --------------------------------------------------------------------
"Dense device vector: plays `Vector{T}` for storage_type dispatch. Contiguous views are pointer+offset."
mutable struct MXVector{T} <: AbstractVector{T}
  ptr::Ptr{T}
  len::Int
  owner::Any            # nothing => owns the allocation; otherwise the parent kept alive by a view
end
function MXVector{T}(::UndefInitializer, n::Integer) where {T}
  r = Ref{Ptr{Cvoid}}()
  check(ccall((:mxlo_malloc, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}), ctx(), max(n, 1) * sizeof(T), r))
  v = MXVector{T}(Ptr{T}(r[]), n, nothing)
  finalizer(x -> ccall((:mxlo_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), x.ptr), v)
end
MXVector(h::Vector{T}) where {T} = copyto!(MXVector{T}(undef, length(h)), h)
size(v::MXVector) = (v.len,)
length(v::MXVector) = v.len
similar(v::MXVector{T}) where {T} = MXVector{T}(undef, v.len)
similar(v::MXVector, ::Type{T}, n::Integer) where {T} = MXVector{T}(undef, n)
unsafe_convert(::Type{Ptr{Cvoid}}, v::MXVector) = Ptr{Cvoid}(v.ptr)
getindex(v::MXVector, i::Integer) = error("scalar indexing of a device vector; use Array(v)")
getindex(v::MXVector, r::UnitRange{<:Integer}) = copyto!(similar(v, eltype(v), length(r)), view(v, r))   # d[1:nrow] (special-operators.jl:159)
Base.isreal(::MXVector{<:Real}) = true                       # opDiagonal's hermitian flag (special-operators.jl:141)
Base.isreal(::MXVector{<:Complex}) = false                   # (a complex device vector is not scanned for zero imaginary parts)
# contiguous views (cat.jl:17-18, special-operators.jl:263) stay device vectors: base pointer + offset
view(v::MXVector{T}, r::UnitRange{<:Integer}) where {T} =
  MXVector{T}(v.ptr + (first(r) - 1) * sizeof(T), length(r), v)
function copyto!(d::MXVector{T}, s::Vector{T}) where {T}
  GC.@preserve s check(ccall((:mxlo_memcpy_h2d, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                             ctx(), d.ptr, pointer(s), d.len * sizeof(T)))
  d
end
function copyto!(d::MXVector{T}, s::MXVector{T}) where {T}
  check(ccall((:mxlo_memcpy_d2d, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
              ctx(), d.ptr, s.ptr, d.len * sizeof(T)))
  d
end
function Base.Array(s::MXVector{T}) where {T}
  h = Vector{T}(undef, s.len)
  GC.@preserve h check(ccall((:mxlo_memcpy_d2h, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                             ctx(), pointer(h), s.ptr, s.len * sizeof(T)))   # synchronises the ctx stream
  h
end

"Dense column-major device matrix, leading dimension = m."
struct MXMatrix{T} <: AbstractMatrix{T}
  data::MXVector{T}
  m::Int
  n::Int
end
MXMatrix(A::Matrix{T}) where {T} = MXMatrix{T}(MXVector(vec(A)), size(A)...)
size(A::MXMatrix) = (A.m, A.n)
storage_type(::MXMatrix{T}) where {T} = MXVector{T}        # cf. ext/LinearOperatorsAMDGPUExt.jl:6

dt(::Type{Float64}) = Int32(0)      # MXLO_F64
dt(::Type{Float32}) = Int32(1)      # MXLO_F32
dt(::Type{ComplexF64}) = Int32(2)   # MXLO_C64 (the `_c` entry points)
dt(::Type{ComplexF32}) = Int32(3)   # MXLO_C32
const RealT = Union{Float64, Float32}
const CplxT = Union{ComplexF64, ComplexF32}
# Julia does not convert caller scalars to T (SURVEY §8a "Mixed precision"): next to Float32 (ComplexF32) data the
# α-term is evaluated in promote_type(typeof(α), T) and the β-term in promote_type(typeof(β), T), each on its own.
@inline is64(x) = x isa Float64 || x isa ComplexF64
@inline wflags(α, β) = (is64(α) ? Int32(1) : Int32(0)) | (is64(β) ? Int32(8) : Int32(0))      # MXLO_ALPHA_F64 | MXLO_BETA_F64
@inline flags(::Type{Float32}, α, β) = wflags(α, β)
@inline flags(::Type{Float64}, α, β) = Int32(0)
# complex data: Real scalars multiply componentwise (MXLO_ALPHA_REAL 0x20 / MXLO_BETA_REAL 0x40)
@inline rflags(α, β) = (α isa Real ? Int32(0x20) : Int32(0)) | (β isa Real ? Int32(0x40) : Int32(0))
@inline flags(::Type{ComplexF64}, α, β) = rflags(α, β)
@inline flags(::Type{ComplexF32}, α, β) = rflags(α, β) | wflags(α, β)
@inline rpart(x) = Float64(real(x))      # (re, im) pairs of the `_c` entry points; not named re/im: `im` is Base's imaginary unit
@inline ipart(x) = Float64(imag(x))
const P = Ptr{Cvoid}

# ---- prod3! glue on MXVector (src/operations.jl:10-20) ----------------------------------------------------
# `res .*= α`                       -> mxlo_scale
# `res .= α .* Mv .+ β .* res`      -> mxlo_eye_mul without MXLO_TAIL_BETA (generic axpby)
scale!(res::MXVector{T}, α) where {T <: RealT} =
  check(ccall((:mxlo_scale, lib), Int32, (P, Int32, P, Int64, Float64, Int32), ctx(), dt(T), res.ptr, res.len, α,
              flags(T, α, α)))
scale!(res::MXVector{T}, α) where {T <: CplxT} =
  check(ccall((:mxlo_scale_c, lib), Int32, (P, Int32, P, Int64, Float64, Float64, Int32), ctx(), dt(T), res.ptr, res.len,
              rpart(α), ipart(α), flags(T, α, α)))
axpby!(res::MXVector{T}, v::MXVector{T}, α, β) where {T <: RealT} =
  check(ccall((:mxlo_eye_mul, lib), Int32, (P, Int32, P, P, Int64, Int64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, v.ptr, res.len, res.len, α, β, flags(T, α, β)))
axpby!(res::MXVector{T}, v::MXVector{T}, α, β) where {T <: CplxT} =
  check(ccall((:mxlo_eye_mul_c, lib), Int32, (P, Int32, P, P, Int64, Int64, Float64, Float64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, v.ptr, res.len, res.len, rpart(α), ipart(α), rpart(β), ipart(β), flags(T, α, β)))
# conj!(res) / conj.(v) of the wrapper routing (src/adjtrans.jl:127-136,193-204,226-249) on device vectors
function Base.conj!(v::MXVector{T}) where {T <: CplxT}
  check(ccall((:mxlo_conj_c, lib), Int32, (P, Int32, P, P, Int64), ctx(), dt(T), v.ptr, v.ptr, v.len))
  v
end
Base.conj!(v::MXVector{<:Real}) = v
function Base.conj(v::MXVector{T}) where {T <: CplxT}
  out = similar(v)
  check(ccall((:mxlo_conj_c, lib), Int32, (P, Int32, P, P, Int64), ctx(), dt(T), out.ptr, v.ptr, v.len))
  out
end
Base.Broadcast.broadcasted(::typeof(conj), v::MXVector{<:CplxT}) = conj(v)      # `conj.(v)` (adjtrans.jl:128)
Base.Broadcast.broadcasted(::typeof(conj), v::MXVector{<:Real}) = v             # `conj.(d)` of the generic opDiagonal ctprod!

# ---- BLAS-1 on device vectors: what Krylov.jl / JSOSolvers call between two mul! ------------------------------------
# dot / norm need the scalar on the host: the fixed-order device reduction (mxlo_dot, all-reduce hook included, so
# the result is the GLOBAL dot of row-sharded vectors) writes one device double that is read back (8-byte D2H).
const DOTBUF = Ref{Ptr{Cvoid}}(C_NULL)
function dotbuf()
  if DOTBUF[] == C_NULL
    r = Ref{Ptr{Cvoid}}()
    check(ccall((:mxlo_malloc, lib), Int32, (P, Int64, Ptr{P}), ctx(), 16, r))
    DOTBUF[] = r[]
  end
  DOTBUF[]
end
function LinearAlgebra.dot(a::MXVector{T}, b::MXVector{T}) where {T <: CplxT}
  length(a) == length(b) || throw(DimensionMismatch("dot"))
  check(ccall((:mxlo_dot_c, lib), Int32, (P, Int32, P, P, Int64, P), ctx(), dt(T), a.ptr, b.ptr, length(a), dotbuf()))
  out = zeros(Float64, 2)
  check(ccall((:mxlo_memcpy_d2h, lib), Int32, (P, Ptr{Float64}, P, Int64), ctx(), out, dotbuf(), 16))
  T(out[1], out[2])
end
function LinearAlgebra.dot(a::MXVector{T}, b::MXVector{T}) where {T <: RealT}
  length(a) == length(b) || throw(DimensionMismatch("dot"))
  check(ccall((:mxlo_dot, lib), Int32, (P, Int32, P, P, Int64, P), ctx(), dt(T), a.ptr, b.ptr, length(a), dotbuf()))
  out = Ref{Float64}(0.0)
  check(ccall((:mxlo_memcpy_d2h, lib), Int32, (P, Ptr{Float64}, P, Int64), ctx(), out, dotbuf(), 8))
  T(out[])
end
LinearAlgebra.norm(a::MXVector) = sqrt(dot(a, a))
LinearAlgebra.axpy!(α::Number, x::MXVector{T}, y::MXVector{T}) where {T} = (axpby!(y, x, T(α), one(T)); y)       # y += αx
LinearAlgebra.axpby!(α::Number, x::MXVector{T}, β::Number, y::MXVector{T}) where {T} = (axpby!(y, x, T(α), T(β)); y)
LinearAlgebra.rmul!(x::MXVector{T}, α::Number) where {T} = (scale!(x, T(α)); x)
LinearAlgebra.lmul!(α::Number, x::MXVector{T}) where {T} = (scale!(x, T(α)); x)
Base.fill!(v::MXVector{T}, x::Number) where {T <: RealT} =
  (check(ccall((:mxlo_fill, lib), Int32, (P, Int32, P, Int64, Float64), ctx(), dt(T), v.ptr, v.len, Float64(x))); v)

# ---- kernel-function methods: what the reference's OWN closures dispatch to when handed device vectors --------------
# mulSquareOpDiagonal! / mulOpDiagonal! (src/special-operators.jl:125-131,144-151)
mulSquareOpDiagonal!(res::MXVector{T}, d::MXVector{T}, v::MXVector{T}, α, β) where {T <: RealT} =
  check(ccall((:mxlo_diag_mul, lib), Int32, (P, Int32, P, P, P, Int64, Int64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, d.ptr, v.ptr, length(res), length(res), α, β,
              flags(T, α, β) | (length(d) == 1 && length(res) != 1 ? Int32(2) : Int32(0))))   # MXLO_D_SCALAR
mulOpDiagonal!(res::MXVector{T}, d::MXVector{T}, v::MXVector{T}, α, β, n_min) where {T <: RealT} =
  check(ccall((:mxlo_diag_mul, lib), Int32, (P, Int32, P, P, P, Int64, Int64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, d.ptr, v.ptr, n_min, length(res), α, β, flags(T, α, β)))
cdiag(res::MXVector{T}, d::MXVector{T}, v::MXVector{T}, α, β, n_min, conjd::Bool) where {T <: CplxT} =
  check(ccall((:mxlo_diag_mul_c, lib), Int32, (P, Int32, P, P, P, Int64, Int64, Float64, Float64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, d.ptr, v.ptr, n_min, length(res), rpart(α), ipart(α), rpart(β), ipart(β),
              flags(T, α, β) | (conjd ? Int32(0x10) : Int32(0))))                              # MXLO_CONJ_D
mulSquareOpDiagonal!(res::MXVector{T}, d::MXVector{T}, v::MXVector{T}, α, β) where {T <: CplxT} =
  cdiag(res, d, v, α, β, length(res), false)
mulOpDiagonal!(res::MXVector{T}, d::MXVector{T}, v::MXVector{T}, α, β, n_min) where {T <: CplxT} =
  cdiag(res, d, v, α, β, n_min, false)
# mulOpEye! (src/special-operators.jl:36-44): the tail receives β itself (MXLO_TAIL_BETA)
mulOpEye!(res::MXVector{T}, v::MXVector{T}, α, β, n_min) where {T <: RealT} =
  check(ccall((:mxlo_eye_mul, lib), Int32, (P, Int32, P, P, Int64, Int64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, v.ptr, n_min, length(res), α, β, flags(T, α, β) | Int32(4)))
mulOpEye!(res::MXVector{T}, v::MXVector{T}, α, β, n_min) where {T <: CplxT} =
  check(ccall((:mxlo_eye_mul_c, lib), Int32, (P, Int32, P, P, Int64, Int64, Float64, Float64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, v.ptr, n_min, length(res), rpart(α), ipart(α), rpart(β), ipart(β), flags(T, α, β) | Int32(4)))
# mulOpOnes! (src/special-operators.jl:79-85) — real element types (fixed-order device sum, all-reduce hook included)
mulOpOnes!(res::MXVector{T}, v::MXVector{T}, α, β) where {T <: RealT} =
  check(ccall((:mxlo_ones_mul, lib), Int32, (P, Int32, P, Int64, P, Int64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, length(res), v.ptr, length(v), α, β, flags(T, α, β)))
# mulOpZeros! (src/special-operators.jl:102-108)
mulOpZeros!(res::MXVector{T}, v::MXVector, α, β) where {T <: RealT} =
  check(ccall((:mxlo_zeros_mul, lib), Int32, (P, Int32, P, Int64, Float64, Int32), ctx(), dt(T), res.ptr, length(res), β,
              flags(T, α, β)))
mulOpZeros!(res::MXVector{T}, v::MXVector, α, β) where {T <: CplxT} =
  check(ccall((:mxlo_zeros_mul_c, lib), Int32, (P, Int32, P, Int64, Float64, Float64, Int32), ctx(), dt(T), res.ptr,
              length(res), rpart(β), ipart(β), flags(T, α, β)))
# mulHouseholder! (src/linalg.jl:77-83); complex h: LinearAlgebra.dot conjugates it
mulHouseholder!(res::MXVector{T}, h::MXVector{T}, v::MXVector{T}, α, β) where {T <: RealT} =
  check(ccall((:mxlo_householder_mul, lib), Int32, (P, Int32, P, P, P, Int64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, h.ptr, v.ptr, length(res), α, β, flags(T, α, β)))
mulHouseholder!(res::MXVector{T}, h::MXVector{T}, v::MXVector{T}, α, β) where {T <: CplxT} =
  check(ccall((:mxlo_householder_mul_c, lib), Int32, (P, Int32, P, P, P, Int64, Float64, Float64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, h.ptr, v.ptr, length(res), rpart(α), ipart(α), rpart(β), ipart(β), flags(T, α, β)))

# ---- a3/a4 opDiagonal (src/special-operators.jl:133-165): own constructor only so that ctprod! does not allocate
# `conj.(d)` on every call (:139-141) — MXLO_CONJ_D conjugates d inside the kernel.
function opDiagonal(d::MXVector{T}) where {T <: CplxT}
  n = length(d)
  prod! = (res, v, α, β) -> cdiag(res, d, v, α, β, n, false)
  ctprod! = (res, w, α, β) -> cdiag(res, d, w, α, β, n, true)
  LinearOperator{T, MXVector{T}}(n, n, true, false, prod!, prod!, ctprod!)          # hermitian = isreal(d) (:142)
end
function opDiagonal(nrow::I, ncol::I, d::MXVector{T}) where {T <: CplxT, I <: Integer}
  nrow == ncol <= length(d) && return opDiagonal(view(d, 1:nrow))
  n_min = min(nrow, ncol)
  prod! = (res, v, α, β) -> cdiag(res, d, v, α, β, n_min, false)
  ctprod! = (res, w, α, β) -> cdiag(res, d, w, α, β, n_min, true)
  LinearOperator{T, MXVector{T}}(nrow, ncol, false, false, prod!, prod!, ctprod!)
end
# real d: the reference's generic constructors (opDiagonal(d), opDiagonal(nrow, ncol, d), opEye/opOnes/opZeros with
# S = MXVector{T}, opHouseholder(h)) work as they are through the kernel-function methods above; opHouseholder's
# hard-coded S = Vector{T} (src/linalg.jl:94) only matters for temporaries of compose, so give it the right S:
function opHouseholder(h::MXVector{T}) where {T}
  n = length(h)
  prod! = (res, v, α, β) -> mulHouseholder!(res, h, v, α, β)
  LinearOperator{T, MXVector{T}}(n, n, T <: Real, true, prod!, nothing, prod!)        # symmetric = isreal(h)
end
# `op + x` / `x + op` build `x * opOnes(op.nrow, op.ncol)` with T = Float64, S = Vector{Float64}
# (src/operations.jl:222-223): for device operators the ones-operator must carry the device storage type.
# The ones-operator takes the REAL component type of T (mulOpOnes! is a real kernel; next to a complex operator the sum
# promotes as Real + Complex does and reaches the real operator through the two planes of the vector).
ones_like(op::LinearOperator{T, MXVector{T}}) where {T} = opOnes(real(T), op.nrow, op.ncol; S = MXVector{real(T)})
Base.:+(op::LinearOperator{T, MXVector{T}}, x::Number) where {T} = op + x * ones_like(op)
Base.:+(x::Number, op::LinearOperator{T, MXVector{T}}) where {T} = x * ones_like(op) + op

# ---- a6 opHermitian (src/linalg.jl:97-127): the ORIGINAL matrix is passed; only tril(A,-1) is read ------------
# A callable rather than a closure, so that `mul!` on MATRICES can dispatch to the block entry point: the strict lower triangle
# is read once per 4 columns and every column gets the bits of the single apply. This is an EXTENSION of the reference, not a
# match: its closure ends in `(...)[:]` (src/linalg.jl:99-101), so `mul!(res::Matrix, opHermitian(d, A), V)` with more than one
# column throws DimensionMismatch upstream. Complex data goes column by column through the generic apply_columns below.
struct HermApply{T}
  d::MXVector{T}
  A::MXMatrix{T}
end
(f::HermApply{T})(res::MXVector{T}, v::MXVector{T}, α, β) where {T <: RealT} = check(ccall((:mxlo_hermitian_mul, lib), Int32,
    (P, Int32, P, P, P, Int64, P, Int64, Float64, Float64, Int32),
    ctx(), dt(T), res.ptr, f.d.ptr, f.A.data.ptr, f.A.m, v.ptr, f.A.n, α, β, flags(T, α, β)))
(f::HermApply{T})(res::MXMatrix{T}, V::MXMatrix{T}, α, β) where {T <: RealT} = check(ccall((:mxlo_hermitian_mul_block, lib), Int32,
    (P, Int32, P, Int64, P, P, Int64, P, Int64, Int64, Int64, Float64, Float64, Int32),
    ctx(), dt(T), res.data.ptr, res.m, f.d.ptr, f.A.data.ptr, f.A.m, V.data.ptr, V.m, f.A.n, size(V, 2), α, β, flags(T, α, β)))
function apply_columns(f::HermApply{T}, res::MXMatrix{T}, m::MXMatrix{T}, α, β) where {T <: RealT}     # one call for the block
  size(res, 2) == size(m, 2) || throw(LinearOperatorException("shape mismatch"))
  f(res, m, α, β)
  res
end
function opHermitian(d::MXVector{T}, A::MXMatrix{T}) where {T <: RealT}
  m, n = size(A)
  m == n == length(d) || throw(LinearOperatorException("shape mismatch"))
  LinearOperator{T, MXVector{T}}(m, m, true, true, HermApply{T}(d, A), nothing, nothing)
end

# complex A (test/test_linop.jl:360-370: ComplexF64 A, d = real.(diag(A))): L' is the conjugate transpose, symmetric =
# isreal(A) = false; a Real diagonal is passed as such (MXLO_D_REAL = 0x80: d .* v is Real * Complex)
function opHermitian(d::MXVector{S}, A::MXMatrix{T}) where {S <: Union{RealT, CplxT}, T <: CplxT}
  m, n = size(A)
  m == n == length(d) || throw(LinearOperatorException("shape mismatch"))
  (S === T || S === real(T)) || throw(ArgumentError("opHermitian: convert d to $(T) or $(real(T)) first"))
  dflag = S <: Real ? Int32(0x80) : Int32(0)
  prod! = (res, v, α, β) -> check(ccall((:mxlo_hermitian_mul_c, lib), Int32,
      (P, Int32, P, P, P, Int64, P, Int64, Float64, Float64, Float64, Float64, Int32),
      ctx(), dt(T), res.ptr, d.ptr, A.data.ptr, m, v.ptr, n, rpart(α), ipart(α), rpart(β), ipart(β), flags(T, α, β) | dflag))
  LinearOperator{T, MXVector{T}}(m, m, false, true, prod!, nothing, nothing)
end

# ---- dense LinearOperator(M) (src/constructors.jl:19-29) ----------------------------------------------------
function LinearOperator(M::MXMatrix{T}; symmetric = false, hermitian = false, S = MXVector{T}) where {T <: CplxT}
  m, n = size(M)
  gemv(mode) = (res, v, α, β) -> check(ccall((:mxlo_gemv_c, lib), Int32,
      (P, Int32, P, P, Int64, Int64, Int64, P, Float64, Float64, Float64, Float64, Int32, Int32),
      ctx(), dt(T), res.ptr, M.data.ptr, m, n, m, v.ptr, rpart(α), ipart(α), rpart(β), ipart(β), Int32(mode), flags(T, α, β)))
  LinearOperator{T, S}(m, n, symmetric, hermitian, gemv(0), gemv(1), gemv(2))     # M*v, transpose(M)*u, M'*w
end
# The closure of a real dense operator is a callable with a vector method (GEMV) and a matrix method (the block GEMV:
# `mul!(res::Matrix, op, V::Matrix, α, β)`, src/operations.jl:34-36, reads M once per 8 columns of V).
struct DenseApply{T}
  M::MXMatrix{T}
  mode::Int32          # MXLO_OP_N / _T / _C
end
(f::DenseApply{T})(res::MXVector{T}, v::MXVector{T}, α, β) where {T <: RealT} = check(ccall((:mxlo_gemv, lib), Int32,
    (P, Int32, P, P, Int64, Int64, Int64, P, Float64, Float64, Int32, Int32),
    ctx(), dt(T), res.ptr, f.M.data.ptr, f.M.m, f.M.n, f.M.m, v.ptr, α, β, f.mode, flags(T, α, β)))
(f::DenseApply{T})(res::MXMatrix{T}, V::MXMatrix{T}, α, β) where {T <: RealT} = check(ccall((:mxlo_gemv_block, lib), Int32,
    (P, Int32, P, Int64, P, Int64, Int64, Int64, P, Int64, Int64, Float64, Float64, Int32, Int32),
    ctx(), dt(T), res.data.ptr, res.m, f.M.data.ptr, f.M.m, f.M.n, f.M.m, V.data.ptr, V.m, size(V, 2), α, β, f.mode,
    flags(T, α, β)))
function LinearOperator(M::MXMatrix{T}; symmetric = false, hermitian = false, S = MXVector{T}) where {T <: RealT}
  m, n = size(M)     # S: the keyword of src/constructors.jl:15 (test/gpu/test_S_kwarg.jl:19 passes it)
  LinearOperator{T, S}(m, n, symmetric, hermitian, DenseApply{T}(M, Int32(0)), DenseApply{T}(M, Int32(1)),
                       DenseApply{T}(M, Int32(2)))
end

# ---- 5-arg mul! of a plain device MATRIX (and its lazy transpose / adjoint) on device vectors --------------------------
# The reference's generic `BlockDiagonalOperator(A, B, C)` accepts plain matrices next to operators
# (src/abstract.jl:173 `has_args5(::AbstractMatrix) = true`, src/special-operators.jl:258-289: per block
# `mul!(view(y), op, view(x), α, β)`, `transpose(op)`, `adjoint(op)`); test/gpu/amdgpu.jl:4-12 builds exactly that from
# three ROCArray matrices. With these methods the same call works on MXMatrix blocks (one GEMV per block; the fused
# one-launch form is `BlockDiagonalOperator(T, blocks...)` below).
const LazyT{T} = Union{Transpose{T, MXMatrix{T}}, Adjoint{T, MXMatrix{T}}}
storage_type(::LazyT{T}) where {T} = MXVector{T}            # src/abstract.jl:181-182 forwards to the parent; explicit here
gemv_mode(::MXMatrix) = Int32(0)
gemv_mode(::Transpose) = Int32(1)
gemv_mode(::Adjoint) = Int32(2)
function LinearAlgebra.mul!(res::MXVector{T}, M::Union{MXMatrix{T}, LazyT{T}}, v::MXVector{T}, α::Number, β::Number) where {T <: RealT}
  A = M isa MXMatrix ? M : parent(M)
  (length(v) == size(M, 2) && length(res) == size(M, 1)) ||
    throw(DimensionMismatch("mul!: res has $(length(res)) rows, M is $(size(M)), v has $(length(v))"))
  check(ccall((:mxlo_gemv, lib), Int32, (P, Int32, P, P, Int64, Int64, Int64, P, Float64, Float64, Int32, Int32),
              ctx(), dt(T), res.ptr, A.data.ptr, A.m, A.n, A.m, v.ptr, α, β, gemv_mode(M), flags(T, α, β)))
  res
end
function LinearAlgebra.mul!(res::MXVector{T}, M::Union{MXMatrix{T}, LazyT{T}}, v::MXVector{T}, α::Number, β::Number) where {T <: CplxT}
  A = M isa MXMatrix ? M : parent(M)
  (length(v) == size(M, 2) && length(res) == size(M, 1)) ||
    throw(DimensionMismatch("mul!: res has $(length(res)) rows, M is $(size(M)), v has $(length(v))"))
  check(ccall((:mxlo_gemv_c, lib), Int32,
              (P, Int32, P, P, Int64, Int64, Int64, P, Float64, Float64, Float64, Float64, Int32, Int32),
              ctx(), dt(T), res.ptr, A.data.ptr, A.m, A.n, A.m, v.ptr, rpart(α), ipart(α), rpart(β), ipart(β), gemv_mode(M),
              flags(T, α, β)))
  res
end
LinearAlgebra.mul!(res::MXVector{T}, M::Union{MXMatrix{T}, LazyT{T}}, v::MXVector{T}) where {T} = mul!(res, M, v, one(T), zero(T))
Base.:*(M::Union{MXMatrix{T}, LazyT{T}}, v::MXVector{T}) where {T} = mul!(MXVector{T}(undef, size(M, 1)), M, v)

# ---- mul! on matrices (src/operations.jl:34-36; wrappers src/adjtrans.jl:139-156, 207-224) ------------------------------
# The reference hands `res` and `m` to the closure as they are (`LinearOperator(M)`: a GEMM; the elementwise closures
# broadcast over the columns). The device closures are vector kernels, so a device matrix is applied column by column —
# the columns of an MXMatrix are contiguous views, nothing is copied (test/test_linop.jl:64-76: hcat(v, -2v)).
column(A::MXMatrix{T}, j::Integer) where {T} = view(A.data, ((j - 1) * A.m + 1):(j * A.m))
function apply_columns(f::DenseApply{T}, res::MXMatrix{T}, m::MXMatrix{T}, α, β) where {T}      # real dense: one call for the block
  size(res, 2) == size(m, 2) || throw(LinearOperatorException("shape mismatch"))
  f(res, m, α, β)
  res
end
function apply_columns(f, res::MXMatrix, m::MXMatrix, α, β)
  f === nothing && error("Not implemented")
  size(res, 2) == size(m, 2) || throw(LinearOperatorException("shape mismatch"))
  for j = 1:size(m, 2)
    f(column(res, j), column(m, j), α, β)
  end
  res
end
function LinearAlgebra.mul!(res::MXMatrix, op::LinearOperator{T, MXVector{T}}, m::MXMatrix, α, β) where {T}
  # the reference leaves this to BLAS / broadcast (DimensionMismatch); the device closures take raw pointers and their
  # sizes from the operator, so the row counts are checked here, before any launch
  (size(m, 1) == size(op, 2) && size(res, 1) == size(op, 1) && size(m, 2) == size(res, 2)) ||
    throw(LinearOperatorException("shape mismatch"))
  apply_columns(op.prod!, res, m, α, β)
end
function LinearAlgebra.mul!(res::MXMatrix, op::LinearOperators.AdjointLinearOperator{T, <:LinearOperator{T, MXVector{T}}},
                            m::MXMatrix, α, β) where {T}
  p = op.parent
  (size(m, 1) == size(p, 1) && size(res, 1) == size(p, 2) && size(m, 2) == size(res, 2)) ||
    throw(LinearOperatorException("shape mismatch"))
  LinearOperators.ishermitian(p) ? mul!(res, p, m, α, β) : apply_columns(p.ctprod!, res, m, α, β)
end
function LinearAlgebra.mul!(res::MXMatrix, op::LinearOperators.TransposeLinearOperator{T, <:LinearOperator{T, MXVector{T}}},
                            m::MXMatrix, α, β) where {T}
  p = op.parent
  (size(m, 1) == size(p, 1) && size(res, 1) == size(p, 2) && size(m, 2) == size(res, 2)) ||
    throw(LinearOperatorException("shape mismatch"))
  LinearOperators.issymmetric(p) ? mul!(res, p, m, α, β) : apply_columns(p.tprod!, res, m, α, β)
end

# ---- a7 mulRestrict! / multRestrict! (src/special-operators.jl:167-174) --------------------------------------------
# α, β are ignored by the reference and are not ABI parameters. `opRestriction(I, ncol; S = MXVector{T})` and
# `opExtension` are the reference's own constructors; their closures land here. Ranges need no device memory; an
# index Vector is uploaded once and cached per (objectid, length) together with its last-write-wins scatter plan.
mutable struct ScatterPlan    # duplicates: `res[I] = u` is sequential, the LAST write wins — resolved once, into a SORTED plan
  const didx::MXVector{Int64}       # I as given (gather)
  const idx::MXVector{Int64}        # strictly increasing target indices (scatter)
  const pos::Union{MXVector{Int64}, Nothing}   # 0-based position in u of the surviving write; nothing = identity
  const n::Int
  const hidx::Vector{Int64}         # the strictly increasing indices on the host (what an index plan is built from)
  const masks::Dict{Int, Ptr{Cvoid}}   # length of the long vector => mxlo_index_plan (bit mask + ranks), built on first use
end
# the cached index plans are library handles (n/4 bytes of device memory each): released with the plan (mutable => finalizable)
function release_masks(pl::ScatterPlan)
  for h in values(pl.masks)
    ccall((:mxlo_index_plan_destroy, lib), Int32, (P,), h)
  end
  empty!(pl.masks)
end
function ScatterPlan(I::AbstractVector{<:Integer})
  didx = MXVector(collect(Int64, I))
  (issorted(I) && allunique(I)) &&
    return finalizer(release_masks, ScatterPlan(didx, didx, nothing, length(I), collect(Int64, I), Dict{Int, Ptr{Cvoid}}()))
  last = Dict{Int64, Int64}()
  for (k, i) in enumerate(I)
    last[i] = k - 1                                   # 0-based source position of the surviving write
  end
  ks = sort!(collect(keys(last)))
  finalizer(release_masks, ScatterPlan(didx, MXVector(ks), MXVector([last[i] for i in ks]), length(ks), ks, Dict{Int, Ptr{Cvoid}}()))
end
# A dense enough strictly increasing index set is applied as bit mask + ranks (include/mxlo.h "index plans"): neither
# mulRestrict! nor multRestrict! then reads the index list. Built once per (I, length of the long vector); C_NULL below
# the density where the list is the smaller description (1/32).
function mask_plan(pl::ScatterPlan, n::Int)
  (pl.n > 0 && 32 * pl.n >= n) || return C_NULL
  get!(pl.masks, n) do
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:mxlo_index_plan_create, lib), Int32, (P, Ptr{Int64}, Int64, Int64, Ptr{P}), ctx(), pl.hidx, Int64(pl.n), Int64(n), h))
    h[]
  end
end
const PLANS = IdDict{Any, ScatterPlan}()
plan(I::Vector{<:Integer}) = get!(() -> ScatterPlan(I), PLANS, I)
const RangeIdx = Union{UnitRange{<:Integer}, StepRange{<:Integer, <:Integer}}
mulRestrict!(res::MXVector, Idx::RangeIdx, v::MXVector, α, β) =
  check(ccall((:mxlo_gather_range, lib), Int32, (P, Int32, P, P, Int64, Int64, Int64, Int64), ctx(),
              Int32(sizeof(eltype(v))), res.ptr, v.ptr, length(v), Int64(first(Idx)), Int64(step(Idx)), Int64(length(Idx))))
multRestrict!(res::MXVector, Idx::RangeIdx, u::MXVector, α, β) =
  check(ccall((:mxlo_scatter_zero_range, lib), Int32, (P, Int32, P, Int64, P, Int64, Int64, Int64), ctx(),
              Int32(sizeof(eltype(u))), res.ptr, length(res), u.ptr, Int64(first(Idx)), Int64(step(Idx)), Int64(length(Idx))))
function mulRestrict!(res::MXVector, Idx::Vector{<:Integer}, v::MXVector, α, β)
  pl = plan(Idx)
  mp = (pl.pos === nothing && 8 * pl.n >= length(v)) ? mask_plan(pl, length(v)) : C_NULL   # increasing I, >= 1/8 density
  mp == C_NULL || return check(ccall((:mxlo_gather_plan, lib), Int32, (P, Int32, P, P, Int64, P), ctx(), Int32(sizeof(eltype(v))),
                                     res.ptr, v.ptr, length(v), mp))
  check(ccall((:mxlo_gather, lib), Int32, (P, Int32, P, P, Int64, P, Int64), ctx(), Int32(sizeof(eltype(v))), res.ptr,
              v.ptr, length(v), pl.didx.ptr, length(pl.didx)))
end
function multRestrict!(res::MXVector, Idx::Vector{<:Integer}, u::MXVector, α, β)
  pl = plan(Idx)
  mp = mask_plan(pl, length(res))
  mp == C_NULL || return check(ccall((:mxlo_scatter_zero_plan, lib), Int32, (P, Int32, P, Int64, P, P, P), ctx(), Int32(sizeof(eltype(u))),
                                     res.ptr, length(res), u.ptr, pl.pos === nothing ? C_NULL : pl.pos.ptr, mp))
  check(ccall((:mxlo_scatter_zero_sorted, lib), Int32, (P, Int32, P, Int64, P, P, P, Int64), ctx(), Int32(sizeof(eltype(u))),
              res.ptr, length(res), u.ptr, pl.idx.ptr, pl.pos === nothing ? C_NULL : pl.pos.ptr, pl.n))
end

# ---- sparse LinearOperator(M::SparseMatrixCSC) (src/constructors.jl:15-29 -> SparseArrays' mul!) ---------------------
# `MXSparseMatrixCSC(A)` uploads the three arrays of a SparseMatrixCSC{T,Int64} AS STORED (1-based) and creates the
# library handle (compressed-row view + chunk tables, include/mxlo.h). Aᵀ*x / A'*x read `nzval` in place; A*x reads a
# row-ordered snapshot of the values: after changing `A.nzval` on the device call `refresh!(A)` (one gather pass).
mutable struct MXSparseMatrixCSC{T} <: AbstractMatrix{T}
  h::Ptr{Cvoid}
  m::Int
  n::Int
  colptr::MXVector{Int64}
  rowval::MXVector{Int64}
  nzval::MXVector{T}
end
function MXSparseMatrixCSC(m::Integer, n::Integer, colptr::Vector{Int64}, rowval::Vector{Int64}, nzval::Vector{T}) where {T <: Union{RealT, CplxT}}
  cp, rv, nz = MXVector(colptr), MXVector(rowval), MXVector(nzval)
  out = Ref{Ptr{Cvoid}}()
  check(ccall((:mxlo_csc_create, lib), Int32, (P, Int32, Int64, Int64, P, P, P, Int32, Ptr{P}),
              ctx(), dt(T), m, n, cp.ptr, rv.ptr, nz.ptr, Int32(1), out))
  finalizer(x -> ccall((:mxlo_csc_destroy, lib), Int32, (P,), x.h), MXSparseMatrixCSC{T}(out[], m, n, cp, rv, nz))
end
# any SparseMatrixCSC-like object with the documented field names (SparseArrays is not a dependency of this file)
MXSparseMatrixCSC(A) = MXSparseMatrixCSC(size(A, 1), size(A, 2), collect(Int64, A.colptr), collect(Int64, A.rowval), collect(A.nzval))
size(A::MXSparseMatrixCSC) = (A.m, A.n)
storage_type(::MXSparseMatrixCSC{T}) where {T} = MXVector{T}
refresh!(A::MXSparseMatrixCSC) = (check(ccall((:mxlo_csc_refresh, lib), Int32, (P,), A.h)); A)
function sparse_info(A::MXSparseMatrixCSC)
  info = Vector{Int64}(undef, 8)
  check(ccall((:mxlo_csc_info, lib), Int32, (P, Ptr{Int64}), A.h, info))
  (m = info[1], n = info[2], nnz = info[3], chunks = info[4], chunks_t = info[5], long_rows = info[6], long_cols = info[7])
end
struct SparseApply{T}
  A::MXSparseMatrixCSC{T}
  mode::Int32
end
(f::SparseApply{T})(res::MXVector{T}, v::MXVector{T}, α, β) where {T <: RealT} = check(ccall((:mxlo_csc_mul, lib), Int32,
    (P, P, P, Float64, Float64, Int32, Int32), f.A.h, res.ptr, v.ptr, α, β, f.mode, flags(T, α, β)))
# complex element types (test/test_linop.jl:44: simple_sparse_matrix(ComplexF64, …)): modes N / T on the values as stored,
# mode C conjugates them in the sweep; scalars as (re, im) pairs like every `_c` entry point
(f::SparseApply{T})(res::MXVector{T}, v::MXVector{T}, α, β) where {T <: CplxT} = check(ccall((:mxlo_csc_mul_c, lib), Int32,
    (P, P, P, Float64, Float64, Float64, Float64, Int32, Int32),
    f.A.h, res.ptr, v.ptr, real(α), imag(α), real(β), imag(β), f.mode, flags(T, α, β)))
function LinearOperator(A::MXSparseMatrixCSC{T}; symmetric = false, hermitian = false, S = MXVector{T}) where {T <: CplxT}
  LinearOperator{T, S}(A.m, A.n, symmetric, hermitian, SparseApply{T}(A, Int32(0)), SparseApply{T}(A, Int32(1)),
                       SparseApply{T}(A, Int32(2)))
end
# mul! on matrices (src/operations.jl:34-36): the stored matrix is read once per 8 columns of the block
(f::SparseApply{T})(res::MXMatrix{T}, V::MXMatrix{T}, α, β) where {T <: RealT} = check(ccall((:mxlo_csc_mul_block, lib), Int32,
    (P, P, Int64, P, Int64, Int64, Float64, Float64, Int32, Int32),
    f.A.h, res.data.ptr, res.m, V.data.ptr, V.m, size(V, 2), α, β, f.mode, flags(T, α, β)))
# (real element types only: mxlo_csc_mul_block refuses complex handles, and SparseApply has no matrix call method for
#  T <: CplxT — a complex sparse operator falls through to the generic column-by-column apply_columns, like the Python mirror)
function apply_columns(f::SparseApply{T}, res::MXMatrix{T}, m::MXMatrix{T}, α, β) where {T <: RealT}     # one call for the block
  size(res, 2) == size(m, 2) || throw(LinearOperatorException("shape mismatch"))
  f(res, m, α, β)
  res
end
function LinearOperator(A::MXSparseMatrixCSC{T}; symmetric = false, hermitian = false, S = MXVector{T}) where {T <: RealT}
  LinearOperator{T, S}(A.m, A.n, symmetric, hermitian, SparseApply{T}(A, Int32(0)), SparseApply{T}(A, Int32(1)),
                       SparseApply{T}(A, Int32(2)))
end
# the reference's generic BlockDiagonalOperator(A, B, C) calls mul! on plain matrix blocks (test/test_linop.jl:743-756 has
# a sprand block): the same for a device sparse block and its lazy transpose / adjoint
LinearAlgebra.mul!(res::MXVector{T}, A::MXSparseMatrixCSC{T}, v::MXVector{T}, α::Number, β::Number) where {T <: RealT} =
  (SparseApply{T}(A, Int32(0))(res, v, α, β); res)
LinearAlgebra.mul!(res::MXVector{T}, A::Transpose{T, MXSparseMatrixCSC{T}}, v::MXVector{T}, α::Number, β::Number) where {T <: RealT} =
  (SparseApply{T}(parent(A), Int32(1))(res, v, α, β); res)
LinearAlgebra.mul!(res::MXVector{T}, A::Adjoint{T, MXSparseMatrixCSC{T}}, v::MXVector{T}, α::Number, β::Number) where {T <: RealT} =
  (SparseApply{T}(parent(A), Int32(2))(res, v, α, β); res)

# ---- a8 BlockDiagonalOperator (src/special-operators.jl:249-294): ONE launch per apply --------------------
struct BlockDesc                # mxlo_block_desc, 56 bytes, same field order as include/mxlo.h
  kind::Int32
  reserved::Int32
  row_off::Int64
  col_off::Int64
  m::Int64
  n::Int64
  data::Ptr{Cvoid}
  ld::Int64
end
mutable struct BlockDiagHandle
  h::Ptr{Cvoid}
  keep::Vector{Any}             # block operands stay alive as long as the descriptor table
end
"blocks: opDiagonal data vectors (MXVector), dense MXMatrix blocks, sparse MXSparseMatrixCSC blocks (no row or column
above 2048 stored entries: `sparse_info`), `(:eye, n)` or `(:zeros, m, n)`."
function BlockDiagonalOperator(::Type{T}, blocks...; S = MXVector{T}) where {T}
  descs = BlockDesc[]
  r = c = 0
  for b in blocks
    if b isa MXVector
      push!(descs, BlockDesc(0, 0, r, c, length(b), length(b), b.ptr, 0)); r += length(b); c += length(b)
    elseif b isa MXMatrix
      push!(descs, BlockDesc(1, 0, r, c, b.m, b.n, b.data.ptr, b.m)); r += b.m; c += b.n
    elseif b isa MXSparseMatrixCSC
      push!(descs, BlockDesc(4, 0, r, c, b.m, b.n, b.h, 0)); r += b.m; c += b.n      # MXLO_BLK_CSC: data = the handle
    elseif b[1] === :eye
      push!(descs, BlockDesc(2, 0, r, c, b[2], b[2], C_NULL, 0)); r += b[2]; c += b[2]
    else
      push!(descs, BlockDesc(3, 0, r, c, b[2], b[3], C_NULL, 0)); r += b[2]; c += b[3]
    end
  end
  out = Ref{Ptr{Cvoid}}()
  check(ccall((:mxlo_blockdiag_create, lib), Int32, (P, Int32, Ptr{BlockDesc}, Int64, Ptr{P}),
              ctx(), dt(T), descs, length(descs), out))
  bd = finalizer(x -> ccall((:mxlo_blockdiag_destroy, lib), Int32, (P,), x.h), BlockDiagHandle(out[], collect(Any, blocks)))
  mulmode(mode) = (res, v, α, β) -> check(ccall((:mxlo_blockdiag_mul, lib), Int32,
      (P, P, P, Float64, Float64, Int32, Int32), bd.h, res.ptr, v.ptr, α, β, Int32(mode), flags(T, α, β)))
  symm = all(b -> !(b isa MXMatrix) && !(b isa MXSparseMatrixCSC) && !(b isa Tuple && b[1] === :zeros && b[2] != b[3]), blocks)
  LinearOperator{T, S}(r, c, symm, symm, mulmode(0), mulmode(1), mulmode(2))
end

# ---- a9 kron (src/kron.jl:10-49): two MFMA GEMMs, no CPU copy of x (:16), work vector allocated once --------
function kron(A::MXMatrix{T}, B::MXMatrix{T}) where {T <: RealT}
  m, n = size(A)
  p, q = size(B)
  work = MXVector{T}(undef, max(p * n, q * m))
  mulmode(mode) = (res, x, α, β) -> check(ccall((:mxlo_kron_mul, lib), Int32,
      (P, Int32, P, P, Int64, Int64, Int64, P, Int64, Int64, Int64, P, P, Float64, Float64, Int32, Int32),
      ctx(), dt(T), res.ptr, A.data.ptr, m, n, m, B.data.ptr, p, q, p, x.ptr, work.ptr, α, β, Int32(mode),
      flags(T, α, β)))
  LinearOperator{T, MXVector{T}}(m * p, n * q, false, false, mulmode(0), mulmode(1), mulmode(2))
end
# kron with lazily transposed factors (`kron(transpose(A), B)`): a transposition is a flag of the GEMM kernel's
# operand layout, not a copy (mxlo_kron_mul_ex: per-factor flags; tprod!/ctprod! flip both).
const MaybeT{T} = Union{MXMatrix{T}, Transpose{T, MXMatrix{T}}, Adjoint{T, MXMatrix{T}}}
stored(A::MXMatrix) = (A, Int32(0))
stored(A::Union{Transpose, Adjoint}) = (parent(A), Int32(1))
function kron(A::MaybeT{T}, B::MaybeT{T}) where {T <: RealT}
  (As, ta), (Bs, tb) = stored(A), stored(B)
  m, n = size(A)
  p, q = size(B)
  work = MXVector{T}(undef, max(p * n, q * m))
  mulmode(tr) = (res, x, α, β) -> check(ccall((:mxlo_kron_mul_ex, lib), Int32,
      (P, Int32, P, P, Int64, Int64, Int64, Int32, P, Int64, Int64, Int64, Int32, P, P, Float64, Float64, Int32),
      ctx(), dt(T), res.ptr, As.data.ptr, As.m, As.n, As.m, ta ⊻ Int32(tr), Bs.data.ptr, Bs.m, Bs.n, Bs.m, tb ⊻ Int32(tr),
      x.ptr, work.ptr, α, β, flags(T, α, β)))
  LinearOperator{T, MXVector{T}}(m * p, n * q, false, false, mulmode(0), mulmode(1), mulmode(1))
end

# kron with complex factors (test/test_kron.jl:3-8 pairs a Float64 A with a ComplexF64 B): the library works on REAL
# planes — a complex factor is split once into (re, im) column-major MXMatrix planes, a real factor is passed as it is
# with a NULL imaginary plane; every complex product is 3 (Gauss form; 2 for a real factor) real MFMA GEMMs. mode bit 0 = transposed, bit 1 = conjugated.
struct Planes{R}
  re::MXMatrix{R}
  im::Union{MXMatrix{R}, Nothing}
end
planes(A::MXMatrix{R}) where {R <: RealT} = Planes{R}(A, nothing)
function planes(A::MXMatrix{Complex{R}}) where {R <: RealT}
  h = Array(A.data)                                     # split ONCE at construction (host round trip is fine here)
  m, n = size(A)
  Planes{R}(MXMatrix{R}(MXVector(real.(h)), m, n), MXMatrix{R}(MXVector(imag.(h)), m, n))
end
function kron(A::MXMatrix{TA}, B::MXMatrix{TB}) where {TA <: Union{RealT, CplxT}, TB <: CplxT}
  ckron(A, B)
end
kron(A::MXMatrix{TA}, B::MXMatrix{TB}) where {TA <: CplxT, TB <: RealT} = ckron(A, B)
function ckron(A::MXMatrix, B::MXMatrix)
  T = promote_type(eltype(A), eltype(B))
  R = real(T)
  pa, pb = planes(A), planes(B)
  (eltype(pa.re) === R && eltype(pb.re) === R) || throw(ArgumentError("kron: convert the factors to a common precision first"))
  m, n = size(A)
  p, q = size(B)
  # Gauss form: 3 real GEMMs per complex product (mxlo_kron_mul_c3); the library states its workspace need itself
  wsz(mode) = ccall((:mxlo_kron_c3_work_size, lib), Int64, (Int64, Int64, Int32, Int64, Int64, Int32), m, n, Int32(mode), p, q, Int32(mode))
  work = MXVector{R}(undef, max(wsz(0), wsz(1)))
  imptr(x) = x === nothing ? C_NULL : x.data.ptr
  # factor-sum planes re + s*im of the Gauss form, formed ONCE here (the factors of this constructor are immutable
  # snapshots: `planes` copied them): s = +1 for prod! / tprod!, -1 for ctprod! (conjugated factors)
  function sumplane(pl, rows, cols, s)
    pl.im === nothing && return nothing
    out = MXVector{R}(undef, rows * cols)
    check(ccall((:mxlo_plane_sum, lib), Int32, (P, Int32, P, P, P, Int64, Int64, Int64, Float64),
                ctx(), dt(T), out.ptr, pl.re.data.ptr, pl.im.data.ptr, rows, cols, rows, s))
    out
  end
  sums = Dict(s => (sumplane(pa, m, n, s), sumplane(pb, p, q, s)) for s in (1.0, -1.0))
  vptr(x) = x === nothing ? C_NULL : x.ptr
  mulmode(mode) = (res, x, α, β) -> begin
    sa, sb = sums[(mode & 2) != 0 ? -1.0 : 1.0]
    check(ccall((:mxlo_kron_mul_c3, lib), Int32,
      (P, Int32, P, P, P, P, Int64, Int64, Int64, Int32, P, P, P, Int64, Int64, Int64, Int32, P, P, Float64, Float64, Float64, Float64, Int32),
      ctx(), dt(T), res.ptr, pa.re.data.ptr, imptr(pa.im), vptr(sa), m, n, m, Int32(mode), pb.re.data.ptr, imptr(pb.im), vptr(sb), p, q, p,
      Int32(mode), x.ptr, work.ptr, rpart(α), ipart(α), rpart(β), ipart(β), flags(T, α, β)))
  end
  LinearOperator{T, MXVector{T}}(m * p, n * q, false, false, mulmode(0), mulmode(1), mulmode(3))
end

# ---- a REAL operator applied to complex device vectors (test/test_kron.jl "issue110": K * x with x::Vector{ComplexF64})
# The leaves are instantiated per element type, so the real operator is applied to the two planes of x and the result is
# joined with the caller's (possibly complex) α, β: res = α*(op*real(x) + i*op*imag(x)) + β*res.
function LinearAlgebra.mul!(res::MXVector{Complex{R}}, op::LinearOperators.AbstractLinearOperator{R}, v::MXVector{Complex{R}},
                            α, β) where {R <: RealT}
  T = Complex{R}
  xr, xi = MXVector{R}(undef, length(v)), MXVector{R}(undef, length(v))
  yr, yi = MXVector{R}(undef, length(res)), MXVector{R}(undef, length(res))
  check(ccall((:mxlo_split_c, lib), Int32, (P, Int32, P, P, P, Int64), ctx(), dt(T), xr.ptr, xi.ptr, v.ptr, length(v)))
  mul!(yr, op, xr)
  mul!(yi, op, xi)
  check(ccall((:mxlo_join_c, lib), Int32, (P, Int32, P, P, P, Int64, Float64, Float64, Float64, Float64, Int32),
              ctx(), dt(T), res.ptr, yr.ptr, yi.ptr, length(res), rpart(α), ipart(α), rpart(β), ipart(β), flags(T, α, β)))
  res
end

# a REAL device vector handed to a COMPLEX operator (`aopA * rand(5)`, test/test_adjtrans.jl:31-34): promoted on the device
function LinearAlgebra.mul!(res::MXVector{Complex{R}}, op::LinearOperators.AbstractLinearOperator{Complex{R}}, v::MXVector{R},
                            α, β) where {R <: RealT}
  T = Complex{R}
  vc = MXVector{T}(undef, length(v))
  check(ccall((:mxlo_join_c, lib), Int32, (P, Int32, P, P, P, Int64, Float64, Float64, Float64, Float64, Int32),
              ctx(), dt(T), vc.ptr, v.ptr, C_NULL, length(v), 1.0, 0.0, 0.0, 0.0, Int32(0x20 | 0x40)))
  mul!(res, op, vc, α, β)
end

# ---- a12-a16 quasi-Newton operators: the structural contract (src/lbfgs.jl:62-104, src/lsr1.jl:39-78) ---------
struct MXQNData                # stands in for op.data: fields the reference's tests read come from the handle
  h::Ptr{Cvoid}
  mem::Int                     # max(mem, 1) (lbfgs.jl:37, lsr1.jl:21)
  scaling::Bool                # the `const` fields of LBFGSData / LSR1Data (lbfgs.jl:6-8, lsr1.jl:6)
  damped::Bool
end
function scalars(d::MXQNData)
  sc = Vector{Float64}(undef, 5); ys = Vector{Float64}(undef, d.mem); aux = Vector{Float64}(undef, d.mem)
  check(ccall((:mxlo_qn_get_scalars, lib), Int32, (P, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), d.h, sc, ys, aux))
  (insert = Int(sc[1]), scaling_factor = sc[2], opnorm_upper_bound = sc[3], ys = ys, aux = aux)
end
Base.getproperty(d::MXQNData, s::Symbol) =
  s in (:h, :mem, :scaling, :damped) ? getfield(d, s) : getproperty(scalars(d), s)   # op.data.insert, .scaling_factor, .ys

mutable struct MXQNOperator{T, F, Ft} <: AbstractQuasiNewtonOperator{T}
  const nrow::Int
  const ncol::Int
  const symmetric::Bool
  const hermitian::Bool
  const prod!::F
  const tprod!::Ft
  const ctprod!::Ft
  const kind::Int32             # MXLO_QN_LBFGS_INV / _FWD / _LSR1
  const inverse::Bool
  const data::MXQNData
  nprod::Int
  ntprod::Int
  nctprod::Int
end
has_args5(::MXQNOperator) = true
isallocated5(::MXQNOperator) = true
storage_type(::MXQNOperator{T}) where {T} = MXVector{T}

# Keywords and defaults are the reference's: LBFGSData(T, n; mem = 5, scaling = true, damped = false, inverse = true,
# σ₂ = 0.99, σ₃ = 10.0) (src/lbfgs.jl:26-35) and LSR1Data(T, n; mem = 5, scaling = true) (src/lsr1.jl:19 — the CODE default is
# `true`; the docstring at lsr1.jl:81 says `false` and is wrong). `inverse` is accepted and ignored like the reference's
# constructors do (`delete!(kwargs, :inverse)`, lbfgs.jl:115,171): the constructor's name decides. LSR1Data takes neither
# `damped` nor σ₂/σ₃ (a MethodError upstream): lsr1 below passes only its own two.
function mxqn(::Type{T}, kind::Integer, n::Int; mem::Int = 5, scaling::Bool = true, damped::Bool = false,
              inverse::Bool = true, σ₂::Float64 = 0.99, σ₃::Float64 = 10.0) where {T}
  r = Ref{Ptr{Cvoid}}()
  check(ccall((:mxlo_qn_create, lib), Int32, (P, Int32, Int32, Int64, Int64, Int32, Int32, Float64, Float64, Ptr{P}),
              ctx(), Int32(kind), dt(T), n, mem, scaling, damped, σ₂, σ₃, r))
  h = r[]
  prod! = (res, x, α, β) -> check(ccall((:mxlo_qn_mul, lib), Int32, (P, P, P, Float64, Float64, Int32),
                                        h, res.ptr, x.ptr, α, β, flags(T, α, β)))
  t = kind == 2 ? nothing : prod!                                 # L-SR1: tprod! = ctprod! = nothing (lsr1.jl:108-110)
  op = MXQNOperator{T, typeof(prod!), typeof(t)}(n, n, true, true, prod!, t, t, Int32(kind), kind == 0,
                                                 MXQNData(h, max(mem, 1), scaling, damped), 0, 0, 0)
  finalizer(o -> ccall((:mxlo_qn_destroy, lib), Int32, (P,), o.data.h), op)
end
lsr1(::Type{T}, n::Int; mem::Int = 5, scaling::Bool = true) where {T} = mxqn(T, 2, n; mem = mem, scaling = scaling)
InverseLBFGSOperator(::Type{T}, n::Int, ::Type{MXVector{T}}; kw...) where {T} = mxqn(T, 0, n; kw...)
LBFGSOperator(::Type{T}, n::Int, ::Type{MXVector{T}}; kw...) where {T} = mxqn(T, 1, n; kw...)
LSR1Operator(::Type{T}, n::Int, ::Type{MXVector{T}}; kw...) where {T} = lsr1(T, n; kw...)
# the `T`-less forms (lbfgs.jl:160,208, lsr1.jl:113: Float64): the storage type names the element type
InverseLBFGSOperator(n::Int, ::Type{MXVector{T}}; kw...) where {T} = mxqn(T, 0, n; kw...)
LBFGSOperator(n::Int, ::Type{MXVector{T}}; kw...) where {T} = mxqn(T, 1, n; kw...)
LSR1Operator(n::Int, ::Type{MXVector{T}}; kw...) where {T} = lsr1(T, n; kw...)

# push! ×4 (src/lbfgs.jl:269-367) + L-SR1's single method (src/lsr1.jl:119-184); rejection of a pair is silent, exactly like
# the reference. The guards are the reference's, statement for statement (tests/test_julia_semantics.py evaluates both texts
# over every (damped, inverse, arity) and compares with the Python mirror); the library would refuse the same calls with
# MXLO_ESTATE (also an ErrorException), but the messages and the redirects are part of the contract.
function push!(op::MXQNOperator{T}, s::MXVector{T}, y::MXVector{T}) where {T}
  if op.data.damped                                         # lbfgs.jl:274-276 (L-SR1 is never damped)
    return push!(op, s, y, similar(s))
  end
  acc = Ref{Int32}(0)
  check(ccall((:mxlo_qn_push, lib), Int32, (P, P, P, Ptr{Int32}), op.data.h, s.ptr, y.ptr, acc))
  op
end
function push!(op::MXQNOperator{T}, s::MXVector{T}, y::MXVector{T}, Bs::MXVector{T}) where {T}
  op.kind == 2 && throw(MethodError(push!, (op, s, y, Bs)))   # lsr1.jl:119 is the only method for LSR1Operator
  if !op.data.damped                                        # lbfgs.jl:295-299
    error("This push! should be used for damped operators")
  elseif op.inverse
    error("This function be used for forward operators. Use push!(op, s, y, α, g, Bs) instead.")
  end
  acc = Ref{Int32}(0)
  check(ccall((:mxlo_qn_push_damped_fwd, lib), Int32, (P, P, P, P, Ptr{Int32}), op.data.h, s.ptr, y.ptr, Bs.ptr, acc))
  op
end
function push!(op::MXQNOperator{T}, s::MXVector{T}, y::MXVector{T}, α::T, g::MXVector{T}, Bs::MXVector{T}) where {T}
  op.kind == 2 && throw(MethodError(push!, (op, s, y, α, g, Bs)))
  if !op.data.damped                                        # lbfgs.jl:333-337
    error("This push! should be used for damped operators")
  elseif !op.inverse
    error("This function be used for inverse operators. Use push!(op, s, y, Bs) instead.")
  end
  acc = Ref{Int32}(0)
  check(ccall((:mxlo_qn_push_damped_inv, lib), Int32, (P, P, P, Float64, P, P, Ptr{Int32}),
              op.data.h, s.ptr, y.ptr, α, g.ptr, Bs.ptr, acc))
  op
end
function push!(op::MXQNOperator{T}, s::MXVector{T}, y::MXVector{T}, α::T, g::MXVector{T}) where {T}   # lbfgs.jl:359-367
  push!(op, s, y, α, g, similar(g))
end
function reset!(op::MXQNOperator)
  check(ccall((:mxlo_qn_reset, lib), Int32, (P,), op.data.h))
  op.nprod = op.ntprod = op.nctprod = 0
  op
end
function diag!(op::MXQNOperator{T}, d::MXVector{T}) where {T}
  check(ccall((:mxlo_qn_diag, lib), Int32, (P, P), op.data.h, d.ptr))
  d
end
LinearAlgebra.diag(op::MXQNOperator{T}) where {T} = diag!(op, MXVector{T}(undef, op.nrow))

# ShiftedOperator(H, σ) over a quasi-Newton H: shifted_prod! (src/shifted_operators.jl:16-25) in ONE apply —
# the axpy!(α σ, x, y) rides in the combine pass (bit-identical to mul! followed by axpy! in T arithmetic).
function LinearOperators.shifted_prod!(y::MXVector{T}, data::LinearOperators.ShiftedData{T, <:MXQNOperator{T}},
                                       x::MXVector{T}, α, β) where {T}
  data.H.nprod += 1
  check(ccall((:mxlo_qn_mul_shifted, lib), Int32, (P, P, P, Float64, Float64, Float64, Int32),
              data.H.data.h, y.ptr, x.ptr, α, β, data.σ, flags(T, α, β)))
  y
end

# solve_shifted_system! / ldiv! (src/utilities.jl:207-248, 281-289); returns x itself (test_solve_shifted_system.jl:33)
function solve_shifted_system!(x::MXVector{T}, B::MXQNOperator{T}, b::MXVector{T}, σ::T) where {T}
  check(ccall((:mxlo_qn_solve_shifted, lib), Int32, (P, P, P, Float64), B.data.h, x.ptr, b.ptr, σ))
  x
end
ldiv!(x::MXVector{T}, B::MXQNOperator{T}, b::MXVector{T}) where {T} = solve_shifted_system!(x, B, b, zero(T))

# ---- diagonal quasi-Newton family (src/DiagonalHessianApproximation.jl) ------------------------------------------
# The structs are generic in the vector type V, so DiagonalPSB(d::MXVector) etc. construct unchanged and their mul!
# reaches mulSquareOpDiagonal!; the extension supplies that kernel and the fused push!/reset!.
# (their mul! is mulSquareOpDiagonal! on MXVectors, defined with the other kernel-function methods above)
dqn_kind(::LinearOperators.DiagonalPSB) = Int32(0)
dqn_kind(::LinearOperators.DiagonalAndrei) = Int32(1)
dqn_kind(::LinearOperators.DiagonalBFGS) = Int32(2)
function push!(B::Union{LinearOperators.DiagonalPSB{T}, LinearOperators.DiagonalAndrei{T}, LinearOperators.DiagonalBFGS{T}},
               s::MXVector{T}, y::MXVector{T}) where {T}
  st = Ref{Int32}(0)
  check(ccall((:mxlo_diagqn_push, lib), Int32, (P, Int32, Int32, P, P, P, Int64, Ptr{Int32}),
              ctx(), dt(T), dqn_kind(B), B.d.ptr, s.ptr, y.ptr, length(s), st))
  st[] == 0 || error("Cannot update DiagonalQN operator with s=0")
  B
end
function reset!(op::LinearOperators.AbstractDiagonalQuasiNewtonOperator{T}) where {T}   # d::MXVector
  op.d isa MXVector || return invoke(reset!, Tuple{LinearOperators.AbstractQuasiNewtonOperator}, op)
  check(ccall((:mxlo_fill, lib), Int32, (P, Int32, P, Int64, Float64), ctx(), dt(T), op.d.ptr, length(op.d), 1.0))
  op.nprod = op.ntprod = op.nctprod = 0
  op
end
# SpectralGradient hard-codes d::Vector{T} (:151): the device form keeps the single element in an MXVector{T}(1)
# and calls mxlo_diagqn_push with kind 3 (MXLO_DQN_SPECTRAL).

# ---- hipGraph replay of launch-bound inner loops (mxlo_graph_*) ----------------------------------------------------
"`g = capture(() -> (mul!(r, A, x); mul!(y, B, r, 1.0, 1.0)))` then `replay(g)` in the solver loop."
mutable struct Graph; h::Ptr{Cvoid}; end
function capture(f)
  check(ccall((:mxlo_ctx_create_stream, lib), Int32, (P, Ptr{P}), ctx(), C_NULL))   # a non-default stream owned by the ctx
  f()                                                                              # warm-up: lazy temporaries, workspaces
  check(ccall((:mxlo_graph_begin, lib), Int32, (P,), ctx()))
  f()
  r = Ref{Ptr{Cvoid}}()
  check(ccall((:mxlo_graph_end, lib), Int32, (P, Ptr{P}), ctx(), r))
  finalizer(g -> ccall((:mxlo_graph_destroy, lib), Int32, (P,), g.h), Graph(r[]))
end
replay(g::Graph) = check(ccall((:mxlo_graph_launch, lib), Int32, (P,), g.h))

# ---- row sharding: one Julia process per GPU (DESIGN.md §6) -------------------------------------------------
"`id` = the 128 bytes rank 0 obtained from `rccl_unique_id()`, broadcast by MPI.jl / a file / sockets."
rccl_unique_id() = (id = zeros(UInt8, 128); check(ccall((:mxlo_rccl_unique_id, rccl), Int32, (Ptr{UInt8},), id)); id)
function install_rccl!(rank::Integer, world::Integer, id::Vector{UInt8})
  comm = Ref{Ptr{Cvoid}}()
  ccall((:mxlo_rccl_comm_create, rccl), Int32, (Int32, Int32, Ptr{UInt8}, Ptr{P}), rank, world, id, comm) == 0 ||
    error(unsafe_string(ccall((:mxlo_rccl_last_error, rccl), Cstring, ())))
  hook = cglobal((:mxlo_rccl_allreduce_hook, rccl))          # a C function: no Julia code runs inside an apply
  check(ccall((:mxlo_ctx_set_allreduce, lib), Int32, (P, P, P), ctx(), hook, comm[]))
  comm[]
end

# ---- row sharding inside ONE Julia process (include/mxlo_rccl.h, single-process API) -------------------------------
# `sc = ShardCtx([0, 1, ..., 7])` builds per device a stream, a ctx, an RCCL communicator (ncclCommInitAll), the
# all-reduce hook and a worker thread inside libmxlo_rccl.so; a `_sharded` call takes one pointer per device.
mutable struct ShardCtx
  h::Ptr{Cvoid}
  ndev::Int
end
# transport of the scalar all-reduce (include/mxlo_rccl.h): :auto (RCCL for distinct devices, loopback for repeated ids),
# :rccl, :loopback, :peer (the peer-mapped one-shot exchange: mailboxes in fine-grained device memory, one kernel per
# collective, no RCCL call)
const SHARD_TRANSPORT = Dict(:auto => Int32(0), :rccl => Int32(1), :loopback => Int32(2), :peer => Int32(3))
function ShardCtx(devs::Vector{<:Integer}; transport::Symbol = :auto)
  r = Ref{Ptr{Cvoid}}()
  ids = collect(Int32, devs)
  st = ccall((:mxlo_shard_ctx_create_ex, rccl), Int32, (Int32, Ptr{Int32}, Int32, Ptr{P}), length(ids), ids, SHARD_TRANSPORT[transport], r)
  st == 0 || error(unsafe_string(ccall((:mxlo_shard_last_error, rccl), Cstring, ())))
  finalizer(s -> ccall((:mxlo_shard_ctx_destroy, rccl), Int32, (P,), s.h), ShardCtx(r[], length(ids)))
end
"""
    preflight(sc; reps = 50, timeout_ms = 60_000) -> (latency_us_8B, latency_us_320B, latency_us_6912B)

Collective over all shards, BEFORE any production collective: known-answer and identical-bits all-reduces of the three
payload sizes of the hot path through the transport the `_sharded` entry points use, an agreed verdict, and the latency of
back-to-back all-reduces. Throws (naming shard, payload and phase) instead of hanging when a device does not answer.
"""
function preflight(sc::ShardCtx; reps::Integer = 50, timeout_ms::Integer = 60_000)
  lat = zeros(Float64, 3)
  scheck(ccall((:mxlo_shard_ctx_preflight, rccl), Int32, (P, Int32, Int32, Ptr{Float64}), sc.h, Int32(reps), Int32(timeout_ms), lat))
  (lat[1], lat[2], lat[3])
end
shard_ctx(sc::ShardCtx, i::Integer) = ccall((:mxlo_shard_ctx_get, rccl), P, (P, Int32), sc.h, i)   # for mxlo_malloc / memcpy
shard_sync(sc::ShardCtx) = check(ccall((:mxlo_shard_ctx_sync, rccl), Int32, (P,), sc.h))
scheck(st::Int32) = st == 0 ? nothing : error(unsafe_string(ccall((:mxlo_shard_last_error, rccl), Cstring, ())))
ptrs(vs) = P[Ptr{Cvoid}(v.ptr) for v in vs]
"`res`, `h`, `v`: one MXVector per device (row ranges); the only exchange is the 8-byte all-reduce of h'v."
function householder_mul_sharded!(sc::ShardCtx, res, h, v, α, β)
  T = eltype(first(res))
  nloc = Int64[length(r) for r in res]
  scheck(ccall((:mxlo_householder_mul_sharded, rccl), Int32, (P, Int32, Ptr{P}, Ptr{P}, Ptr{P}, Ptr{Int64}, Float64, Float64, Int32),
               sc.h, dt(T), ptrs(res), ptrs(h), ptrs(v), nloc, α, β, flags(T, α, β)))
  res
end
mutable struct ShardedQN{T}
  h::Ptr{Cvoid}
  sc::ShardCtx
end
# keywords and defaults as mxqn (= the reference's LBFGSData / LSR1Data); the sharded handle has the plain push! only
# (mxlo_qn_push_sharded): a damped sharded operator is refused at its first push! (MXLO_ESTATE -> ErrorException)
function ShardedQN(::Type{T}, sc::ShardCtx, kind::Integer, nloc::Vector{<:Integer}; mem::Int = 5, scaling::Bool = true,
                   damped::Bool = false, inverse::Bool = true, σ₂::Float64 = 0.99, σ₃::Float64 = 10.0) where {T <: RealT}
  r = Ref{Ptr{Cvoid}}()
  scheck(ccall((:mxlo_qn_create_sharded, rccl), Int32, (P, Int32, Int32, Ptr{Int64}, Int64, Int32, Int32, Float64, Float64, Ptr{P}),
               sc.h, Int32(kind), dt(T), collect(Int64, nloc), mem, scaling, damped, σ₂, σ₃, r))
  finalizer(q -> ccall((:mxlo_qn_destroy_sharded, rccl), Int32, (P,), q.h), ShardedQN{T}(r[], sc))
end
function push!(q::ShardedQN, s, y)
  acc = Ref{Int32}(0)
  scheck(ccall((:mxlo_qn_push_sharded, rccl), Int32, (P, Ptr{P}, Ptr{P}, Ptr{Int32}), q.h, ptrs(s), ptrs(y), acc))
  q
end
function qn_mul_sharded!(q::ShardedQN{T}, res, x, α, β) where {T}
  scheck(ccall((:mxlo_qn_mul_sharded, rccl), Int32, (P, Ptr{P}, Ptr{P}, Float64, Float64, Int32), q.h, ptrs(res), ptrs(x),
               α, β, flags(T, α, β)))
  res
end
function solve_shifted_sharded!(q::ShardedQN{T}, x, b, σ) where {T}
  scheck(ccall((:mxlo_qn_solve_shifted_sharded, rccl), Int32, (P, Ptr{P}, Ptr{P}, Float64), q.h, ptrs(x), ptrs(b), σ))
  x
end
reset!(q::ShardedQN) = (scheck(ccall((:mxlo_qn_reset_sharded, rccl), Int32, (P,), q.h)); q)

end # module
