# runtests_mxlo.jl — smoke / parity script for a Julia-equipped MI355X host (NOT EXECUTED in the build image: no Julia).
#
#   LD_LIBRARY_PATH=<repo>/linearoperators.jl_amd/csrc julia --project=<env with LinearOperators, Krylov> runtests_mxlo.jl
#
# Every device operator is compared with the SAME reference operator on host Vectors (LinearOperators.jl itself is the
# oracle here), with the tolerances of DESIGN.md §2; the allocation contract of test/test_lbfgs.jl:180-218 is checked
# through mxlo_debug_counters. Mirrors, in Julia, what tests/test_gpu_*.py check through the same C ABI.
using Test, LinearAlgebra, LinearOperators
include(joinpath(@__DIR__, "LinearOperatorsMXLOExt.jl"))
const MX = LinearOperatorsMXLOExt
using .LinearOperatorsMXLOExt: MXVector, MXMatrix
const lib = MX.lib

dev(x::Vector) = MXVector(x)
host(x::MXVector) = Array(x)
rel(a, b) = norm(a - b) / max(norm(b), floatmin(Float64))
counters() = (c = zeros(Int64, 12); ccall((:mxlo_debug_counters, lib), Int32, (Ptr{Int64},), c); c)

@testset "elementwise leaves are bit-exact" begin
  for T in (Float64, Float32, ComplexF64, ComplexF32), n in (1, 7, 1000, 100_003)
    d, v, r = rand(T, n), rand(T, n), rand(T, n)
    Dh, Dd = opDiagonal(d), opDiagonal(dev(d))
    for (α, β) in ((one(T), zero(T)), (T(2), T(-3)), (2.0, -3.0), (2.0f0, 0.5))
      rh = copy(r); mul!(rh, Dh, v, α, β)
      rd = dev(copy(r)); mul!(rd, Dd, dev(v), α, β)
      @test host(rd) == rh
      rh = copy(r); mul!(rh, Dh', v, α, β)
      rd = dev(copy(r)); mul!(rd, Dd', dev(v), α, β)
      @test host(rd) == rh
    end
    Ih, Id = opEye(T, n), opEye(T, n; S = MXVector{T})
    @test host(Id * dev(v)) == Ih * v
  end
end

@testset "opHouseholder / opHermitian / dense / restriction" begin
  n = 2000
  for T in (Float64, ComplexF64)
    h = rand(T, n); h ./= norm(h)
    v = rand(T, n)
    @test rel(host(opHouseholder(dev(h)) * dev(v)), opHouseholder(h) * v) <= 1e-12
    A = rand(T, 300, 300); d = rand(real(T), 300); x = rand(T, 300)
    @test rel(host(opHermitian(dev(d), MXMatrix(A)) * dev(x)), opHermitian(d, A) * x) <= 1e-12
    M = rand(T, 200, 300)
    opd, oph = LinearOperator(MXMatrix(M)), LinearOperator(M)
    @test rel(host(opd * dev(x)), oph * x) <= 1e-12
    u = rand(T, 200)
    @test rel(host(opd' * dev(u)), oph' * u) <= 1e-12
    @test rel(host(transpose(opd) * dev(u)), transpose(oph) * u) <= 1e-12
    # mul! on matrices (test/test_linop.jl:64-76: hcat(v, -2v)) and op ± scalar (:164-176) next to device operators
    mv, mu = hcat(x, -2x), hcat(u, -2u)
    rm = MXMatrix(zeros(T, 200, 2)); mul!(rm, opd, MXMatrix(mv))
    @test rel(host(rm.data), vec(M * mv)) <= 1e-12
    rt = MXMatrix(zeros(T, 300, 2)); mul!(rt, transpose(opd), MXMatrix(mu))
    @test rel(host(rt.data), vec(transpose(M) * mu)) <= 1e-12
    ra = MXMatrix(zeros(T, 300, 2)); mul!(ra, opd', MXMatrix(mu), 2.0, 0.0)
    @test rel(host(ra.data), vec(2.0 .* (M' * mu))) <= 1e-12
    @test rel(host((opd + 2.12345) * dev(x)), (M .+ 2.12345) * x) <= 1e-12
    @test rel(host((2.12345 - opd) * dev(x)), (2.12345 .- M) * x) <= 1e-12
  end
  I = [5, 2, 9, 2, 7]
  v = rand(20)
  R = opRestriction(I, 20; S = MXVector{Float64})
  res = MXVector{Float64}(undef, 5); mul!(res, R, dev(v))
  @test host(res) == v[I]
  u = rand(5); back = MXVector{Float64}(undef, 20); mul!(back, R', dev(u))
  want = zeros(20); want[I] = u
  @test host(back) == want                                   # duplicates: the last write wins
end

@testset "kron, quasi-Newton operators, solve_shifted_system!" begin
  A, B = rand(40, 30), rand(20, 50)
  x = rand(30 * 50)
  @test rel(host(kron(MXMatrix(A), MXMatrix(B)) * dev(x)), kron(A, B) * x) <= 1e-12
  n, mem = 10_000, 5
  Bd, Bh = LBFGSOperator(Float64, n, MXVector{Float64}; mem = mem), LBFGSOperator(n; mem = mem)
  Hd, Hh = InverseLBFGSOperator(Float64, n, MXVector{Float64}; mem = mem), InverseLBFGSOperator(n; mem = mem)
  for _ = 1:(mem + 3)
    s = rand(n); y = s .* (0.5 .+ rand(n))
    push!(Bd, dev(s), dev(y)); push!(Bh, s, y)
    push!(Hd, dev(s), dev(y)); push!(Hh, s, y)
  end
  v = rand(n)
  @test rel(host(Bd * dev(v)), Bh * v) <= 1e-9
  @test rel(host(Hd * dev(v)), Hh * v) <= 1e-9
  @test Bd.data.insert == Bh.data.insert
  b = rand(n); xs = MXVector{Float64}(undef, n)
  @test rel(host(solve_shifted_system!(xs, Bd, dev(b), 0.1)), solve_shifted_system!(zeros(n), Bh, b, 0.1)) <= 1e-8
  # the reference's only performance contract (test/test_lbfgs.jl:180-218): a warmed mul! asks nothing of the allocator
  res = MXVector{Float64}(undef, n); dv = dev(v)
  mul!(res, Bd, dv); mul!(res, Hd, dv)
  c0 = counters()
  for _ = 1:10
    mul!(res, Bd, dv); mul!(res, Hd, dv)
  end
  dc = counters() .- c0
  @test dc[1] == 0 && dc[2] == 0 && dc[3] == 0 && dc[4] == 0 && dc[7] == 0 && dc[8] == 0     # malloc, free, H2D, D2H, syncs
  @test dc[11] >= 20                                                                          # launches only
end

# ---- the reference's own device test for storage-type propagation, run with the extension's array types exactly the way
# the reference runs it for JLArrays / CUDA / AMDGPU (test/gpu/jlarrays.jl:1, test/gpu/amdgpu.jl:2): include ITS file and
# call ITS function — `arrayType(rand(Float32, 32, 32))` is an MXMatrix, `arrayType(rand(Float32, 32))` an MXVector.
# The assertions it makes (test/gpu/test_S_kwarg.jl:15-43) cover LinearOperator(mat), LinearOperator(mat; S = vecTother),
# LinearOperator(Symmetric(mat); S = vecT), LinearOperator(Hermitian(mat); S = vecT),
# LinearOperator(Float32, 32, 32, true, true, () -> 0; S = vecT), opEye(Float32, 32; S = vecT), opEye(Float32, 16, 32; S = vecT),
# opOnes(Float32, 32, 32; S = vecT), opZeros(Float32, 32, 32; S = vecT), opDiagonal(vec), opDiagonal(32, 32, vec),
# opRestriction([1, 2, 3], 32; S = vecT), opExtension([1, 2, 3], 32; S = vecT), BlockDiagonalOperator(mat, mat) and
# BlockDiagonalOperator(mat, mat; S = vecTother); with storage_type(LinearOperator(mat)) == LinearOperators.storage_type(mat)
# as the default rule.
mxlo_array(A::AbstractMatrix) = MXMatrix(A)
mxlo_array(v::AbstractVector) = MXVector(v)
include(joinpath(pkgdir(LinearOperators), "test", "gpu", "test_S_kwarg.jl"))
test_S_kwarg(arrayType = mxlo_array)

@testset "MXLO -- mirror of test/gpu/amdgpu.jl" begin
  Ah, Bh, Ch = rand(Float32, 5, 5), rand(Float32, 10, 10), rand(Float32, 20, 20)
  A, B, C = MXMatrix(Ah), MXMatrix(Bh), MXMatrix(Ch)
  M = BlockDiagonalOperator(A, B, C)            # the reference's generic constructor on plain device matrices
  vh = rand(Float32, 35)
  v = MXVector(vh)
  y = M * v
  @test y isa MXVector{Float32}
  # ... and, beyond the reference's type-only check, the numbers (Float32 GEMV: 3e-5, DESIGN.md §2)
  @test rel(host(y), BlockDiagonalOperator(Ah, Bh, Ch) * vh) <= 3e-5
  @test rel(host(transpose(M) * v), transpose(BlockDiagonalOperator(Ah, Bh, Ch)) * vh) <= 3e-5
  @test LinearOperators.storage_type(A) == LinearOperators.storage_type(adjoint(A))
  @test LinearOperators.storage_type(A) == LinearOperators.storage_type(transpose(A))
  @test LinearOperators.storage_type(A) == LinearOperators.storage_type(adjoint(A))
  @test LinearOperators.storage_type(Diagonal(v)) == typeof(v)
  @testset "MXLO S kwarg" end

# ---- the reference's allocation contract, literally: test/test_lbfgs.jl:180-218 ("LBFGS allocations") and
# test/test_lsr1.jl:88-106 on device operators. `@allocated` counts HOST (GC) bytes: a ccall closure with concrete
# captured types allocates nothing; the device side of the same contract is the mxlo_debug_counters block above.
@testset "sparse LinearOperator and sparse blocks (mirror of test/test_linop.jl:739-756, test/test_kron.jl:3-8)" begin
  using SparseArrays
  for T in (Float64, Float32), (m, n, dens) in ((7, 5, 0.5), (300, 300, 0.02), (5000, 4000, 0.002))
    A = sprand(T, m, n, dens)
    Ah, Ad = LinearOperator(A), LinearOperator(MX.MXSparseMatrixCSC(A))
    v, u, r = rand(T, n), rand(T, m), rand(T, m)
    tol = T == Float64 ? 1e-13 : 2f-6
    for (α, β) in ((one(T), zero(T)), (T(2), T(-3)), (2.0, -3.0))
      rh = copy(r); mul!(rh, Ah, v, α, β)
      rd = dev(copy(r)); mul!(rd, Ad, dev(v), α, β)
      @test norm(host(rd) - rh, Inf) <= tol * (abs(α) * maximum(abs.(A) * abs.(v); init = zero(T)) + abs(β))
    end
    @test rel(host(Ad' * dev(u)), Ah' * u) <= 100tol
    @test rel(host(transpose(Ad) * dev(u)), transpose(Ah) * u) <= 100tol
  end
  # BlockDiagonalOperator(A, B, C) with an operator, a Matrix and a sprand block (test_linop.jl:739-756)
  dinv = [0.5, 0.25, 0.125]
  B, C = rand(4, 2), sprand(2, 4, 0.5)
  D = [Diagonal(dinv) zeros(3, 2) zeros(3, 4); zeros(4, 3) B zeros(4, 4); zeros(2, 3) zeros(2, 2) Matrix(C)]
  M = BlockDiagonalOperator(opDiagonal(dev(dinv)), MXMatrix(B), MX.MXSparseMatrixCSC(C))          # the reference's per-block loop
  M1 = BlockDiagonalOperator(Float64, dev(dinv), MXMatrix(B), MX.MXSparseMatrixCSC(C))           # ONE launch
  for op in (M, M1)
    @test size(op) == (9, 9)
    x = rand(9)
    @test norm(host(op * dev(x)) - D * x) <= sqrt(eps()) * norm(D)
    @test norm(host(op' * dev(x)) - D' * x) <= sqrt(eps()) * norm(D)
    @test norm(host(transpose(op) * dev(x)) - transpose(D) * x) <= sqrt(eps()) * norm(D)
  end
  # in-place value update: Aᵀ*x sees it at once, A*x after refresh!
  A = sprand(50, 40, 0.2); As = MX.MXSparseMatrixCSC(A); op = LinearOperator(As)
  v = rand(40); y0 = host(op * dev(v))
  copyto!(As.nzval, 2 .* A.nzval); MX.refresh!(As)
  @test rel(host(op * dev(v)), 2 .* y0) <= 1e-14
end

@testset "LBFGS / LSR1 allocations (mirror of test_lbfgs.jl:180-218, test_lsr1.jl:88-106)" begin
  n, mem = 100, 20
  B = LBFGSOperator(Float64, n, MXVector{Float64}; mem = mem)
  H = InverseLBFGSOperator(Float64, n, MXVector{Float64}; mem = mem)
  BD = LBFGSOperator(Float64, n, MXVector{Float64}; mem = mem, damped = true)
  HD = InverseLBFGSOperator(Float64, n, MXVector{Float64}; mem = mem, damped = true)
  L = LSR1Operator(Float64, n, MXVector{Float64}; mem = mem)
  tmpd = dev(zeros(n))
  for _ = 1:2:n
    s = dev(rand(n)); y = dev(rand(n)); g = dev(rand(n))
    push!(B, s, y); push!(H, s, y); push!(L, s, y)
    push!(BD, s, y, tmpd)
    push!(HD, s, y, 1.0, g, tmpd)
  end
  x, y = dev(rand(n)), dev(rand(n))
  res = similar(x)
  mul!(res, B, x)  # warmup
  @test (@allocated mul!(res, B, x)) == 0
  mul!(res, H, x)  # warmup
  @test (@allocated mul!(res, H, x)) == 0
  mul!(res, L, x)  # warmup
  @test (@allocated mul!(res, L, x)) == 0
  diag!(B, res)    # warmup
  @test (@allocated diag!(B, res)) == 0
  push!(B, x, y); push!(H, x, y)   # warmup (Ref{Int32} for `accepted` must not escape)
  @test (@allocated push!(B, x, y)) == 0
  @test (@allocated push!(H, x, y)) == 0
  push!(BD, x, y, tmpd); push!(HD, x, y, 1.0, x, tmpd)
  @test (@allocated push!(BD, x, y, tmpd)) == 0
  @test (@allocated push!(HD, x, y, 1.0, x, tmpd)) == 0
end
@testset "push! contract and constructor defaults (src/lbfgs.jl:26-35,269-367, src/lsr1.jl:19,119; tests/golden/reference_semantics.json)" begin
  n = 50
  S = MXVector{Float64}
  s, y, g, tmp = dev(rand(n)), dev(rand(n) .+ 1), dev(rand(n)), dev(zeros(n))
  # defaults are the reference's CODE defaults (L-SR1: scaling = true, lsr1.jl:19)
  for op in (LBFGSOperator(Float64, n, S), InverseLBFGSOperator(Float64, n, S), LSR1Operator(Float64, n, S), LBFGSOperator(n, S))
    @test op.data.mem == 5 && op.data.scaling && !op.data.damped
  end
  @test InverseLBFGSOperator(Float64, n, S; inverse = false).inverse          # `inverse` is ignored (lbfgs.jl:115)
  @test_throws MethodError LSR1Operator(Float64, n, S; damped = true)         # LSR1Data has no such keyword
  B, H = LBFGSOperator(Float64, n, S), InverseLBFGSOperator(Float64, n, S)
  BD, HD = LBFGSOperator(Float64, n, S; damped = true), InverseLBFGSOperator(Float64, n, S; damped = true)
  @test push!(BD, s, y) === BD                                               # damped: redirected to push!(op, s, y, similar(s))
  @test_throws ErrorException push!(HD, s, y)                                # ... which refuses inverse operators (lbfgs.jl:297-299)
  @test push!(HD, s, y, 1.0, g) === HD                                       # push!(op, s, y, α, g): Bs = similar(g) (lbfgs.jl:359-367)
  @test_throws ErrorException push!(B, s, y, tmp)                            # "should be used for damped operators"
  @test_throws ErrorException push!(H, s, y, 1.0, g, tmp)
  @test_throws ErrorException push!(BD, s, y, 1.0, g, tmp)                   # forward operator, inverse-form push!
  @test_throws ErrorException push!(HD, s, y, tmp)
  @test_throws MethodError push!(LSR1Operator(Float64, n, S), s, y, tmp)     # lsr1.jl:119 is the only method
end
# The reference's diagnostics (src/utilities.jl:20-135) probe with HOST vectors (`ones(eltype(S), m)`, `rand(n)`), which a
# device closure cannot take; the bodies are otherwise storage-agnostic. Here: the same statements with the probes on
# the device (dot / norm on MXVector are mxlo_dot / mxlo_dot_c) — what tests/test_gpu_utilities.py runs through the
# Python mirror (test/test_normest.jl, test/test_linop.jl:346-372, test/test_lbfgs.jl:48-52).
function normest_dev(S, tol = -1, maxiter = 100)
  T = eltype(S)
  m, n = size(S)
  cnt = 0
  tol == -1 && (tol = Float64(eps(real(T))))
  vh = ones(T, m); vh[randn(m) .< 0] .= -1
  x = S' * dev(vh)
  e = norm(x)
  e == 0 && return e, cnt
  rmul!(x, one(T) / e)
  e_0 = zero(e)
  while abs(e - e_0) > tol * e
    e_0 = e
    Sx = S * x
    x = S' * Sx
    normx = norm(x)
    e = normx / norm(Sx)
    rmul!(x, one(T) / normx)
    cnt += 1
    cnt > maxiter && break
  end
  return e, cnt
end
function check_hermitian_dev(op)
  T = eltype(op)
  v = dev(T.(rand(size(op, 1))))
  w = op * v
  s = dot(w, w)
  t = dot(v, op * w)
  ε = eps(real(T))
  return abs(s - t) < (abs(s) + ε) * ε^(1 / 3)
end
function check_positive_definite_dev(op; semi = false)
  T = eltype(op)
  v = dev(T.(rand(size(op, 1))))
  vw = dot(v, op * v)
  ε = eps(real(T))
  imag(vw) > sqrt(ε) * abs(vw) && return false
  return semi ? (real(vw) ≥ 0) : (real(vw) > 0)
end

@testset "normest / check_* with device probes (mirror of test_normest.jl, test_linop.jl:346-372, test_lbfgs.jl:48-52)" begin
  for (nrow, ncol) in ((10, 10), (3, 5), (10, 5)), T in (Float64, ComplexF64)
    U, _ = qr(rand(T, nrow, nrow)); V, _ = qr(rand(T, ncol, ncol))
    A = Matrix(U) * T[(1 + (i - 1) / (nrow - 1)) * (i == j) for i = 1:nrow, j = 1:ncol] * Matrix(V)'
    est, _ = normest_dev(LinearOperator(MXMatrix(A)), eps(Float64), 10000)
    @test abs(est - opnorm(A, 2)) / opnorm(A, 2) <= 1e-3
  end
  n = 10
  h = dev(ComplexF64[-(-1.0)^i for i = 1:n] ./ sqrt(n))
  H = opHouseholder(h)
  op = H * opDiagonal(dev(ComplexF64.(1:n))) * H'
  @test check_positive_definite_dev(op)
  @test check_positive_definite_dev(H * opDiagonal(dev(ComplexF64.(0:(n - 1)))) * H', semi = true)
  B = LBFGSOperator(Float64, n, MXVector{Float64}; mem = 5)
  Hq = InverseLBFGSOperator(Float64, n, MXVector{Float64}; mem = 5)
  for i = 1:7
    s = dev(fill(Float64(i), n)); y = dev([Float64(i); ones(n - 1)])
    push!(B, s, y); push!(Hq, s, y)
  end
  @test check_positive_definite_dev(B) && check_positive_definite_dev(Hq)
  @test check_hermitian_dev(B) && check_hermitian_dev(Hq)
end
println("mxlo Julia smoke finished")
