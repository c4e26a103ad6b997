// complex.hip — ComplexF64 / ComplexF32 instantiation of the elementwise leaves and of opHouseholder:
//   mulSquareOpDiagonal!/mulOpDiagonal! incl. ctprod! with conj.(d) (src/special-operators.jl:125-165),
//   mulOpEye!, mulOpZeros!, prod3!'s `res .*= α` (src/special-operators.jl:36-44,102-108; src/operations.jl:13-15),
//   mulHouseholder! with LinearAlgebra.dot conjugating h (src/linalg.jl:77-83),
//   conj!/conj. of the Adjoint/Transpose/Conjugate wrapper routing (src/adjtrans.jl:90-261).
// The reference's own tests of opDiagonal / opHouseholder run on ComplexF64 (test/test_linop.jl:308-318,511-517).
//
// Arithmetic is Julia's, component by component, nothing fused (-ffp-contract=off):
//   Complex*Complex : (zr*wr - zi*wi, zr*wi + zi*wr)        Real*Complex : (x*wr, x*wi)       (base/complex.jl)
// A complex element is 16 bytes (ComplexF64: one element per 16-byte lane access) or 8 bytes (ComplexF32: two), so
// the same streaming kernel (stream_kernels.h: map_kernel) runs them at the rate of the real leaves:
// complex opDiagonal at n = 5e7 moves the same 2.4 GB as the real one at n = 1e8.
// Caller scalars arrive as (re, im) pairs plus MXLO_ALPHA_REAL / MXLO_BETA_REAL (a Real scalar multiplies
// componentwise, which differs from Complex(x, 0) in signed zeros / non-finite values) and the usual width flags.
#include "common.h"
#include "stream_kernels.h"
#include "complex_scalars.h"

using namespace mxlo;

namespace mxlo {
int32_t finalize_and_reduce(mxlo_ctx *ctx, int ncols, int nblocks, double *out_dev);
int32_t allreduce_hook(mxlo_ctx *ctx, double *dev, int64_t count);
template <typename R>
int32_t kron_planes(mxlo_ctx *ctx, R *rr, R *ri, const R *Ar, const R *Ai, int64_t am, int64_t an, int64_t lda,
                    bool trans_a, double sign_ai, const R *Br, const R *Bi, int64_t bp, int64_t bq, int64_t ldb,
                    bool trans_b, double sign_bi, const R *xr, const R *xi, R *utr, R *uti);   // dense.hip
template <typename R>
int32_t kron_planes3(mxlo_ctx *ctx, R *rr, R *ri, const R *Ar, const R *Ai, const R *As, int64_t am, int64_t an, int64_t lda,
                     bool trans_a, double sign_ai, const R *Br, const R *Bi, const R *Bs, int64_t bp, int64_t bq, int64_t ldb,
                     bool trans_b, double sign_bi, const R *xr, const R *xi, const R *xd, const R *xs, R *work,
                     const R **kout);   // dense.hip
template <typename R>
int32_t plane_sum(mxlo_ctx *ctx, R *out, const R *a, const R *b, int64_t rows, int64_t cols, int64_t ld, double sign);   // dense.hip
}  // namespace mxlo

namespace {


// res = (a*d')*v (+ b*res), d' = d or conj(d)
template <typename R, typename RA, typename RB, bool BETA0, bool CONJD>
struct CDiagOp {
  Sc<RA> a;
  Sc<RB> b;
  __device__ void init() {}
  __device__ C<R> operator()(C<R> d, C<R> v, C<R> r) const {
    if constexpr (CONJD) d.im = -d.im;                      // conj.(d): exact
    RA tr, ti;
    a.mul(d, tr, ti);                                       // α*d
    const RA vr = (RA)v.re, vi = (RA)v.im;
    const RA ur = (tr * vr) - (ti * vi), ui = (tr * vi) + (ti * vr);   // (α*d)*v
    return cfin<R, RA, RB, BETA0>(ur, ui, b.re, b.im, b.real, r);
  }
};
// res = a*v (+ b*res)
template <typename R, typename RA, typename RB, bool BETA0>
struct CAxpbyOp {
  Sc<RA> a;
  Sc<RB> b;
  __device__ void init() {}
  __device__ C<R> operator()(C<R> v, C<R>, C<R> r) const {
    RA tr, ti;
    a.mul(v, tr, ti);
    return cfin<R, RA, RB, BETA0>(tr, ti, b.re, b.im, b.real, r);
  }
};
// res = res*s   (res .*= β of mulOpZeros!, res .*= α of prod3!): Complex*Complex with z = res, w = s
template <typename R, typename RS>
struct CScaleOp {
  Sc<RS> s;
  __device__ void init() {}
  __device__ C<R> operator()(C<R>, C<R>, C<R> r) const {
    const RS rr = (RS)r.re, ri = (RS)r.im;
    if (s.real) return C<R>((R)(rr * s.re), (R)(ri * s.re));
    return C<R>((R)((rr * s.re) - (ri * s.im)), (R)((rr * s.im) + (ri * s.re)));
  }
};
template <typename R>
struct CFillOp {
  C<R> c;
  __device__ void init() {}
  __device__ C<R> operator()(C<R>, C<R>, C<R>) const { return c; }
};
template <typename R>
struct CConjOp {
  __device__ void init() {}
  __device__ C<R> operator()(C<R> v, C<R>, C<R>) const { return C<R>(v.re, -v.im); }
};
// res = a*(v - c*h) (+ b*res), c = 2*dot(h, v) (dot conjugates h) read from two device doubles
template <typename R, typename RA, typename RB, bool BETA0>
struct CHouseholderOp {
  Sc<RA> a;
  Sc<RB> b;
  const double *dot;
  R cr, ci;
  __device__ void init() {
    cr = (R)2 * (R)dot[0];     // 2 * dot(h, v): Int * Complex is componentwise, dot is a Complex{R}
    ci = (R)2 * (R)dot[1];
  }
  __device__ C<R> operator()(C<R> h, C<R> v, C<R> r) const {
    const R pr = (cr * h.re) - (ci * h.im), pi = (cr * h.im) + (ci * h.re);   // c .* h
    const C<R> inner(v.re - pr, v.im - pi);                                    // v .- c .* h
    RA tr, ti;
    a.mul(inner, tr, ti);
    return cfin<R, RA, RB, BETA0>(tr, ti, b.re, b.im, b.real, r);
  }
};

// dot(h, v) = sum conj(h_i) v_i -> partial columns 0 (re) and 1 (im); f64 accumulation, fixed-order finalize
template <typename R>
__global__ void __launch_bounds__(kBlock)
cdotc_kernel(const C<R> *__restrict__ h, const C<R> *__restrict__ v, int64_t n, bool vec_ok,
             double *__restrict__ partials) {
  constexpr int VEC = Vec16<C<R>>::N;
  using V = typename Vec16<C<R>>::type;
  double ar = 0.0, ai = 0.0;
  const int64_t nvec = vec_ok ? n / VEC : 0;   // misaligned ComplexF32 views: element accesses only
  const V *hv = reinterpret_cast<const V *>(h), *vv = reinterpret_cast<const V *>(v);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  auto acc = [&](const V &a, const V &b) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const double hr = (double)a[2 * e], hi = (double)a[2 * e + 1], vr = (double)b[2 * e], vi = (double)b[2 * e + 1];
      ar = fma(hr, vr, ar);
      ar = fma(hi, vi, ar);
      ai = fma(hr, vi, ai);
      ai = fma(-hi, vr, ai);
    }
  };
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    V a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = __builtin_nontemporal_load(hv + i + u * stride);
      b[u] = __builtin_nontemporal_load(vv + i + u * stride);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc(a[u], b[u]);
  }
  for (; i < nvec; i += stride) acc(hv[i], vv[i]);
  for (int64_t k = nvec * VEC + (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride) {
    const double hr = (double)h[k].re, hi = (double)h[k].im, vr = (double)v[k].re, vi = (double)v[k].im;
    ar = fma(hr, vr, ar);
    ar = fma(hi, vi, ar);
    ai = fma(hr, vi, ai);
    ai = fma(-hi, vr, ai);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ar += __shfl_down(ar, off, 64);
    ai += __shfl_down(ai, off, 64);
  }
  __shared__ double lds[2][kBlock / kWave];
  if ((threadIdx.x & 63) == 0) {
    lds[0][threadIdx.x >> 6] = ar;
    lds[1][threadIdx.x >> 6] = ai;
  }
  __syncthreads();
  if (threadIdx.x < 2)
    partials[(int64_t)threadIdx.x * kMaxRedBlocks + blockIdx.x] =
        (lds[threadIdx.x][0] + lds[threadIdx.x][1]) + (lds[threadIdx.x][2] + lds[threadIdx.x][3]);
}

template <typename R>
int32_t cdotc(mxlo_ctx *ctx, const C<R> *h, const C<R> *v, int64_t n, double *out2) {
  if (n <= 0) {
    MXLO_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(double), ctx->stream));
  } else {
    // 16-byte vector loads need 16-byte aligned operands (always true for ComplexF64; a ComplexF32 view may start
    // at an odd element)
    const bool al = ((((uintptr_t)h) | ((uintptr_t)v)) & 15u) == 0;
    const int grid = grid_for(ctx, n, kBlock * 4, ctx->tune.red_blocks_per_cu);
    hipLaunchKernelGGL((cdotc_kernel<R>), dim3(grid), dim3(kBlock), 0, ctx->stream, h, v, n, al, ctx->partials);
    MXLO_LAUNCH_CHECK();
    MXLO_TRY(finalize_and_reduce(ctx, 2, grid, out2));
  }
  return allreduce_hook(ctx, out2, 2);
}

template <typename R>
int32_t cfill(mxlo_ctx *ctx, C<R> *p, int64_t n, C<R> c) {
  if (n <= 0) return MXLO_OK;
  if (c.re == R(0) && c.im == R(0) && !std::signbit(c.re) && !std::signbit(c.im)) {
    MXLO_HIP(hipMemsetAsync(p, 0, sizeof(C<R>) * (size_t)n, ctx->stream));
    return MXLO_OK;
  }
  return launch_map<C<R>, 0, false, false>(ctx, p, (const C<R> *)nullptr, (const C<R> *)nullptr, n, CFillOp<R>{c});
}

template <typename R>
int32_t cscale(mxlo_ctx *ctx, C<R> *res, int64_t n, double sre, double sim, bool s_real, bool s64) {
  if (s64)
    return launch_map<C<R>, 0, true, false>(ctx, res, (const C<R> *)nullptr, (const C<R> *)nullptr, n,
                                            CScaleOp<R, double>{{sre, sim, s_real}});
  return launch_map<C<R>, 0, true, false>(ctx, res, (const C<R> *)nullptr, (const C<R> *)nullptr, n,
                                          CScaleOp<R, float>{{(float)sre, (float)sim, s_real}});
}

template <typename R>
int32_t cdiag(mxlo_ctx *ctx, C<R> *res, const C<R> *d, const C<R> *v, int64_t n_min, int64_t nrow, const ScalArgs &s,
              bool conj_d) {
  MXLO_TRY((dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
    const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
    if (conj_d) return launch_map<C<R>, 2, !B0, false>(ctx, res, d, v, n_min, CDiagOp<R, RA, RB, B0, true>{a, b});
    return launch_map<C<R>, 2, !B0, false>(ctx, res, d, v, n_min, CDiagOp<R, RA, RB, B0, false>{a, b});
  })));
  return cfill<R>(ctx, res + n_min, nrow - n_min, C<R>());   // res[n_min+1:end] .= 0 regardless of β (:150)
}

template <typename R>
int32_t ceye(mxlo_ctx *ctx, C<R> *res, const C<R> *v, int64_t n_min, int64_t nrow, const ScalArgs &s, int32_t flags) {
  MXLO_TRY((dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
    const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
    return launch_map<C<R>, 1, !B0, false>(ctx, res, v, (const C<R> *)nullptr, n_min, CAxpbyOp<R, RA, RB, B0>{a, b});
  })));
  const int64_t ntail = nrow - n_min;
  if (ntail <= 0) return MXLO_OK;
  if (s.bre == 0 && s.bim == 0) return cfill<R>(ctx, res + n_min, ntail, C<R>());
  if (flags & MXLO_TAIL_BETA) return cfill<R>(ctx, res + n_min, ntail, C<R>((R)s.bre, (R)s.bim));   // `.= β` (:42)
  return cscale<R>(ctx, res + n_min, ntail, s.bre, s.bim, s.b_real, s.b64);
}

template <typename R>
int32_t chouse(mxlo_ctx *ctx, C<R> *res, const C<R> *h, const C<R> *v, int64_t n, const ScalArgs &s) {
  double *dot = ctx->scalars;   // slots 0, 1
  MXLO_TRY(cdotc<R>(ctx, h, v, n, dot));
  const bool rev = ctx->tune.house_reverse != 0;
  return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    CHouseholderOp<R, RA, RB, B0> op{{(RA)s.are, (RA)s.aim, s.a_real}, {(RB)s.bre, (RB)s.bim, s.b_real}, dot, R(0), R(0)};
    if (rev) return launch_map<C<R>, 2, !B0, true>(ctx, res, h, v, n, op);
    return launch_map<C<R>, 2, !B0, false>(ctx, res, h, v, n, op);
  });
}

// ---- dense complex leaves: LinearOperator(M) (mul!(res, op(M), v, α, β), src/constructors.jl:19-29) and opHermitian
// (src/linalg.jl:97-127) on ComplexF64 / ComplexF32 — the element types the reference's own "Hermitian" and "Transpose
// and adjoint" testsets use (test/test_linop.jl:360-420). Reductions accumulate in f64 with explicit fma (like every
// reduction of this library), results are rounded once to Complex{R}, then res = α*t (+ β*res) component by component.
// These are correctness-first instantiations (one wave per column / one lane per row, fixed-order partials): they are
// not a BASELINE configuration and not tuned like the real-valued GEMV / strip kernels of dense.hip.
enum { CG_N = 0, CG_T = 1, CG_C = 2, CG_J = 3 };   // M*v, transpose(M)*v, M'*v, conj(M)*v

// column sums: out[j] = sum_{i >= i0(j)} op(M[i,j]) * v[i],  op = conj iff CONJ;  LOWER: strict lower triangle (i > j).
// One wave per column, 4 rows in flight per lane. RAW: store the sums as Complex{R}; else apply α, β into res.
template <typename R, typename RA, typename RB, bool BETA0, bool CONJ, bool LOWER, bool RAW, bool VEC = false, bool NT = true>
__global__ void __launch_bounds__(kBlock)
cgemv_cols_kernel(C<R> *__restrict__ res, const C<R> *__restrict__ M, int64_t m, int64_t n, int64_t ld,
                  const C<R> *__restrict__ v, Sc<RA> a, Sc<RB> b) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t j = wave; j < n; j += nwaves) {
    const C<R> *col = M + j * ld;
    double ar = 0.0, ai = 0.0, br_ = 0.0, bi_ = 0.0;
    auto acc = [&](const C<R> &e, const C<R> &x, double &sr, double &si) {
      const double er = (double)e.re, ei = CONJ ? -(double)e.im : (double)e.im, xr = (double)x.re, xi = (double)x.im;
      sr = fma(er, xr, sr);
      sr = fma(-ei, xi, sr);
      si = fma(er, xi, si);
      si = fma(ei, xr, si);
    };
    int64_t i = (LOWER ? j + 1 : 0) + lane;
    if constexpr (VEC && !LOWER) {
      // aligned full columns (round 5): 16-byte NONTEMPORAL loads of M (one ComplexF64 / two ComplexF32 per lane; M is
      // streamed once — the hint alone is worth 6.3 -> 7.1 TB/s on a read-only pass, profiles/r05_tune_read.txt), 4 in flight
      constexpr int VR = 16 / (int)sizeof(C<R>);
      typedef R VV __attribute__((ext_vector_type(2 * VR)));
      const int64_t mp = m / VR;                      // m % VR == 0 (cgemv_cols)
      const VV *cp = reinterpret_cast<const VV *>(col);
      const VV *xp = reinterpret_cast<const VV *>(v);
      auto accv = [&](const VV &e, const VV &x, double &sr, double &si) {
#pragma unroll
        for (int q = 0; q < VR; ++q) {
          const double er = (double)e[2 * q], ei = CONJ ? -(double)e[2 * q + 1] : (double)e[2 * q + 1];
          const double xr = (double)x[2 * q], xi = (double)x[2 * q + 1];
          sr = fma(er, xr, sr);
          sr = fma(-ei, xi, sr);
          si = fma(er, xi, si);
          si = fma(ei, xr, si);
        }
      };
      int64_t p = lane;
      for (; p + 192 < mp; p += 256) {
        auto ldm = [&](const VV *q) { return NT ? __builtin_nontemporal_load(q) : *q; };
        const VV e0 = ldm(cp + p), e1 = ldm(cp + p + 64), e2 = ldm(cp + p + 128), e3 = ldm(cp + p + 192);
        const VV x0 = xp[p], x1 = xp[p + 64], x2 = xp[p + 128], x3 = xp[p + 192];
        accv(e0, x0, ar, ai);
        accv(e1, x1, br_, bi_);
        accv(e2, x2, ar, ai);
        accv(e3, x3, br_, bi_);
      }
      for (; p < mp; p += 64) accv(cp[p], xp[p], ar, ai);
      i = m;                                          // nothing left for the element loops below
    }
    for (; i + 192 < m; i += 256) {
      const C<R> e0 = col[i], e1 = col[i + 64], e2 = col[i + 128], e3 = col[i + 192];
      const C<R> x0 = v[i], x1 = v[i + 64], x2 = v[i + 128], x3 = v[i + 192];
      acc(e0, x0, ar, ai);
      acc(e1, x1, br_, bi_);
      acc(e2, x2, ar, ai);
      acc(e3, x3, br_, bi_);
    }
    for (; i < m; i += 64) acc(col[i], v[i], ar, ai);
    ar += br_;
    ai += bi_;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ar += __shfl_down(ar, off, 64);
      ai += __shfl_down(ai, off, 64);
    }
    if (lane == 0) {
      const C<R> t((R)ar, (R)ai);
      if constexpr (RAW) {
        res[j] = t;
      } else {
        RA tr, ti;
        a.mul(t, tr, ti);
        res[j] = cfin<R, RA, RB, BETA0>(tr, ti, b.re, b.im, b.real, BETA0 ? C<R>() : res[j]);
      }
    }
  }
}

// row partials: part[chunk][i] = sum_{j in chunk, (LOWER: j < i)} op(M[i,j]) * v[j]; one lane per row, 4 columns in flight
template <typename R, bool CONJ, bool LOWER>
__global__ void __launch_bounds__(kBlock)
cgemv_rows_partial_kernel(double *__restrict__ part, const C<R> *__restrict__ M, int64_t m, int64_t n, int64_t ld,
                          const C<R> *__restrict__ v, int64_t cols_per_chunk) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const int64_t j0 = (int64_t)blockIdx.y * cols_per_chunk;
  int64_t j1 = j0 + cols_per_chunk;
  if (j1 > n) j1 = n;
  if (LOWER && j1 > i) j1 = i;                        // strict lower triangle: columns j < i
  double ar = 0.0, ai = 0.0, br_ = 0.0, bi_ = 0.0;
  auto acc = [&](const C<R> &e, const C<R> &x, double &sr, double &si) {
    const double er = (double)e.re, ei = CONJ ? -(double)e.im : (double)e.im, xr = (double)x.re, xi = (double)x.im;
    sr = fma(er, xr, sr);
    sr = fma(-ei, xi, sr);
    si = fma(er, xi, si);
    si = fma(ei, xr, si);
  };
  int64_t j = j0;
  for (; j + 7 < j1; j += 8) {                        // 8 columns (8 x 1 KiB per wave for ComplexF64) in flight
    C<R> e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = M[i + (j + u) * ld];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      acc(e[u], v[j + u], ar, ai);
      acc(e[u + 1], v[j + u + 1], br_, bi_);
    }
  }
  for (; j < j1; ++j) acc(M[i + j * ld], v[j], ar, ai);
  double *p = part + ((int64_t)blockIdx.y * m + i) * 2;
  p[0] = ar + br_;
  p[1] = ai + bi_;
}

// fixed-order sum of the chunk partials of a row, then RAW store or the α, β epilogue. 32 rows per workgroup, 8 lanes per
// row: lane `sub` adds chunks sub, sub+8, ... four loads at a time; the 8 sub-sums are combined in a fixed order
// (m/32 workgroups instead of m/256, and 8 x 4 partial loads of a row in flight instead of one).
template <typename R, typename RA, typename RB, bool BETA0, bool RAW>
__global__ void __launch_bounds__(kBlock)
cgemv_rows_finish_kernel(C<R> *__restrict__ res, const double *__restrict__ part, int64_t m, int nchunks, Sc<RA> a,
                         Sc<RB> b) {
  const int r = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + r;
  __shared__ double sre[8][32], sim[8][32];
  double sr = 0.0, si = 0.0;
  if (i < m) {
    for (int base = sub; base < nchunks; base += 32) {
      f64x2 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = base + 8 * u;
        x[u] = c < nchunks ? *reinterpret_cast<const f64x2 *>(part + ((int64_t)c * m + i) * 2) : f64x2{0.0, 0.0};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sr += x[u][0];
        si += x[u][1];
      }
    }
  }
  sre[sub][r] = sr;
  sim[sub][r] = si;
  __syncthreads();
  if (sub != 0 || i >= m) return;
  double tr_ = 0.0, ti_ = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    tr_ += sre[q][r];
    ti_ += sim[q][r];
  }
  const C<R> t((R)tr_, (R)ti_);
  if constexpr (RAW) {
    res[i] = t;
  } else {
    RA tr, ti;
    a.mul(t, tr, ti);
    res[i] = cfin<R, RA, RB, BETA0>(tr, ti, b.re, b.im, b.real, BETA0 ? C<R>() : res[i]);
  }
}

template <typename R, bool LOWER, bool RAW>
int32_t cgemv_rows(mxlo_ctx *ctx, C<R> *res, const C<R> *M, int64_t m, int64_t n, int64_t ld, const C<R> *v, bool conj,
                   const ScalArgs &s) {
  const int64_t cap = (int64_t)kMaxRedCols * kMaxRedBlocks / 2;        // complex partials in ctx->partials
  MXLO_REQUIRE(m <= cap, MXLO_ESHAPE, "complex gemv: m = %lld exceeds the partial workspace (%lld rows)", (long long)m,
               (long long)cap);
  const int64_t row_blocks = (m + kBlock - 1) / kBlock;
  int64_t nchunks = (n + 63) / 64;
  const int64_t want = (int64_t)ctx->num_cu * 8 / (row_blocks > 0 ? row_blocks : 1) + 1;
  if (nchunks > want) nchunks = want;
  if (nchunks > cap / (m > 0 ? m : 1)) nchunks = cap / (m > 0 ? m : 1);
  if (nchunks > 65535) nchunks = 65535;
  if (nchunks < 1) nchunks = 1;
  const int64_t cpc = n > 0 ? (n + nchunks - 1) / nchunks : 1;
  nchunks = n > 0 ? (n + cpc - 1) / cpc : 1;
  dim3 grid((unsigned)row_blocks, (unsigned)nchunks);
  if (conj)
    hipLaunchKernelGGL((cgemv_rows_partial_kernel<R, true, LOWER>), grid, dim3(kBlock), 0, ctx->stream, ctx->partials, M, m,
                       n, ld, v, cpc);
  else
    hipLaunchKernelGGL((cgemv_rows_partial_kernel<R, false, LOWER>), grid, dim3(kBlock), 0, ctx->stream, ctx->partials, M, m,
                       n, ld, v, cpc);
  MXLO_LAUNCH_CHECK();
  return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
    const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
    hipLaunchKernelGGL((cgemv_rows_finish_kernel<R, RA, RB, B0, RAW>), dim3((unsigned)((m + 31) / 32)), dim3(kBlock), 0,
                       ctx->stream, res, ctx->partials, m, (int)nchunks, a, b);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

template <typename R, bool LOWER, bool RAW>
int32_t cgemv_cols(mxlo_ctx *ctx, C<R> *res, const C<R> *M, int64_t m, int64_t n, int64_t ld, const C<R> *v, bool conj,
                   const ScalArgs &s) {
  int64_t blocks = (n + 3) / 4;
  const int64_t cap = (int64_t)ctx->num_cu * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
    const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
    constexpr int VR = 16 / (int)sizeof(C<R>);
    const bool vec = !LOWER && m >= 256 * VR && ((((uintptr_t)M | (uintptr_t)v) & 15u) == 0) && ld % VR == 0 && m % VR == 0;
    if constexpr (!LOWER) {
      if (vec) {
        const bool nt = (int64_t)sizeof(C<R>) * m * n >= ctx->tune.nt_min_bytes;   // cache-sized matrices keep default loads
#define CCOLS(CJ_, NT_)                                                                                                \
  hipLaunchKernelGGL((cgemv_cols_kernel<R, RA, RB, B0, CJ_, LOWER, RAW, true, NT_>), dim3((unsigned)blocks), dim3(kBlock), 0, \
                     ctx->stream, res, M, m, n, ld, v, a, b)
        if (conj) { if (nt) CCOLS(true, true); else CCOLS(true, false); }
        else { if (nt) CCOLS(false, true); else CCOLS(false, false); }
#undef CCOLS
        MXLO_LAUNCH_CHECK();
        return MXLO_OK;
      }
    }
    if (conj)
      hipLaunchKernelGGL((cgemv_cols_kernel<R, RA, RB, B0, true, LOWER, RAW>), dim3((unsigned)blocks), dim3(kBlock), 0,
                         ctx->stream, res, M, m, n, ld, v, a, b);
    else
      hipLaunchKernelGGL((cgemv_cols_kernel<R, RA, RB, B0, false, LOWER, RAW>), dim3((unsigned)blocks), dim3(kBlock), 0,
                         ctx->stream, res, M, m, n, ld, v, a, b);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

// N mode, ROW BANDS (round 5; the real form is dense.hip: gemv_n_rows_kernel): a 512-thread workgroup owns RBAND rows of a
// complex matrix across ALL its columns — LPR lanes cover the band's piece of one column with 16-byte loads (one ComplexF64
// or two ComplexF32 per lane), the other lanes take other columns, 8 loads in flight per lane — so the sum over the columns
// never leaves the workgroup: one launch, no partial workspace, no finish launch; fixed-order sums through LDS.
constexpr int kCRowsBlock = 512;
template <typename R, typename RA, typename RB, bool BETA0, bool CONJ, int RBAND, bool NT = true>
__global__ void __launch_bounds__(kCRowsBlock)
cgemv_rows_band_kernel(C<R> *__restrict__ res, const C<R> *__restrict__ M, int64_t m, int64_t n, int64_t ld,
                       const C<R> *__restrict__ v, Sc<RA> a, Sc<RB> b) {
  constexpr int VR = 16 / (int)sizeof(C<R>);          // complex elements per 16-byte load: 1 (ComplexF64) or 2 (ComplexF32)
  typedef R VV __attribute__((ext_vector_type(2 * VR)));
  constexpr int LPR = RBAND / VR, NCL = kCRowsBlock / LPR, U = 8;
  static_assert(RBAND % VR == 0 && kCRowsBlock % LPR == 0, "bad row band");
  __shared__ double sre[NCL][RBAND], sim[NCL][RBAND];
  const int tid = threadIdx.x, seg = tid % LPR, cl = tid / LPR;
  const int64_t row = (int64_t)blockIdx.x * RBAND + (int64_t)seg * VR;
  double sr[VR], si[VR];
#pragma unroll
  for (int e = 0; e < VR; ++e) sr[e] = si[e] = 0.0;
  auto acc = [&](const VV &av, const C<R> &x) {
    const double xr = (double)x.re, xi = (double)x.im;
#pragma unroll
    for (int e = 0; e < VR; ++e) {
      const double er = (double)av[2 * e], ei = CONJ ? -(double)av[2 * e + 1] : (double)av[2 * e + 1];
      sr[e] = fma(er, xr, sr[e]);
      sr[e] = fma(-ei, xi, sr[e]);
      si[e] = fma(er, xi, si[e]);
      si[e] = fma(ei, xr, si[e]);
    }
  };
  if (row < m) {                                       // m % VR == 0 (cgemv_any): a vector is wholly inside or outside
    const C<R> *base = M + row;
    int64_t j = cl;
    for (; j + (int64_t)(U - 1) * NCL < n; j += (int64_t)U * NCL) {
      VV av[U];
      C<R> x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const VV *q = reinterpret_cast<const VV *>(base + (j + (int64_t)u * NCL) * ld);
        av[u] = NT ? __builtin_nontemporal_load(q) : *q;
        x[u] = v[j + (int64_t)u * NCL];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc(av[u], x[u]);
    }
    for (; j < n; j += NCL) acc(*reinterpret_cast<const VV *>(base + j * ld), v[j]);
  }
#pragma unroll
  for (int e = 0; e < VR; ++e) {
    sre[cl][seg * VR + e] = sr[e];
    sim[cl][seg * VR + e] = si[e];
  }
  __syncthreads();
  constexpr int Q = RBAND * 8 <= kCRowsBlock ? 8 : kCRowsBlock / RBAND;
  static_assert(Q >= 1 && NCL % Q == 0 && RBAND * Q <= kCRowsBlock, "bad finish shape");
  double pr = 0.0, pi = 0.0;
  const int r = tid % RBAND, q = tid / RBAND;
  if (tid < RBAND * Q) {
    for (int c = q; c < NCL; c += Q) {
      pr += sre[c][r];
      pi += sim[c][r];
    }
  }
  __syncthreads();
  if (tid < RBAND * Q) {
    sre[q][r] = pr;
    sim[q][r] = pi;
  }
  __syncthreads();
  if (tid < RBAND) {
    const int64_t i = (int64_t)blockIdx.x * RBAND + tid;
    if (i < m) {
      double tr_ = 0.0, ti_ = 0.0;
#pragma unroll
      for (int qq = 0; qq < Q; ++qq) {
        tr_ += sre[qq][tid];
        ti_ += sim[qq][tid];
      }
      const C<R> t((R)tr_, (R)ti_);
      RA tr, ti;
      a.mul(t, tr, ti);
      res[i] = cfin<R, RA, RB, BETA0>(tr, ti, b.re, b.im, b.real, BETA0 ? C<R>() : res[i]);
    }
  }
}

// band height for an m x n complex operand (0: the column-chunk schedule); policy and thresholds as dense.hip: gemv_rows_band
template <typename R>
int cgemv_rows_band(const mxlo_ctx *ctx, const C<R> *M, int64_t m, int64_t n, int64_t ld) {
  constexpr int VR = 16 / (int)sizeof(C<R>);
  const bool vec = (((uintptr_t)M & 15u) == 0) && ld % VR == 0 && m % VR == 0;
  if (ctx->tune.gemv_n_rows != 1 || !vec || n < 1024) return 0;
  int rb = 0;
  if (m >= (int64_t)32 * VR * ctx->num_cu) rb = 32 * VR;
  else if (m >= (int64_t)16 * VR * ctx->num_cu) rb = 16 * VR;
  else if (m >= (int64_t)8 * VR * ctx->num_cu) rb = 8 * VR;
  if (rb == 16 * VR && n >= 16384) rb = 0;
  if (rb == 8 * VR && n > 16384) rb = 0;
  return rb;
}

template <typename R>
int32_t cgemv_any(mxlo_ctx *ctx, C<R> *res, const C<R> *M, int64_t m, int64_t n, int64_t ld, const C<R> *v, int mode,
                  const ScalArgs &s, int32_t flags) {
  const bool rows = mode == CG_N || mode == CG_J, conj = mode == CG_C || mode == CG_J;
  const int64_t nres = rows ? m : n, nin = rows ? n : m;
  if (nres == 0) return MXLO_OK;
  if (nin == 0) {   // empty sum: res = β*res (or 0)
    if (s.bre == 0 && s.bim == 0) return cfill<R>(ctx, res, nres, C<R>());
    return cscale<R>(ctx, res, nres, s.bre, s.bim, s.b_real, s.b64);
  }
  if (rows) {
    if (const int rb = cgemv_rows_band<R>(ctx, M, m, n, ld); rb != 0) {
      constexpr int VR = 16 / (int)sizeof(C<R>);
      return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
        const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
        const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
#define CROWS(CJ_, RB_)                                                                                               \
  if (nt) hipLaunchKernelGGL((cgemv_rows_band_kernel<R, RA, RB, B0, CJ_, RB_, true>), dim3((unsigned)((m + (RB_) - 1) / (RB_))), \
                             dim3(kCRowsBlock), 0, ctx->stream, res, M, m, n, ld, v, a, b);                             \
  else hipLaunchKernelGGL((cgemv_rows_band_kernel<R, RA, RB, B0, CJ_, RB_, false>), dim3((unsigned)((m + (RB_) - 1) / (RB_))), \
                          dim3(kCRowsBlock), 0, ctx->stream, res, M, m, n, ld, v, a, b)
        const bool nt = (int64_t)sizeof(C<R>) * m * n >= ctx->tune.nt_min_bytes;   // cache-sized matrices keep default loads
        if (conj) {
          if (rb == 32 * VR) { CROWS(true, 32 * VR); }
          else if (rb == 16 * VR) { CROWS(true, 16 * VR); }
          else { CROWS(true, 8 * VR); }
        } else {
          if (rb == 32 * VR) { CROWS(false, 32 * VR); }
          else if (rb == 16 * VR) { CROWS(false, 16 * VR); }
          else { CROWS(false, 8 * VR); }
        }
#undef CROWS
        MXLO_LAUNCH_CHECK();
        return MXLO_OK;
      });
    }
    return cgemv_rows<R, false, false>(ctx, res, M, m, n, ld, v, conj, s);
  }
  return cgemv_cols<R, false, false>(ctx, res, M, m, n, ld, v, conj, s);
}

// res_i = α*((d_i*v_i + t1_i) + t2_i) (+ β*res_i): the sum of src/linalg.jl:99-101 in Complex{R}, in that order
template <typename R, typename RA, typename RB, bool BETA0, bool DREAL>
__global__ void __launch_bounds__(kBlock)
cherm_final_kernel(C<R> *__restrict__ res, const void *__restrict__ dptr, const C<R> *__restrict__ v,
                   const C<R> *__restrict__ t1, const C<R> *__restrict__ t2, int64_t n, Sc<RA> a, Sc<RB> b) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    R pr, pi;
    if constexpr (DREAL) {
      const R d = static_cast<const R *>(dptr)[i];
      pr = d * v[i].re;
      pi = d * v[i].im;
    } else {
      const C<R> d = static_cast<const C<R> *>(dptr)[i];
      pr = (d.re * v[i].re) - (d.im * v[i].im);
      pi = (d.re * v[i].im) + (d.im * v[i].re);
    }
    const C<R> inner((pr + t1[i].re) + t2[i].re, (pi + t1[i].im) + t2[i].im);
    RA tr, ti;
    a.mul(inner, tr, ti);
    res[i] = cfin<R, RA, RB, BETA0>(tr, ti, b.re, b.im, b.real, BETA0 ? C<R>() : res[i]);
  }
}

// ---- complex opHermitian, ONE pass over the strict lower triangle --------------------------------------------------
// The decomposition of the real kernel (dense.hip: herm_strip_body) on 16-byte complex lanes: a lane holds RPL = 1
// (ComplexF64) or 2 (ComplexF32) rows of a column, 128 lanes span a row group of HR = 128*RPL rows, a tile is HR x 32,
// a workgroup owns a strip of CT consecutive tiles of one row group and reads it once. Per tile every wave produces
//   cols : Pcol[2G + half][32J + c] = sum over its 64*RPL rows of conj(L[r][c]) * v[r]      (part of L'*v = (v'*L)')
// by a halving butterfly over the 64 lanes (16 columns x {re, im} = 32 values -> one per lane pair), and keeps
//   rows : Prow[s][HR*G + r]       = sum over the strip's columns of L[r][c] * v[c]         (part of L*v)
// in registers until the strip ends. f64 accumulation for both element types; a finish kernel adds the partials in a
// fixed order and applies res = α*((d.*v + L*v) + L'*v) (+ β*res) (src/linalg.jl:97-103) in Complex{R}.
constexpr int CHC = 32;            // tile columns
template <typename R>
struct CHermCfg {
  static constexpr int RPL = 16 / (int)sizeof(C<R>);
  static constexpr int HR = 128 * RPL;
  static constexpr int DT = HR / CHC;     // tiles across a diagonal block: 4 (ComplexF64) or 8 (ComplexF32)
};

// EDGE / mode as in the real kernel: EDGE = false — strips of full row groups strictly below the diagonal, 16-byte
// loads, no masks; EDGE = true — mode 0: the same strips, masked (A not 16-byte aligned); mode 1: strips of the ragged
// last row group; mode 2 (CT = 1): one tile of a diagonal block per workgroup.
// DSEL (EDGE = false, CT = 1): a tile of the diagonal block of a FULL row group of an aligned matrix — unmasked loads,
// then a select zeroes what is at or above the diagonal (dense.hip: herm_strip_body).
template <typename R, int CT, bool EDGE, bool DSEL = false, bool NT = true>
__device__ __forceinline__ void
cherm_strip_body(const C<R> *__restrict__ A, int64_t lda, const C<R> *__restrict__ v, int64_t n,
                 double *__restrict__ Prow, double *__restrict__ Pcol, int64_t ng, int qint, int mode, int64_t t,
                 double (*rowred)[CHermCfg<R>::HR][2], int colmajor_g = 0) {
  constexpr int RPL = CHermCfg<R>::RPL, HR = CHermCfg<R>::HR, DT = CHermCfg<R>::DT;
  using V = typename Vec16<C<R>>::type;      // f64x2 / f32x4: RPL complex elements
  static_assert(!DSEL || (!EDGE && CT == 1), "DSEL: one unmasked-load tile of a diagonal block");
  int64_t G, tile0, slot;                    // row group, first column tile, row-partial slot
  if constexpr (DSEL) {                      // diagonal block of row group t/DT, tile t%DT
    G = t / DT;
    tile0 = DT * G + t % DT;
    slot = (int64_t)qint * G + t % DT;
  } else if (!EDGE || mode == 0) {           // triangular enumeration u = G'(G'+1)/2 + r, r <= G'; G = G'+1
    constexpr int Q = DT / CT;
    // colmajor_g = g > 0 (interior strips of g full row groups; tune key herm_order, dense.hip: herm_strip_body): the pairs are
    // walked column block by column block, so that workgroups running together read contiguous runs down the same columns
    const int64_t u0 = t / Q;
    const int64_t u = colmajor_g > 0 ? (int64_t)colmajor_g * (colmajor_g - 1) / 2 - 1 - u0 : u0;
    int64_t Gp = (int64_t)((sqrt(8.0 * (double)u + 1.0) - 1.0) * 0.5);
    while (Gp * (Gp + 1) / 2 > u) --Gp;
    while ((Gp + 1) * (Gp + 2) / 2 <= u) ++Gp;
    const int64_t rem = u - Gp * (Gp + 1) / 2;
    int64_t cb;
    if (colmajor_g > 0) {
      cb = colmajor_g - 2 - Gp;
      G = cb + 1 + (Gp - rem);
    } else {
      G = Gp + 1;
      cb = rem;
    }
    slot = cb * Q + t % Q;                                // strip s < Q*G
    tile0 = slot * CT;
  } else if (mode == 1) {                    // strips left of the diagonal block of the last row group
    G = ng - 1;
    slot = t;
    tile0 = slot * CT;
  } else {                                   // diagonal block of row group t/DT, tile t%DT
    G = t / DT;
    tile0 = DT * G + t % DT;
    slot = (int64_t)qint * G + t % DT;
  }
  const int64_t i0 = G * HR;
  const int tid = threadIdx.x, lane = tid & 63, rp = tid & 127;   // rows RPL*rp .. of the row group
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 7);          // columns cg + 2k, k < 16, of each tile
  const int half = __builtin_amdgcn_readfirstlane((tid >> 6) & 1);  // which 64*RPL rows this wave covers
  const int64_t gr = i0 + RPL * rp;
  double vrr[RPL], vri[RPL], prr[RPL], pri[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const bool in = !EDGE || gr + r < n;
    vrr[r] = in ? (double)v[gr + r].re : 0.0;
    vri[r] = in ? (double)v[gr + r].im : 0.0;
    prr[r] = pri[r] = 0.0;
  }
#pragma unroll 1
  for (int jt = 0; jt < CT; ++jt) {
    const int64_t j0 = (tile0 + jt) * CHC;
    if (EDGE && j0 >= n) break;              // ragged last row group: tiles past the matrix
    V e[16];
    if constexpr (!EDGE) {                   // every element is strictly below the diagonal and inside
      const C<R> *base = A + (j0 + cg) * lda + gr;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        e[k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(base + (int64_t)(2 * k) * lda))   // (NT: dense.hip,
                  : *reinterpret_cast<const V *>(base + (int64_t)(2 * k) * lda);                                //  herm_nt_policy)
      if constexpr (DSEL) {
        const int below = (int)(gr - (j0 + cg));      // row - column of this lane's element of column k = 0
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            const bool keep = below + r > 2 * k;                                 // strict lower triangle only
            e[k][2 * r] = keep ? e[k][2 * r] : R(0);
            e[k][2 * r + 1] = keep ? e[k][2 * r + 1] : R(0);
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int64_t gc = j0 + cg + 2 * k;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          C<R> x;
          if (gc < n && gr + r > gc && gr + r < n) x = A[gr + r + gc * lda];   // strict lower triangle only
          e[k][2 * r] = x.re;
          e[k][2 * r + 1] = x.im;
        }
      }
    }
    // FMAs + first butterfly stage, column pair (q, q+8) at a time (keeps the live set small)
    const bool hi = (lane & 32) != 0;
    double w8r[8], w8i[8], w4r[4], w4i[4], w2r[2], w2i[2], w1r, w1i;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t ca = j0 + cg + 2 * q, cb = ca + 16;
      const bool ina = !EDGE || ca < n, inb = !EDGE || cb < n;
      const double xar = ina ? (double)v[ca].re : 0.0, xai = ina ? (double)v[ca].im : 0.0;
      const double xbr = inb ? (double)v[cb].re : 0.0, xbi = inb ? (double)v[cb].im : 0.0;
      double par = 0.0, pai = 0.0, pbr = 0.0, pbi = 0.0;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const double ar = (double)e[q][2 * r], ai = (double)e[q][2 * r + 1];
        const double br = (double)e[q + 8][2 * r], bi = (double)e[q + 8][2 * r + 1];
        prr[r] = fma(ar, xar, prr[r]);       // L[r][c] * v[c]
        prr[r] = fma(-ai, xai, prr[r]);
        pri[r] = fma(ar, xai, pri[r]);
        pri[r] = fma(ai, xar, pri[r]);
        prr[r] = fma(br, xbr, prr[r]);
        prr[r] = fma(-bi, xbi, prr[r]);
        pri[r] = fma(br, xbi, pri[r]);
        pri[r] = fma(bi, xbr, pri[r]);
        par = fma(ar, vrr[r], par);          // conj(L[r][c]) * v[r]
        par = fma(ai, vri[r], par);
        pai = fma(ar, vri[r], pai);
        pai = fma(-ai, vrr[r], pai);
        pbr = fma(br, vrr[r], pbr);
        pbr = fma(bi, vri[r], pbr);
        pbi = fma(br, vri[r], pbi);
        pbi = fma(-bi, vrr[r], pbi);
      }
      w8r[q] = (hi ? pbr : par) + __shfl_xor(hi ? par : pbr, 32, 64);
      w8i[q] = (hi ? pbi : pai) + __shfl_xor(hi ? pai : pbi, 32, 64);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const bool h2 = (lane & 16) != 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        w4r[q] = (h2 ? w8r[4 + q] : w8r[q]) + __shfl_xor(h2 ? w8r[q] : w8r[4 + q], 16, 64);
        w4i[q] = (h2 ? w8i[4 + q] : w8i[q]) + __shfl_xor(h2 ? w8i[q] : w8i[4 + q], 16, 64);
      }
    }
    {
      const bool h2 = (lane & 8) != 0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        w2r[q] = (h2 ? w4r[2 + q] : w4r[q]) + __shfl_xor(h2 ? w4r[q] : w4r[2 + q], 8, 64);
        w2i[q] = (h2 ? w4i[2 + q] : w4i[q]) + __shfl_xor(h2 ? w4i[q] : w4i[2 + q], 8, 64);
      }
    }
    {
      const bool h2 = (lane & 4) != 0;
      w1r = (h2 ? w2r[1] : w2r[0]) + __shfl_xor(h2 ? w2r[0] : w2r[1], 4, 64);
      w1i = (h2 ? w2i[1] : w2i[0]) + __shfl_xor(h2 ? w2i[0] : w2i[1], 4, 64);
    }
    const bool him = (lane & 2) != 0;        // last halving step is over the two components: lanes with bit 1 keep im
    double w = (him ? w1i : w1r) + __shfl_xor(him ? w1r : w1i, 2, 64);
    w += __shfl_xor(w, 1, 64);
    if ((lane & 1) == 0) {
      const int k = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
      const int64_t gc = j0 + cg + 2 * k;
      if (!EDGE || gc < n) Pcol[((2 * G + half) * n + gc) * 2 + (him ? 1 : 0)] = w;
    }
  }
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    rowred[cg][RPL * rp + r][0] = prr[r];
    rowred[cg][RPL * rp + r][1] = pri[r];
  }
  __syncthreads();
  for (int tt = tid; tt < HR; tt += kBlock) {
    const int64_t row = i0 + tt;
    if (row < n) {
      f64x2 o;
      o[0] = rowred[0][tt][0] + rowred[1][tt][0];
      o[1] = rowred[0][tt][1] + rowred[1][tt][1];
      *reinterpret_cast<f64x2 *>(Prow + (slot * n + row) * 2) = o;
    }
  }
}

// ONE launch for everything that can use unmasked 16-byte loads (aligned A): the interior strips of the full row
// groups, then the tiles of their diagonal blocks (DSEL) — both at the unmasked path's register footprint.
template <typename R, int CT, bool NT>
__global__ void __launch_bounds__(kBlock)
cherm_pass_kernel(const C<R> *__restrict__ A, int64_t lda, const C<R> *__restrict__ v, int64_t n,
                  double *__restrict__ Prow, double *__restrict__ Pcol, int64_t ng, int qint, int64_t n_int, int colmajor_g) {
  __shared__ double rowred[2][CHermCfg<R>::HR][2];
  const int64_t t = blockIdx.x;
  if (t < n_int) return cherm_strip_body<R, CT, false, false, NT>(A, lda, v, n, Prow, Pcol, ng, qint, 0, t, rowred, colmajor_g);
  cherm_strip_body<R, 1, false, true, NT>(A, lda, v, n, Prow, Pcol, ng, qint, 2, t - n_int, rowred);
}

// The masked remainder, one launch: every strip when A is not 16-byte aligned (mode 0), the strips of the ragged last
// row group (mode 1), the diagonal tiles of the row groups from g0 on (mode 2).
template <typename R, int CT>
__global__ void __launch_bounds__(kBlock)
cherm_edge_kernel(const C<R> *__restrict__ A, int64_t lda, const C<R> *__restrict__ v, int64_t n,
                  double *__restrict__ Prow, double *__restrict__ Pcol, int64_t ng, int qint, int64_t n_all,
                  int64_t n_last, int64_t g0) {
  __shared__ double rowred[2][CHermCfg<R>::HR][2];
  int64_t t = blockIdx.x;
  if (t < n_all) return cherm_strip_body<R, CT, true>(A, lda, v, n, Prow, Pcol, ng, qint, 0, t, rowred);
  t -= n_all;
  if (t < n_last) return cherm_strip_body<R, CT, true>(A, lda, v, n, Prow, Pcol, ng, qint, 1, t, rowred);
  t -= n_last;
  cherm_strip_body<R, 1, true>(A, lda, v, n, Prow, Pcol, ng, qint, 2, t + CHermCfg<R>::DT * g0, rowred);
}

// 32 rows per workgroup, 8 lanes per row, fixed-order sums (as herm_finish_kernel), then the sum of
// src/linalg.jl:99-101 in Complex{R}: (d_i*v_i + (L*v)_i) + (L'*v)_i, α, β.
template <typename R, typename RA, typename RB, bool BETA0, bool DREAL>
__global__ void __launch_bounds__(kBlock)
cherm_finish_kernel(C<R> *__restrict__ res, const void *__restrict__ dptr, const C<R> *__restrict__ v,
                    const double *__restrict__ Prow, const double *__restrict__ Pcol, int64_t n, int ng, int q,
                    Sc<RA> a, Sc<RB> b) {
  const int r = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + r;
  __shared__ double s1r[8][32], s1i[8][32], s2r[8][32], s2i[8][32];
  double t1r = 0.0, t1i = 0.0, t2r = 0.0, t2i = 0.0;
  if (i < n) {
    constexpr int HR = CHermCfg<R>::HR, DT = CHermCfg<R>::DT;
    const int G = (int)(i / HR);
    for (int sx = sub; sx < q * G + DT; sx += 8) {          // L*v : q*G strips + DT diagonal tiles
      const f64x2 p = *reinterpret_cast<const f64x2 *>(Prow + ((int64_t)sx * n + i) * 2);
      t1r += p[0];
      t1i += p[1];
    }
    for (int h = 2 * G + sub; h < 2 * ng; h += 8) {         // L'*v: row halves at / below i
      const f64x2 p = *reinterpret_cast<const f64x2 *>(Pcol + ((int64_t)h * n + i) * 2);
      t2r += p[0];
      t2i += p[1];
    }
  }
  s1r[sub][r] = t1r;
  s1i[sub][r] = t1i;
  s2r[sub][r] = t2r;
  s2i[sub][r] = t2i;
  __syncthreads();
  if (sub != 0 || i >= n) return;
  double a1r = 0.0, a1i = 0.0, a2r = 0.0, a2i = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a1r += s1r[k][r];
    a1i += s1i[k][r];
    a2r += s2r[k][r];
    a2i += s2i[k][r];
  }
  R pr, pi;
  if constexpr (DREAL) {
    const R d = static_cast<const R *>(dptr)[i];
    pr = d * v[i].re;
    pi = d * v[i].im;
  } else {
    const C<R> d = static_cast<const C<R> *>(dptr)[i];
    pr = (d.re * v[i].re) - (d.im * v[i].im);
    pi = (d.re * v[i].im) + (d.im * v[i].re);
  }
  const C<R> inner((pr + (R)a1r) + (R)a2r, (pi + (R)a1i) + (R)a2i);
  RA tr, ti;
  a.mul(inner, tr, ti);
  res[i] = cfin<R, RA, RB, BETA0>(tr, ti, b.re, b.im, b.real, BETA0 ? C<R>() : res[i]);
}

template <typename R>
int32_t chermitian_two_pass(mxlo_ctx *ctx, C<R> *res, const void *d, bool d_real, const C<R> *A, int64_t lda,
                            const C<R> *v, int64_t n, const ScalArgs &s);

template <typename R>
int32_t chermitian(mxlo_ctx *ctx, C<R> *res, const void *d, bool d_real, const C<R> *A, int64_t lda, const C<R> *v,
                   int64_t n, const ScalArgs &s) {
  if (n == 0) return MXLO_OK;
  if (ctx->tune.cherm_two_pass) return chermitian_two_pass<R>(ctx, res, d, d_real, A, lda, v, n, s);
  constexpr int HR = CHermCfg<R>::HR, DT = CHermCfg<R>::DT, RPL = CHermCfg<R>::RPL;
  const int64_t ng = (n + HR - 1) / HR, ngf = n / HR;
  MXLO_REQUIRE(4 * ng * (ng + 1) < (1LL << 31), MXLO_ESHAPE, "complex opHermitian: n too large");
  // tiles per strip: whole diagonal-block-wide strips once there are two of them per CU, thinner strips below that
  const int64_t pairs = ng * (ng - 1) / 2;
  const int CT = pairs >= 2 * ctx->num_cu ? DT : ((DT / 2) * pairs >= 2 * ctx->num_cu ? 2 : 1), Q = DT / CT;
  const int64_t nslots = Q * (ng - 1) + DT;
  const size_t need = sizeof(double) * 2 * (size_t)(nslots + 2 * ng) * (size_t)n;   // Prow[nslots][n], Pcol[2ng][n], complex
  if (ctx->scratch_bytes < need) {            // stream-ordered users only: drain before the buffer is replaced
    if (ctx->scratch) {
      MXLO_HIP(hipStreamSynchronize(ctx->stream));
      MXLO_HIP(hipFree(ctx->scratch));
    }
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    hipError_t e = hipMalloc(&ctx->scratch, need);
    MXLO_REQUIRE(e == hipSuccess, MXLO_ENOMEM, "complex opHermitian scratch: %s", hipGetErrorString(e));
    ctx->scratch_bytes = need;
    ++ctx->scratch_generation;     // graphs that recorded the old workspace pointer are stale now
  }
  if (ctx->capturing) ctx->scratch_used_in_capture = true;
  double *Prow = (double *)ctx->scratch, *Pcol = Prow + 2 * (size_t)nslots * n;
  const bool aligned = (((uintptr_t)A & 15u) == 0) && (lda % RPL == 0);
  const int64_t gi = aligned ? ngf : 0;
  const int64_t n_int = gi > 1 ? Q * gi * (gi - 1) / 2 : 0;
  const int64_t n_dsel = (int64_t)DT * gi;
  const int64_t n_all = !aligned && ng > 1 ? Q * ng * (ng - 1) / 2 : 0;       // mode 0, masked
  const int64_t n_last = aligned && ng > ngf ? Q * (ng - 1) : 0;               // mode 1
  const int64_t n_diag = (int64_t)DT * (ng - gi);                              // mode 2, row groups gi .. ng-1
  const int64_t n_light = n_int + n_dsel, n_edge = n_all + n_last + n_diag;
  MXLO_REQUIRE(n_light < (1LL << 31) && n_edge < (1LL << 31), MXLO_ESHAPE, "complex opHermitian: n too large");
  // strip loads: nontemporal except for triangles of about the size of the Infinity Cache (the rule of dense.hip: herm_nt_policy)
  const int64_t tri = (int64_t)sizeof(C<R>) * n * (n / 2);
  const bool nt = ctx->tune.herm_nt >= 0 ? ctx->tune.herm_nt != 0
                                         : !(tri >= ctx->tune.herm_dp_min_bytes && tri < ctx->tune.herm_nt_min_bytes);
#define CHERM_LAUNCH(CT_)                                                                                        \
  {                                                                                                              \
    if (n_light > 0) {                                                                                           \
      if (nt)                                                                                                    \
        hipLaunchKernelGGL((cherm_pass_kernel<R, CT_, true>), dim3((unsigned)n_light), dim3(kBlock), 0, ctx->stream, A, lda, \
                           v, n, Prow, Pcol, ng, Q, n_int, ctx->tune.herm_order ? (int)gi : 0);                  \
      else                                                                                                       \
        hipLaunchKernelGGL((cherm_pass_kernel<R, CT_, false>), dim3((unsigned)n_light), dim3(kBlock), 0, ctx->stream, A, lda, \
                           v, n, Prow, Pcol, ng, Q, n_int, ctx->tune.herm_order ? (int)gi : 0);                  \
      MXLO_LAUNCH_CHECK();                                                                                       \
    }                                                                                                            \
    if (n_edge > 0) {                                                                                            \
      hipLaunchKernelGGL((cherm_edge_kernel<R, CT_>), dim3((unsigned)n_edge), dim3(kBlock), 0, ctx->stream, A, lda,  \
                         v, n, Prow, Pcol, ng, Q, n_all, n_last, gi);                                            \
      MXLO_LAUNCH_CHECK();                                                                                       \
    }                                                                                                            \
  }
  if (CT == DT) CHERM_LAUNCH(DT) else if (CT == 2) CHERM_LAUNCH(2) else CHERM_LAUNCH(1)
#undef CHERM_LAUNCH
  const unsigned blocks = (unsigned)((n + 31) / 32);
  return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
    const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
    if (d_real)
      hipLaunchKernelGGL((cherm_finish_kernel<R, RA, RB, B0, true>), dim3(blocks), dim3(kBlock), 0, ctx->stream, res, d, v,
                         Prow, Pcol, n, (int)ng, Q, a, b);
    else
      hipLaunchKernelGGL((cherm_finish_kernel<R, RA, RB, B0, false>), dim3(blocks), dim3(kBlock), 0, ctx->stream, res, d, v,
                         Prow, Pcol, n, (int)ng, Q, a, b);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

// the correctness-first form (two passes over the triangle: L*v by rows, L'*v by columns), kept behind
// mxlo_ctx_tune("cherm_two_pass", 1) as an independent implementation the tests compare the single pass with
template <typename R>
int32_t chermitian_two_pass(mxlo_ctx *ctx, C<R> *res, const void *d, bool d_real, const C<R> *A, int64_t lda,
                            const C<R> *v, int64_t n, const ScalArgs &s) {
  if (n == 0) return MXLO_OK;
  const size_t need = sizeof(C<R>) * 2 * (size_t)n;           // t1 = L*v, t2 = L'*v
  if (ctx->scratch_bytes < need) {            // stream-ordered users only: drain before the buffer is replaced
    if (ctx->scratch) {
      MXLO_HIP(hipStreamSynchronize(ctx->stream));
      MXLO_HIP(hipFree(ctx->scratch));
    }
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    hipError_t e = hipMalloc(&ctx->scratch, need);
    MXLO_REQUIRE(e == hipSuccess, MXLO_ENOMEM, "complex opHermitian scratch: %s", hipGetErrorString(e));
    ctx->scratch_bytes = need;
    ++ctx->scratch_generation;     // graphs that recorded the old workspace pointer are stale now
  }
  if (ctx->capturing) ctx->scratch_used_in_capture = true;
  C<R> *t1 = (C<R> *)ctx->scratch, *t2 = t1 + n;
  ScalArgs one{1.0, 0.0, 0.0, 0.0, true, true, true, true};
  MXLO_TRY((cgemv_rows<R, true, true>(ctx, t1, A, n, n, lda, v, false, one)));     // L*v
  MXLO_TRY((cgemv_cols<R, true, true>(ctx, t2, A, n, n, lda, v, true, one)));      // (v'*L)' = L'*v
  const int grid = grid_for(ctx, n, kBlock, 8);
  return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
    const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
    if (d_real)
      hipLaunchKernelGGL((cherm_final_kernel<R, RA, RB, B0, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, res, d, v,
                         (const C<R> *)t1, (const C<R> *)t2, n, a, b);
    else
      hipLaunchKernelGGL((cherm_final_kernel<R, RA, RB, B0, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, res, d, v,
                         (const C<R> *)t1, (const C<R> *)t2, n, a, b);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

// ---- kron on complex data: split x into planes, real MFMA GEMMs on planes (dense.hip: kron_planes), join with α, β ------
template <typename R>
__global__ void __launch_bounds__(kBlock)
cplx_split_kernel(R *__restrict__ re, R *__restrict__ im, const C<R> *__restrict__ x, int64_t n,
                  R *__restrict__ dmr = nullptr, R *__restrict__ spr = nullptr) {   // optional: im - re, re + im (Gauss form)
  // two elements per lane when everything is suitably aligned: 2 x sizeof(C<R>) read, 2 x sizeof(R) per plane written
  typedef R R2 __attribute__((ext_vector_type(2)));
  const bool vec = ((((uintptr_t)x) & (2 * sizeof(C<R>) - 1)) | ((((uintptr_t)re) | ((uintptr_t)im) | ((uintptr_t)dmr) | ((uintptr_t)spr)) &
                                                                (2 * sizeof(R) - 1))) == 0;
  const int64_t np = vec ? n / 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < np; i += (int64_t)gridDim.x * kBlock) {
    const C<R> e0 = x[2 * i], e1 = x[2 * i + 1];
    R2 r_, i_;
    r_[0] = e0.re; r_[1] = e1.re; i_[0] = e0.im; i_[1] = e1.im;
    reinterpret_cast<R2 *>(re)[i] = r_;
    reinterpret_cast<R2 *>(im)[i] = i_;
    if (dmr) {
      reinterpret_cast<R2 *>(dmr)[i] = i_ - r_;
      reinterpret_cast<R2 *>(spr)[i] = r_ + i_;
    }
  }
  for (int64_t i = 2 * np + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const C<R> e = x[i];
    re[i] = e.re;
    im[i] = e.im;
    if (dmr) {
      dmr[i] = e.im - e.re;
      spr[i] = e.re + e.im;
    }
  }
}
template <typename R, typename RA, typename RB, bool BETA0>
__global__ void __launch_bounds__(kBlock)
cplx_join_kernel(C<R> *__restrict__ res, const R *__restrict__ rr, const R *__restrict__ ri, int64_t n, Sc<RA> a, Sc<RB> b) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    RA tr, ti;
    a.mul(C<R>(rr[i], ri ? ri[i] : R(0)), tr, ti);      // ri == nullptr: a real vector promoted to Complex{R}
    res[i] = cfin<R, RA, RB, BETA0>(tr, ti, b.re, b.im, b.real, BETA0 ? C<R>() : res[i]);
  }
}

// the join pass of the Gauss form: re = k1 - k3, im = k1 + k2 formed on the fly (no rr / ri planes written and re-read)
template <typename R, typename RA, typename RB, bool BETA0>
__global__ void __launch_bounds__(kBlock)
cplx_join3_kernel(C<R> *__restrict__ res, const R *__restrict__ k1, const R *__restrict__ k2, const R *__restrict__ k3,
                  int64_t n, Sc<RA> a, Sc<RB> b) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const R re = k1[i] - k3[i], im = k1[i] + k2[i];
    RA tr, ti;
    a.mul(C<R>(re, im), tr, ti);
    res[i] = cfin<R, RA, RB, BETA0>(tr, ti, b.re, b.im, b.real, BETA0 ? C<R>() : res[i]);
  }
}

template <typename R>
int32_t ckron(mxlo_ctx *ctx, C<R> *res, const R *Ar, const R *Ai, int64_t am, int64_t an, int64_t lda, int mode_a,
              const R *Br, const R *Bi, int64_t bp, int64_t bq, int64_t ldb, int mode_b, const C<R> *x, R *work,
              const ScalArgs &s) {
  const bool ta = (mode_a & 1) != 0, tb = (mode_b & 1) != 0;
  const int64_t m = ta ? an : am, n = ta ? am : an, p = tb ? bq : bp, q = tb ? bp : bq;
  const int64_t nout = p * m, nin = q * n;
  if (nout == 0) return MXLO_OK;
  if (nin == 0) {
    if (s.bre == 0 && s.bim == 0) return cfill<R>(ctx, res, nout, C<R>());
    return cscale<R>(ctx, res, nout, s.bre, s.bim, s.b_real, s.b64);
  }
  auto pad = [](int64_t k) { return (k + 3) & ~(int64_t)3; };   // every plane starts 16-byte aligned (the DMA GEMM needs it)
  R *xr = work, *xi = xr + pad(nin), *utr = xi + pad(nin), *uti = utr + pad(m * q), *rr = uti + pad(m * q), *ri = rr + pad(nout);
  hipLaunchKernelGGL((cplx_split_kernel<R>), dim3(grid_for(ctx, nin, kBlock, 8)), dim3(kBlock), 0, ctx->stream, xr, xi, x, nin);
  MXLO_LAUNCH_CHECK();
  MXLO_TRY((kron_planes<R>(ctx, rr, ri, Ar, Ai, am, an, lda, ta, (mode_a & 2) ? -1.0 : 1.0, Br, Bi, bp, bq, ldb, tb,
                           (mode_b & 2) ? -1.0 : 1.0, xr, xi, utr, uti)));
  const int grid = grid_for(ctx, nout, kBlock, 8);
  return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
    const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
    hipLaunchKernelGGL((cplx_join_kernel<R, RA, RB, B0>), dim3(grid), dim3(kBlock), 0, ctx->stream, res, (const R *)rr,
                       (const R *)ri, nout, a, b);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

// kron on complex data with 3 real GEMMs per complex product (dense.hip: kron_planes3)
template <typename R>
int32_t ckron3(mxlo_ctx *ctx, C<R> *res, const R *Ar, const R *Ai, const R *As, int64_t am, int64_t an, int64_t lda, int mode_a,
               const R *Br, const R *Bi, const R *Bs, int64_t bp, int64_t bq, int64_t ldb, int mode_b, const C<R> *x, R *work,
               const ScalArgs &s) {
  const bool ta = (mode_a & 1) != 0, tb = (mode_b & 1) != 0;
  const int64_t m = ta ? an : am, n = ta ? am : an, p = tb ? bq : bp, q = tb ? bp : bq;
  const int64_t nout = p * m, nin = q * n;
  if (nout == 0) return MXLO_OK;
  if (nin == 0) {
    if (s.bre == 0 && s.bim == 0) return cfill<R>(ctx, res, nout, C<R>());
    return cscale<R>(ctx, res, nout, s.bre, s.bim, s.b_real, s.b64);
  }
  auto pad = [](int64_t k) { return (k + 3) & ~(int64_t)3; };
  R *xr = work, *xi = xr + pad(nin), *xd = xi + pad(nin), *xs = xd + pad(nin), *rr = xs + pad(nin), *ri = rr + pad(nout),
    *rest = ri + pad(nout);
  hipLaunchKernelGGL((cplx_split_kernel<R>), dim3(grid_for(ctx, nin, kBlock, 8)), dim3(kBlock), 0, ctx->stream, xr, xi, x, nin,
                     Ai ? xd : (R *)nullptr, Ai ? xs : (R *)nullptr);
  MXLO_LAUNCH_CHECK();
  const R *kk[3] = {nullptr, nullptr, nullptr};
  MXLO_TRY((kron_planes3<R>(ctx, rr, ri, Ar, Ai, As, am, an, lda, ta, (mode_a & 2) ? -1.0 : 1.0, Br, Bi, Bs, bp, bq, ldb, tb,
                            (mode_b & 2) ? -1.0 : 1.0, xr, xi, xd, xs, rest, kk)));
  const int grid = grid_for(ctx, nout, kBlock, 8);
  return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
    const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
    if (kk[0])
      hipLaunchKernelGGL((cplx_join3_kernel<R, RA, RB, B0>), dim3(grid), dim3(kBlock), 0, ctx->stream, res, kk[0], kk[1], kk[2],
                         nout, a, b);
    else
      hipLaunchKernelGGL((cplx_join_kernel<R, RA, RB, B0>), dim3(grid), dim3(kBlock), 0, ctx->stream, res, (const R *)rr,
                         (const R *)ri, nout, a, b);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

}  // namespace

#define CHECK_C(name)                                                                                        \
  MXLO_REQUIRE(ctx != nullptr, MXLO_EINVAL, name ": ctx is NULL");                                           \
  MXLO_DEVICE_GUARD(ctx);                                                                                    \
  MXLO_REQUIRE(dtype == MXLO_C64 || dtype == MXLO_C32, MXLO_EINVAL, name ": dtype %d is not a complex dtype", dtype)

MXLO_API int32_t mxlo_diag_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *d, const void *v,
                                 int64_t n_min, int64_t nrow, double alpha_re, double alpha_im, double beta_re,
                                 double beta_im, int32_t flags) {
  CHECK_C("mxlo_diag_mul_c");
  MXLO_REQUIRE(n_min >= 0 && nrow >= n_min, MXLO_ESHAPE, "mxlo_diag_mul_c: n_min=%lld nrow=%lld", (long long)n_min,
               (long long)nrow);
  MXLO_REQUIRE(nrow == 0 || (res && (n_min == 0 || (d && v))), MXLO_EINVAL, "mxlo_diag_mul_c: NULL operand");
  const bool cj = (flags & MXLO_CONJ_D) != 0;
  if (dtype == MXLO_C64)
    return cdiag<double>(ctx, (C<double> *)res, (const C<double> *)d, (const C<double> *)v, n_min, nrow,
                         scal_args(8, alpha_re, alpha_im, beta_re, beta_im, flags), cj);
  return cdiag<float>(ctx, (C<float> *)res, (const C<float> *)d, (const C<float> *)v, n_min, nrow,
                      scal_args(4, alpha_re, alpha_im, beta_re, beta_im, flags), cj);
}

MXLO_API int32_t mxlo_eye_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *v, int64_t n_min, int64_t nrow,
                                double alpha_re, double alpha_im, double beta_re, double beta_im, int32_t flags) {
  CHECK_C("mxlo_eye_mul_c");
  MXLO_REQUIRE(n_min >= 0 && nrow >= n_min, MXLO_ESHAPE, "mxlo_eye_mul_c: n_min=%lld nrow=%lld", (long long)n_min,
               (long long)nrow);
  MXLO_REQUIRE(nrow == 0 || (res && (n_min == 0 || v)), MXLO_EINVAL, "mxlo_eye_mul_c: NULL operand");
  if (dtype == MXLO_C64)
    return ceye<double>(ctx, (C<double> *)res, (const C<double> *)v, n_min, nrow,
                        scal_args(8, alpha_re, alpha_im, beta_re, beta_im, flags), flags);
  return ceye<float>(ctx, (C<float> *)res, (const C<float> *)v, n_min, nrow,
                     scal_args(4, alpha_re, alpha_im, beta_re, beta_im, flags), flags);
}

MXLO_API int32_t mxlo_zeros_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t nrow, double beta_re,
                                  double beta_im, int32_t flags) {
  CHECK_C("mxlo_zeros_mul_c");
  MXLO_REQUIRE(nrow >= 0 && (nrow == 0 || res), MXLO_EINVAL, "mxlo_zeros_mul_c: bad argument");
  const ScalArgs s = scal_args(dtype == MXLO_C64 ? 8 : 4, 0, 0, beta_re, beta_im, flags);
  if (s.bre == 0 && s.bim == 0) {
    if (dtype == MXLO_C64) return cfill<double>(ctx, (C<double> *)res, nrow, C<double>());
    return cfill<float>(ctx, (C<float> *)res, nrow, C<float>());
  }
  if (dtype == MXLO_C64) return cscale<double>(ctx, (C<double> *)res, nrow, s.bre, s.bim, s.b_real, true);
  return cscale<float>(ctx, (C<float> *)res, nrow, s.bre, s.bim, s.b_real, s.b64);
}

MXLO_API int32_t mxlo_scale_c(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t n, double alpha_re, double alpha_im,
                              int32_t flags) {
  CHECK_C("mxlo_scale_c");
  MXLO_REQUIRE(n >= 0 && (n == 0 || res), MXLO_EINVAL, "mxlo_scale_c: bad argument");
  const ScalArgs s = scal_args(dtype == MXLO_C64 ? 8 : 4, alpha_re, alpha_im, 0, 0, flags);
  if (dtype == MXLO_C64) return cscale<double>(ctx, (C<double> *)res, n, s.are, s.aim, s.a_real, true);
  return cscale<float>(ctx, (C<float> *)res, n, s.are, s.aim, s.a_real, s.a64);
}

MXLO_API int32_t mxlo_conj_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *v, int64_t n) {
  CHECK_C("mxlo_conj_c");
  MXLO_REQUIRE(n >= 0 && (n == 0 || (res && v)), MXLO_EINVAL, "mxlo_conj_c: bad argument");
  if (dtype == MXLO_C64)
    return launch_map<C<double>, 1, false, false>(ctx, (C<double> *)res, (const C<double> *)v, (const C<double> *)nullptr,
                                                  n, CConjOp<double>{});
  return launch_map<C<float>, 1, false, false>(ctx, (C<float> *)res, (const C<float> *)v, (const C<float> *)nullptr, n,
                                               CConjOp<float>{});
}

MXLO_API int32_t mxlo_dot_c(mxlo_ctx *ctx, int32_t dtype, const void *a, const void *b, int64_t n, double *out_dev) {
  CHECK_C("mxlo_dot_c");
  MXLO_REQUIRE(n >= 0 && out_dev && (n == 0 || (a && b)), MXLO_EINVAL, "mxlo_dot_c: bad argument");
  if (dtype == MXLO_C64) return cdotc<double>(ctx, (const C<double> *)a, (const C<double> *)b, n, out_dev);
  return cdotc<float>(ctx, (const C<float> *)a, (const C<float> *)b, n, out_dev);
}

MXLO_API int32_t mxlo_householder_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *h, const void *v,
                                        int64_t n, double alpha_re, double alpha_im, double beta_re, double beta_im,
                                        int32_t flags) {
  CHECK_C("mxlo_householder_mul_c");
  MXLO_REQUIRE(n >= 0 && (n == 0 || (res && h && v)), MXLO_EINVAL, "mxlo_householder_mul_c: bad argument");
  if (dtype == MXLO_C64)
    return chouse<double>(ctx, (C<double> *)res, (const C<double> *)h, (const C<double> *)v, n,
                          scal_args(8, alpha_re, alpha_im, beta_re, beta_im, flags));
  return chouse<float>(ctx, (C<float> *)res, (const C<float> *)h, (const C<float> *)v, n,
                       scal_args(4, alpha_re, alpha_im, beta_re, beta_im, flags));
}

MXLO_API int32_t mxlo_gemv_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *M, int64_t m, int64_t n, int64_t ld,
                             const void *v, double alpha_re, double alpha_im, double beta_re, double beta_im,
                             int32_t op_mode, int32_t flags) {
  CHECK_C("mxlo_gemv_c");
  MXLO_REQUIRE(m >= 0 && n >= 0 && ld >= (m > 1 ? m : 1), MXLO_ESHAPE, "mxlo_gemv_c: bad shape");
  MXLO_REQUIRE(op_mode >= MXLO_OP_N && op_mode <= MXLO_OP_J, MXLO_EINVAL, "mxlo_gemv_c: bad op_mode %d", op_mode);
  const int64_t nres = (op_mode == MXLO_OP_N || op_mode == MXLO_OP_J) ? m : n;
  MXLO_REQUIRE(nres == 0 || (res && (m == 0 || n == 0 || (M && v))), MXLO_EINVAL, "mxlo_gemv_c: NULL operand");
  if (dtype == MXLO_C64)
    return cgemv_any<double>(ctx, (C<double> *)res, (const C<double> *)M, m, n, ld, (const C<double> *)v, op_mode,
                             scal_args(8, alpha_re, alpha_im, beta_re, beta_im, flags), flags);
  return cgemv_any<float>(ctx, (C<float> *)res, (const C<float> *)M, m, n, ld, (const C<float> *)v, op_mode,
                          scal_args(4, alpha_re, alpha_im, beta_re, beta_im, flags), flags);
}

MXLO_API int32_t mxlo_hermitian_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *d, const void *A, int64_t lda,
                                      const void *v, int64_t n, double alpha_re, double alpha_im, double beta_re,
                                      double beta_im, int32_t flags) {
  CHECK_C("mxlo_hermitian_mul_c");
  MXLO_REQUIRE(n >= 0 && lda >= (n > 1 ? n : 1), MXLO_ESHAPE, "mxlo_hermitian_mul_c: bad shape");
  MXLO_REQUIRE(n == 0 || (res && d && A && v), MXLO_EINVAL, "mxlo_hermitian_mul_c: NULL operand");
  const bool d_real = (flags & MXLO_D_REAL) != 0;
  if (dtype == MXLO_C64)
    return chermitian<double>(ctx, (C<double> *)res, d, d_real, (const C<double> *)A, lda, (const C<double> *)v, n,
                              scal_args(8, alpha_re, alpha_im, beta_re, beta_im, flags));
  return chermitian<float>(ctx, (C<float> *)res, d, d_real, (const C<float> *)A, lda, (const C<float> *)v, n,
                           scal_args(4, alpha_re, alpha_im, beta_re, beta_im, flags));
}

MXLO_API int32_t mxlo_kron_mul_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *Ar, const void *Ai, int64_t am,
                                 int64_t an, int64_t lda, int32_t mode_a, const void *Br, const void *Bi, int64_t bp,
                                 int64_t bq, int64_t ldb, int32_t mode_b, const void *x, void *work, double alpha_re,
                                 double alpha_im, double beta_re, double beta_im, int32_t flags) {
  CHECK_C("mxlo_kron_mul_c");
  MXLO_REQUIRE(am >= 0 && an >= 0 && bp >= 0 && bq >= 0, MXLO_ESHAPE, "mxlo_kron_mul_c: negative size");
  MXLO_REQUIRE(lda >= (am > 1 ? am : 1) && ldb >= (bp > 1 ? bp : 1), MXLO_ESHAPE, "mxlo_kron_mul_c: bad leading dimension");
  MXLO_REQUIRE(mode_a >= 0 && mode_a <= 3 && mode_b >= 0 && mode_b <= 3, MXLO_EINVAL, "mxlo_kron_mul_c: bad factor mode");
  const int64_t m = (mode_a & 1) ? an : am, p = (mode_b & 1) ? bq : bp;
  if (m * p == 0) return MXLO_OK;
  MXLO_REQUIRE(res && Ar && Br && x && work, MXLO_EINVAL, "mxlo_kron_mul_c: NULL operand");
  if (dtype == MXLO_C64)
    return ckron<double>(ctx, (C<double> *)res, (const double *)Ar, (const double *)Ai, am, an, lda, mode_a, (const double *)Br,
                         (const double *)Bi, bp, bq, ldb, mode_b, (const C<double> *)x, (double *)work,
                         scal_args(8, alpha_re, alpha_im, beta_re, beta_im, flags));
  return ckron<float>(ctx, (C<float> *)res, (const float *)Ar, (const float *)Ai, am, an, lda, mode_a, (const float *)Br,
                      (const float *)Bi, bp, bq, ldb, mode_b, (const C<float> *)x, (float *)work,
                      scal_args(4, alpha_re, alpha_im, beta_re, beta_im, flags));
}

MXLO_API int64_t mxlo_kron_c3_work_size(int64_t am, int64_t an, int32_t mode_a, int64_t bp, int64_t bq, int32_t mode_b) {
  const int64_t m = (mode_a & 1) ? an : am, n = (mode_a & 1) ? am : an, p = (mode_b & 1) ? bq : bp, q = (mode_b & 1) ? bp : bq;
  auto pad = [](int64_t k) { return (k + 3) & ~(int64_t)3; };
  const int64_t kmax = std::max(m * q, p * m), fmax = std::max(am * an, bp * bq);
  return 4 * pad(q * n) + 2 * pad(p * m) + 4 * pad(m * q) + 3 * pad(kmax) + pad(fmax) + 8;
}

MXLO_API int32_t mxlo_plane_sum(mxlo_ctx *ctx, int32_t dtype, void *out, const void *a, const void *b, int64_t rows,
                                int64_t cols, int64_t ld, double sign) {
  CHECK_C("mxlo_plane_sum");
  MXLO_REQUIRE(rows >= 0 && cols >= 0 && ld >= (rows > 1 ? rows : 1), MXLO_ESHAPE, "mxlo_plane_sum: bad shape");
  if (rows * cols == 0) return MXLO_OK;
  MXLO_REQUIRE(out && a && b, MXLO_EINVAL, "mxlo_plane_sum: NULL operand");
  if (dtype == MXLO_C64) return plane_sum<double>(ctx, (double *)out, (const double *)a, (const double *)b, rows, cols, ld, sign);
  return plane_sum<float>(ctx, (float *)out, (const float *)a, (const float *)b, rows, cols, ld, sign);
}

MXLO_API int32_t mxlo_kron_mul_c3(mxlo_ctx *ctx, int32_t dtype, void *res, const void *Ar, const void *Ai, const void *As,
                                  int64_t am, int64_t an, int64_t lda, int32_t mode_a, const void *Br, const void *Bi,
                                  const void *Bs, int64_t bp, int64_t bq, int64_t ldb, int32_t mode_b, const void *x, void *work,
                                  double alpha_re, double alpha_im, double beta_re, double beta_im, int32_t flags) {
  CHECK_C("mxlo_kron_mul_c3");
  MXLO_REQUIRE(am >= 0 && an >= 0 && bp >= 0 && bq >= 0, MXLO_ESHAPE, "mxlo_kron_mul_c3: negative size");
  MXLO_REQUIRE(lda >= (am > 1 ? am : 1) && ldb >= (bp > 1 ? bp : 1), MXLO_ESHAPE, "mxlo_kron_mul_c3: bad leading dimension");
  MXLO_REQUIRE(mode_a >= 0 && mode_a <= 3 && mode_b >= 0 && mode_b <= 3, MXLO_EINVAL, "mxlo_kron_mul_c3: bad factor mode");
  const int64_t m = (mode_a & 1) ? an : am, p = (mode_b & 1) ? bq : bp;
  if (m * p == 0) return MXLO_OK;
  MXLO_REQUIRE(res && Ar && Br && x && work, MXLO_EINVAL, "mxlo_kron_mul_c3: NULL operand");
  MXLO_REQUIRE((((uintptr_t)work) & 15u) == 0, MXLO_EINVAL, "mxlo_kron_mul_c3: work must be 16-byte aligned");
  if (dtype == MXLO_C64)
    return ckron3<double>(ctx, (C<double> *)res, (const double *)Ar, (const double *)Ai, (const double *)As, am, an, lda, mode_a,
                          (const double *)Br, (const double *)Bi, (const double *)Bs, bp, bq, ldb, mode_b,
                          (const C<double> *)x, (double *)work,
                          scal_args(8, alpha_re, alpha_im, beta_re, beta_im, flags));
  return ckron3<float>(ctx, (C<float> *)res, (const float *)Ar, (const float *)Ai, (const float *)As, am, an, lda, mode_a,
                       (const float *)Br, (const float *)Bi, (const float *)Bs, bp, bq, ldb, mode_b, (const C<float> *)x,
                       (float *)work,
                       scal_args(4, alpha_re, alpha_im, beta_re, beta_im, flags));
}

// A REAL operator applied to complex vectors (K * x with eltype(K) = Float64, x::Vector{ComplexF64} — test/test_kron.jl
// "issue110"): the glue splits x into planes, applies the real operator to each, and joins res = α*(yr + i*yi) (+ β*res).
MXLO_API int32_t mxlo_split_c(mxlo_ctx *ctx, int32_t dtype, void *re, void *im, const void *x, int64_t n) {
  CHECK_C("mxlo_split_c");
  MXLO_REQUIRE(n >= 0 && (n == 0 || (re && im && x)), MXLO_EINVAL, "mxlo_split_c: bad argument");
  if (n == 0) return MXLO_OK;
  const int grid = grid_for(ctx, n, kBlock, 8);
  if (dtype == MXLO_C64)
    hipLaunchKernelGGL((cplx_split_kernel<double>), dim3(grid), dim3(kBlock), 0, ctx->stream, (double *)re, (double *)im,
                       (const C<double> *)x, n);
  else
    hipLaunchKernelGGL((cplx_split_kernel<float>), dim3(grid), dim3(kBlock), 0, ctx->stream, (float *)re, (float *)im,
                       (const C<float> *)x, n);
  MXLO_LAUNCH_CHECK();
  return MXLO_OK;
}

namespace {
template <typename R>
int32_t cjoin(mxlo_ctx *ctx, C<R> *res, const R *re, const R *im, int64_t n, const ScalArgs &s) {
  const int grid = grid_for(ctx, n, kBlock, 8);
  return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const Sc<RA> a{(RA)s.are, (RA)s.aim, s.a_real};
    const Sc<RB> b{(RB)s.bre, (RB)s.bim, s.b_real};
    hipLaunchKernelGGL((cplx_join_kernel<R, RA, RB, B0>), dim3(grid), dim3(kBlock), 0, ctx->stream, res, re, im, n, a, b);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}
}  // namespace

MXLO_API int32_t mxlo_join_c(mxlo_ctx *ctx, int32_t dtype, void *res, const void *re, const void *im, int64_t n,
                             double alpha_re, double alpha_im, double beta_re, double beta_im, int32_t flags) {
  CHECK_C("mxlo_join_c");
  MXLO_REQUIRE(n >= 0 && (n == 0 || (res && re)), MXLO_EINVAL, "mxlo_join_c: bad argument");   // im may be NULL (zeros)
  if (n == 0) return MXLO_OK;
  if (dtype == MXLO_C64)
    return cjoin<double>(ctx, (C<double> *)res, (const double *)re, (const double *)im, n,
                         scal_args(8, alpha_re, alpha_im, beta_re, beta_im, flags));
  return cjoin<float>(ctx, (C<float> *)res, (const float *)re, (const float *)im, n,
                      scal_args(4, alpha_re, alpha_im, beta_re, beta_im, flags));
}
