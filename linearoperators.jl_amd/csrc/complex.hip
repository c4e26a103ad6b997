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

using namespace mxlo;

namespace mxlo {
int32_t finalize_and_reduce(mxlo_ctx *ctx, int ncols, int nblocks, double *out_dev);
int32_t allreduce_hook(mxlo_ctx *ctx, double *dev, int64_t count);
}  // namespace mxlo

namespace {

template <typename R>
using C = cx<R>;

// t (complex, component type RA) (+ b*r): the β-term in RB, the sum in the wider type, one rounding to R per component
template <typename R, typename RA, typename RB, bool BETA0>
__device__ __forceinline__ C<R> cfin(RA tr, RA ti, RB br, RB bi, bool b_real, C<R> r) {
  if constexpr (BETA0) return C<R>((R)tr, (R)ti);
  else {
    using P = Wider<RA, RB>;
    const RB rr = (RB)r.re, ri = (RB)r.im;
    RB ur, ui;
    if (b_real) {
      ur = br * rr;
      ui = br * ri;
    } else {
      ur = (br * rr) - (bi * ri);
      ui = (br * ri) + (bi * rr);
    }
    return C<R>((R)((P)tr + (P)ur), (R)((P)ti + (P)ui));
  }
}

template <typename RA>
struct Sc {   // a caller scalar: complex, or real (a_real) multiplying componentwise
  RA re, im;
  bool real;
  template <typename R>
  __device__ __forceinline__ void mul(C<R> w, RA &tr, RA &ti) const {   // scalar * w
    const RA wr = (RA)w.re, wi = (RA)w.im;
    if (real) {
      tr = re * wr;
      ti = re * wi;
    } else {
      tr = (re * wr) - (im * wi);
      ti = (re * wi) + (im * wr);
    }
  }
};

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

struct ScalArgs {
  double are, aim, bre, bim;
  bool a_real, b_real, a64, b64;
};

inline ScalArgs scal_args(size_t comp, double are, double aim, double bre, double bim, int32_t flags) {
  ScalArgs s;
  s.a64 = comp == 8 || (flags & MXLO_ALPHA_F64);
  s.b64 = comp == 8 || (flags & MXLO_BETA_F64);
  s.a_real = (flags & MXLO_ALPHA_REAL) != 0;
  s.b_real = (flags & MXLO_BETA_REAL) != 0;
  auto rnd = [](double x, bool f64) { return f64 ? x : (double)(float)x; };
  s.are = rnd(are, s.a64);
  s.aim = s.a_real ? 0.0 : rnd(aim, s.a64);
  s.bre = rnd(bre, s.b64);
  s.bim = s.b_real ? 0.0 : rnd(bim, s.b64);
  return s;
}

// calls f.template operator()<RA, RB, BETA0>() for the component types of alpha / beta
template <typename R, typename F>
int32_t dispatch_c(const ScalArgs &s, F &&f) {
  const bool b0 = s.bre == 0 && s.bim == 0;   // β == zero(T2)
  if constexpr (sizeof(R) == 8) {
    return b0 ? f.template operator()<double, double, true>() : f.template operator()<double, double, false>();
  } else {
    if (b0) return s.a64 ? f.template operator()<double, double, true>() : f.template operator()<float, float, true>();
    if (s.a64) return s.b64 ? f.template operator()<double, double, false>() : f.template operator()<double, float, false>();
    return s.b64 ? f.template operator()<float, double, false>() : f.template operator()<float, float, false>();
  }
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
