// leaves.hip — C-ABI entry points of the elementwise / data-movement leaf closures:
// opDiagonal, opEye, opZeros, opOnes, prod3! helpers, opHouseholder,
// opRestriction / opExtension.
#include "common.h"
#include "stream_kernels.h"

using namespace mxlo;

#define CHECK_COMMON(name)                                                                       \
  MXLO_REQUIRE(ctx != nullptr, MXLO_EINVAL, name ": ctx is NULL");                               \
  MXLO_REQUIRE(dtype == MXLO_F64 || dtype == MXLO_F32, MXLO_EINVAL, name ": bad dtype %d", dtype)

static inline double eff_beta(int32_t dtype, int32_t flags, double beta) {
  return beta_is_f64(dtype == MXLO_F64 ? 8 : 4, flags) ? beta : (double)(float)beta;
}
static inline double eff_alpha(int32_t dtype, int32_t flags, double alpha) {
  return alpha_is_f64(dtype == MXLO_F64 ? 8 : 4, flags) ? alpha : (double)(float)alpha;
}

// ---- fill helper -------------------------------------------------------------------
template <typename T>
static int32_t fill(mxlo_ctx *ctx, T *p, int64_t n, T c) {
  if (n <= 0) return MXLO_OK;
  if (c == T(0)) {
    MXLO_HIP(hipMemsetAsync(p, 0, sizeof(T) * (size_t)n, ctx->stream));
    return MXLO_OK;
  }
  return launch_map<T, 0, false, false>(ctx, p, (const T *)nullptr, (const T *)nullptr, n,
                                        FillOp<T>{c});
}

// ---- opDiagonal ---------------------------------------------------------------------
template <typename T>
static int32_t diag_mul_t(mxlo_ctx *ctx, T *res, const T *d, const T *v, int64_t n_min,
                          int64_t nrow, double alpha, double beta, int32_t flags) {
  int32_t st = dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    if (flags & MXLO_D_SCALAR) {
      DiagScalarOp<T, CA, CB, B0> op{(CA)alpha, (CB)beta, d, (CA)0};
      return launch_map<T, 1, !B0, false>(ctx, res, v, (const T *)nullptr, n_min, op);
    }
    DiagOp<T, CA, CB, B0> op{(CA)alpha, (CB)beta};
    return launch_map<T, 2, !B0, false>(ctx, res, d, v, n_min, op);
  });
  MXLO_TRY(st);
  // rectangular form: res[n_min+1:end] .= 0 regardless of beta (special-operators.jl:150)
  return fill<T>(ctx, res + n_min, nrow - n_min, T(0));
}

MXLO_API int32_t mxlo_diag_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *d,
                               const void *v, int64_t n_min, int64_t nrow, double alpha,
                               double beta, int32_t flags) {
  CHECK_COMMON("mxlo_diag_mul");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(n_min >= 0 && nrow >= n_min, MXLO_ESHAPE, "mxlo_diag_mul: n_min=%lld nrow=%lld",
               (long long)n_min, (long long)nrow);
  MXLO_REQUIRE(nrow == 0 || (res && (n_min == 0 || (d && v))), MXLO_EINVAL,
               "mxlo_diag_mul: NULL operand");
  alpha = eff_alpha(dtype, flags, alpha);
  beta = eff_beta(dtype, flags, beta);
  if (dtype == MXLO_F64)
    return diag_mul_t<double>(ctx, (double *)res, (const double *)d, (const double *)v, n_min, nrow,
                              alpha, beta, flags);
  return diag_mul_t<float>(ctx, (float *)res, (const float *)d, (const float *)v, n_min, nrow,
                           alpha, beta, flags);
}

// ---- opEye / generic axpby ------------------------------------------------------------
template <typename T>
static int32_t eye_mul_t(mxlo_ctx *ctx, T *res, const T *v, int64_t n_min, int64_t nrow,
                         double alpha, double beta, int32_t flags) {
  int32_t st = dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    AxpbyOp<T, CA, CB, B0> op{(CA)alpha, (CB)beta};
    return launch_map<T, 1, !B0, false>(ctx, res, v, (const T *)nullptr, n_min, op);
  });
  MXLO_TRY(st);
  const int64_t ntail = nrow - n_min;
  if (ntail <= 0) return MXLO_OK;
  if (beta == 0) return fill<T>(ctx, res + n_min, ntail, T(0));
  const bool f64s = beta_is_f64(sizeof(T), flags);
  if (flags & MXLO_TAIL_BETA) return fill<T>(ctx, res + n_min, ntail, (T)beta);
  if (f64s)
    return launch_map<T, 0, true, false>(ctx, res + n_min, (const T *)nullptr, (const T *)nullptr,
                                         ntail, ScaleOp<T, double>{beta});
  return launch_map<T, 0, true, false>(ctx, res + n_min, (const T *)nullptr, (const T *)nullptr,
                                       ntail, ScaleOp<T, float>{(float)beta});
}

MXLO_API int32_t mxlo_eye_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *v,
                              int64_t n_min, int64_t nrow, double alpha, double beta,
                              int32_t flags) {
  CHECK_COMMON("mxlo_eye_mul");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(n_min >= 0 && nrow >= n_min, MXLO_ESHAPE, "mxlo_eye_mul: n_min=%lld nrow=%lld",
               (long long)n_min, (long long)nrow);
  MXLO_REQUIRE(nrow == 0 || (res && (n_min == 0 || v)), MXLO_EINVAL, "mxlo_eye_mul: NULL operand");
  alpha = eff_alpha(dtype, flags, alpha);
  beta = eff_beta(dtype, flags, beta);
  if (dtype == MXLO_F64)
    return eye_mul_t<double>(ctx, (double *)res, (const double *)v, n_min, nrow, alpha, beta, flags);
  return eye_mul_t<float>(ctx, (float *)res, (const float *)v, n_min, nrow, alpha, beta, flags);
}

// ---- opZeros / scale ---------------------------------------------------------------------
template <typename T>
static int32_t scale_t(mxlo_ctx *ctx, T *res, int64_t n, double s, bool f64s) {
  if (f64s)
    return launch_map<T, 0, true, false>(ctx, res, (const T *)nullptr, (const T *)nullptr, n,
                                         ScaleOp<T, double>{s});
  return launch_map<T, 0, true, false>(ctx, res, (const T *)nullptr, (const T *)nullptr, n,
                                       ScaleOp<T, float>{(float)s});
}

MXLO_API int32_t mxlo_zeros_mul(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t nrow, double beta,
                                int32_t flags) {
  CHECK_COMMON("mxlo_zeros_mul");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(nrow >= 0 && (nrow == 0 || res), MXLO_EINVAL, "mxlo_zeros_mul: bad argument");
  beta = eff_beta(dtype, flags, beta);
  if (beta == 0) {
    if (dtype == MXLO_F64) return fill<double>(ctx, (double *)res, nrow, 0.0);
    return fill<float>(ctx, (float *)res, nrow, 0.f);
  }
  if (dtype == MXLO_F64) return scale_t<double>(ctx, (double *)res, nrow, beta, true);
  return scale_t<float>(ctx, (float *)res, nrow, beta, beta_is_f64(4, flags));
}

MXLO_API int32_t mxlo_fill(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t n, double value) {
  CHECK_COMMON("mxlo_fill");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(n >= 0 && (n == 0 || res), MXLO_EINVAL, "mxlo_fill: bad argument");
  if (dtype == MXLO_F64) return fill<double>(ctx, (double *)res, n, value);
  return fill<float>(ctx, (float *)res, n, (float)value);
}

MXLO_API int32_t mxlo_scale(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t n, double alpha,
                            int32_t flags) {
  CHECK_COMMON("mxlo_scale");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(n >= 0 && (n == 0 || res), MXLO_EINVAL, "mxlo_scale: bad argument");
  alpha = eff_alpha(dtype, flags, alpha);
  if (dtype == MXLO_F64) return scale_t<double>(ctx, (double *)res, n, alpha, true);
  return scale_t<float>(ctx, (float *)res, n, alpha, alpha_is_f64(4, flags));
}

// ---- opOnes --------------------------------------------------------------------------------
// sum(v): a dedicated NC=1 reduction (no second operand), same fixed-order finalize.
namespace {
template <typename T>
__global__ void __launch_bounds__(kBlock)
sum_kernel(const T *__restrict__ v, int64_t n, double *__restrict__ partials) {
  constexpr int VEC = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  double acc = 0.0;
  // 16-byte body when v is 16-byte aligned, 4 independent loads in flight per lane; scalar head/tail
  const int64_t head = (((uintptr_t)v & 15u) == 0) ? 0 : n;   // unaligned: everything through the scalar loop
  const int64_t nvec = head == 0 ? n / VEC : 0;
  const V *vv = reinterpret_cast<const V *>(v);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    const V a = __builtin_nontemporal_load(vv + i), b = __builtin_nontemporal_load(vv + i + stride),
            c = __builtin_nontemporal_load(vv + i + 2 * stride), d = __builtin_nontemporal_load(vv + i + 3 * stride);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc += ((double)a[e] + (double)b[e]) + ((double)c[e] + (double)d[e]);
  }
  for (; i < nvec; i += stride) {
    const V a = vv[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc += (double)a[e];
  }
  for (int64_t k = nvec * VEC + (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += stride) acc += (double)v[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double lds[kBlock / kWave];
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}
}  // namespace

namespace mxlo {
int32_t allreduce_hook(mxlo_ctx *ctx, double *dev, int64_t count);
}

template <typename T>
static int32_t ones_mul_t(mxlo_ctx *ctx, T *res, int64_t nrow, const T *v, int64_t ncol,
                          double alpha, double beta, int32_t flags) {
  double *sum = ctx->scalars;  // slot 0
  if (ncol > 0) {
    const int grid = grid_for(ctx, ncol, kBlock * 8, ctx->tune.red_blocks_per_cu);
    hipLaunchKernelGGL((sum_kernel<T>), dim3(grid), dim3(kBlock), 0, ctx->stream, v, ncol,
                       ctx->partials);
    MXLO_LAUNCH_CHECK();
    MXLO_TRY(finalize_and_reduce(ctx, 1, grid, sum));
  } else {
    MXLO_HIP(hipMemsetAsync(sum, 0, sizeof(double), ctx->stream));
  }
  MXLO_TRY(allreduce_hook(ctx, sum, 1));
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    OnesOp<T, CA, CB, B0> op{(CA)alpha, (CB)beta, sum, (CA)0};
    return launch_map<T, 0, !B0, false>(ctx, res, (const T *)nullptr, (const T *)nullptr, nrow, op);
  });
}

MXLO_API int32_t mxlo_ones_mul(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t nrow,
                               const void *v, int64_t ncol, double alpha, double beta,
                               int32_t flags) {
  CHECK_COMMON("mxlo_ones_mul");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(nrow >= 0 && ncol >= 0, MXLO_ESHAPE, "mxlo_ones_mul: negative size");
  alpha = eff_alpha(dtype, flags, alpha);
  beta = eff_beta(dtype, flags, beta);
  if (dtype == MXLO_F64)
    return ones_mul_t<double>(ctx, (double *)res, nrow, (const double *)v, ncol, alpha, beta, flags);
  return ones_mul_t<float>(ctx, (float *)res, nrow, (const float *)v, ncol, alpha, beta, flags);
}

// ---- opHouseholder ----------------------------------------------------------------------------
template <typename T>
static int32_t householder_apply_t(mxlo_ctx *ctx, T *res, const T *h, const T *v, int64_t n,
                                   double alpha, double beta, int32_t flags, const double *dot) {
  const bool rev = ctx->tune.house_reverse != 0;
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    HouseholderOp<T, CA, CB, B0> op{(CA)alpha, (CB)beta, dot, T(0)};
    if (rev) return launch_map<T, 2, !B0, true>(ctx, res, h, v, n, op);
    return launch_map<T, 2, !B0, false>(ctx, res, h, v, n, op);
  });
}

// ---- single-launch opHouseholder for vectors that fit ONE wave of workgroups (n <= 2^21 doubles) ------------------------
// Two dependent launches (dot, update) cost more than the data movement below ~1 MiB. Here every workgroup keeps its
// slice of h and v in registers (read ONCE: 24 B/elt instead of 40), publishes its partial h'v into an exchange slot,
// waits until all G <= 256 co-resident workgroups have published, sums the G partials in the same fixed order (every
// workgroup obtains the bit-identical dot) and writes its slice of res.
// The exchange uses no fence: each partial is ONE self-contained 64-bit agent-scope atomic store, an empty slot is a
// NaN payload that arithmetic cannot produce, and readers poll with agent-scope loads (an agent-scope release fence
// would write back the whole L2 of every XCD). Two slot sets alternate by an epoch bit kept in device memory (graph
// replay safe): a launch uses set e, re-arms set 1-e for its successor, and workgroup 0 flips e after it has seen every
// partial — by then every workgroup has read e.
namespace {
template <typename T, typename CA, typename CB, bool BETA0, int VPT, bool ALIGNED>
__global__ void __launch_bounds__(kBlock)
householder_fused_kernel(T *__restrict__ res, const T *__restrict__ h, const T *__restrict__ v, int64_t n, CA alpha,
                         CB beta, unsigned long long *__restrict__ slots, unsigned long long ticks,
                         unsigned *__restrict__ fault, int drop) {
  constexpr int VEC = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const int tid = threadIdx.x, G = (int)gridDim.x, b = (int)blockIdx.x;
  unsigned long long *epoch = slots + 2 * kFusedSlots;
  const unsigned e = (unsigned)__hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u;
  unsigned long long *mine = slots + e * kFusedSlots, *other = slots + (1u - e) * kFusedSlots;
  for (int i = b + tid * G; i < kFusedSlots; i += G * kBlock)          // re-arm the other set for the next launch
    __hip_atomic_store(other + i, kSlotEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  T hh[VPT][VEC], vv[VPT][VEC];
  double acc = 0.0;
  const int64_t base = (int64_t)b * kBlock * VPT * VEC;
#pragma unroll
  for (int q = 0; q < VPT; ++q) {
    const int64_t i = base + ((int64_t)q * kBlock + tid) * VEC;
    if (ALIGNED && i + VEC <= n) {
      const V a = *reinterpret_cast<const V *>(h + i), c = *reinterpret_cast<const V *>(v + i);
#pragma unroll
      for (int k = 0; k < VEC; ++k) { hh[q][k] = a[k]; vv[q][k] = c[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        hh[q][k] = i + k < n ? h[i + k] : T(0);
        vv[q][k] = i + k < n ? v[i + k] : T(0);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < VPT; ++q)
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc = fma((double)hh[q][k], (double)vv[q][k], acc);

  __shared__ double lds[kBlock / 64];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) lds[wave] = acc;
  __syncthreads();
  if (tid == 0) {
    const double s = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    unsigned long long bits = (unsigned long long)__double_as_longlong(s);
    if (s != s) bits = kCanonicalNaN;                                // canonical NaN: never the empty marker
    if (b != drop) __hip_atomic_store(mine + b, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // wait for all G partials; lane t owns slot t (G <= 256 = one slot per lane). The wait is bounded (poll_slot): a
  // peer that never becomes resident turns into a NaN result and a raised ctx fault word, not into a hung GPU.
  double p = 0.0;                      // (G <= 256: one slot per lane, as rounds 3-5 summed them; G <= 512, round 6: lane t adds slots t, t + 256)
  if (tid < G) p = __longlong_as_double((long long)poll_slot(mine + tid, ticks, fault, kFaultHouseholder));
  if (tid + kBlock < G) p += __longlong_as_double((long long)poll_slot(mine + tid + kBlock, ticks, fault, kFaultHouseholder));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) p += __shfl_down(p, off, 64);
  __syncthreads();                                                     // lds reuse
  if (lane == 0) lds[wave] = p;
  __syncthreads();
  const double dot = (lds[0] + lds[1]) + (lds[2] + lds[3]);            // same order in every workgroup
  if (b == 0 && tid == 0)
    __hip_atomic_store(epoch, (unsigned long long)(1u - e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  HouseholderOp<T, CA, CB, BETA0> op{alpha, beta, nullptr, T(0)};
  op.c = (T)2 * (T)dot;                                                // = HouseholderOp::init()
#pragma unroll
  for (int q = 0; q < VPT; ++q) {
    const int64_t i = base + ((int64_t)q * kBlock + tid) * VEC;
    if (ALIGNED && i + VEC <= n) {
      V r;
      if constexpr (!BETA0) r = *reinterpret_cast<const V *>(res + i);
      V o;
#pragma unroll
      for (int k = 0; k < VEC; ++k) o[k] = op(hh[q][k], vv[q][k], BETA0 ? T(0) : r[k]);
      *reinterpret_cast<V *>(res + i) = o;
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        if (i + k < n) res[i + k] = op(hh[q][k], vv[q][k], BETA0 ? T(0) : res[i + k]);
    }
  }
}
}  // namespace

// workgroups the fused kernel would need with VPT vectors per lane
template <typename T>
static inline int64_t fused_grid(int64_t n, int vpt) {
  const int64_t per = (int64_t)kBlock * vpt * Vec16<T>::N;
  return (n + per - 1) / per;
}

template <typename T>
static int32_t householder_fused_t(mxlo_ctx *ctx, T *res, const T *h, const T *v, int64_t n, double alpha, double beta,
                                   int32_t flags, int vpt, bool *launched) {
  const bool aligned = ((((uintptr_t)res) | ((uintptr_t)h) | ((uintptr_t)v)) & 15u) == 0;
  const int grid = (int)fused_grid<T>(n, vpt);
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    bool fits = true;
    auto go = [&]<int VPT, bool AL>() {
      if (!(fits = coresident<householder_fused_kernel<T, CA, CB, B0, VPT, AL>>(ctx, grid))) return;
      hipLaunchKernelGGL((householder_fused_kernel<T, CA, CB, B0, VPT, AL>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         res, h, v, n, (CA)alpha, (CB)beta, ctx->xslots, fused_timeout_ticks(ctx), ctx->fault_dev,
                         ctx->tune.fused_debug_drop);
    };
    if (aligned) {
      if (vpt == 2) go.template operator()<2, true>();
      else if (vpt == 4) go.template operator()<4, true>();
      else if (vpt == 8) go.template operator()<8, true>();
      else go.template operator()<16, true>();
    } else {
      if (vpt == 2) go.template operator()<2, false>();
      else if (vpt == 4) go.template operator()<4, false>();
      else if (vpt == 8) go.template operator()<8, false>();
      else go.template operator()<16, false>();
    }
    *launched = fits;
    if (!fits) return MXLO_OK;          // the grid would not be co-resident on this device: the caller takes two passes
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

template <typename T>
static int32_t householder_t(mxlo_ctx *ctx, T *res, const T *h, const T *v, int64_t n,
                             double alpha, double beta, int32_t flags) {
  // single launch: no all-reduce hook (its host callback sits between the two passes), every workgroup co-resident
  if (ctx->tune.house_fused && !ctx->allreduce && n > 0 && ctx->fault_dev) {
    MXLO_TRY(fused_fault_check(ctx));   // an earlier single-launch apply that timed out is reported (and repaired) here
  }
  if (ctx->tune.house_fused && !ctx->allreduce && n > 0 && ctx->fault_dev) {
    // up to house_fused_per_cu (2) workgroups per CU: 16 vectors per lane = 160-170 registers, one wave per SIMD and workgroup,
    // two workgroups fit a CU (the occupancy query in householder_fused_t decides; it falls back to two passes otherwise)
    const int64_t cap = std::min<int64_t>((int64_t)ctx->num_cu * (ctx->tune.house_fused_per_cu >= 2 ? 2 : 1), kFusedSlots);
    const int64_t cap1 = std::min<int64_t>(ctx->num_cu, kFusedSlots);
    // vectors per lane: 2 / 4 / 8 up to n = 2^20 doubles, 16 (two 2 KiB register slices per lane) up to 2^21 — beyond
    // that the slices no longer fit the register file of one co-resident wave of workgroups
    const int vpt = fused_grid<T>(n, 2) <= 16 ? 2 : (fused_grid<T>(n, 4) <= 32 ? 4 : (fused_grid<T>(n, 8) <= cap1 ? 8 : 16));
    if (fused_grid<T>(n, vpt) <= cap) {
      bool launched = false;
      const int32_t st = householder_fused_t<T>(ctx, res, h, v, n, alpha, beta, flags, vpt, &launched);
      if (st != MXLO_OK || launched) return st;
    }
  }
  const T *cols[1] = {h};
  if (!ctx->allreduce && n >= 1 && n <= ctx->tune.house_inline_n) {
    // mid sizes: dots pass, then the update pass adds the partial sums up itself — two launches instead of three
    int nb = 0;
    MXLO_TRY(panel_dots_partials<T>(ctx, cols, 1, v, n, &nb));
    const bool rev = ctx->tune.house_reverse != 0;
    return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
      HouseholderPartialsOp<T, CA, CB, B0> op{(CA)alpha, (CB)beta, ctx->partials, nb, T(0)};
      if (rev) return launch_map<T, 2, !B0, true>(ctx, res, h, v, n, op);
      return launch_map<T, 2, !B0, false>(ctx, res, h, v, n, op);
    });
  }
  double *dot = ctx->scalars;  // slot 0
  MXLO_TRY(panel_dots<T>(ctx, cols, 1, v, n, dot));  // phase A (+ all-reduce hook)
  return householder_apply_t<T>(ctx, res, h, v, n, alpha, beta, flags, dot);  // phase B
}

MXLO_API int32_t mxlo_dot(mxlo_ctx *ctx, int32_t dtype, const void *a, const void *b, int64_t n,
                          double *out_dev) {
  CHECK_COMMON("mxlo_dot");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(n >= 0 && out_dev && (n == 0 || (a && b)), MXLO_EINVAL, "mxlo_dot: bad argument");
  if (dtype == MXLO_F64) {
    const double *cols[1] = {(const double *)a};
    return panel_dots<double>(ctx, cols, 1, (const double *)b, n, out_dev);
  }
  const float *cols[1] = {(const float *)a};
  return panel_dots<float>(ctx, cols, 1, (const float *)b, n, out_dev);
}

MXLO_API int32_t mxlo_householder_apply(mxlo_ctx *ctx, int32_t dtype, void *res, const void *h,
                                        const void *v, int64_t n, double alpha, double beta,
                                        int32_t flags, const double *dot_dev) {
  CHECK_COMMON("mxlo_householder_apply");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(n >= 0 && dot_dev && (n == 0 || (res && h && v)), MXLO_EINVAL,
               "mxlo_householder_apply: bad argument");
  alpha = eff_alpha(dtype, flags, alpha);
  beta = eff_beta(dtype, flags, beta);
  if (dtype == MXLO_F64)
    return householder_apply_t<double>(ctx, (double *)res, (const double *)h, (const double *)v, n,
                                       alpha, beta, flags, dot_dev);
  return householder_apply_t<float>(ctx, (float *)res, (const float *)h, (const float *)v, n, alpha,
                                    beta, flags, dot_dev);
}

MXLO_API int32_t mxlo_householder_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *h,
                                      const void *v, int64_t n, double alpha, double beta,
                                      int32_t flags) {
  CHECK_COMMON("mxlo_householder_mul");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(n >= 0 && (n == 0 || (res && h && v)), MXLO_EINVAL,
               "mxlo_householder_mul: bad argument");
  alpha = eff_alpha(dtype, flags, alpha);
  beta = eff_beta(dtype, flags, beta);
  if (dtype == MXLO_F64)
    return householder_t<double>(ctx, (double *)res, (const double *)h, (const double *)v, n, alpha,
                                 beta, flags);
  return householder_t<float>(ctx, (float *)res, (const float *)h, (const float *)v, n, alpha, beta,
                              flags);
}

// ---- opRestriction / opExtension ------------------------------------------------------------------
namespace {
template <typename T>
struct CopyBitsOp {
  __device__ void init() {}
  __device__ T operator()(T x, T, T) const { return x; }
};

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// res[k] = v[idx[k]-1]. Each lane owns KPT = 16/sizeof(E) consecutive outputs so that the index stream is
// read with 16-byte loads (and, for 4/8-byte elements with an aligned res, the output is one 16-byte
// nontemporal store); U such groups are in flight per lane. Random reads of v stay default-policy (cached).
template <typename E, bool VECOUT>
__global__ void __launch_bounds__(kBlock)
gather_idx_kernel(E *__restrict__ res, const E *__restrict__ v, const int64_t *__restrict__ idx,
                  int64_t nidx) {
  constexpr int KPT = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
  constexpr int U = 4;
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  const int64_t ngroups = (nidx + KPT - 1) / KPT;
  const bool idx_al = (((uintptr_t)idx) & 15u) == 0;
  for (int64_t base = ((int64_t)blockIdx.x * U) * kBlock + threadIdx.x; base < ngroups;
       base += (int64_t)gridDim.x * U * kBlock) {
    int64_t j[U][KPT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k0 = (base + (int64_t)u * kBlock) * KPT;
      if (KPT >= 2 && idx_al && k0 + KPT <= nidx) {
#pragma unroll
        for (int q = 0; q < KPT; q += 2) {
          const i64x2 t = __builtin_nontemporal_load(reinterpret_cast<const i64x2 *>(idx + k0 + q));
          j[u][q] = t[0] - 1;          // Julia indices are 1-based
          j[u][q + (KPT >= 2 ? 1 : 0)] = t[1] - 1;
        }
      } else {
#pragma unroll
        for (int q = 0; q < KPT; ++q) j[u][q] = k0 + q < nidx ? idx[k0 + q] - 1 : 0;
      }
    }
    E e[U][KPT];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < KPT; ++q) e[u][q] = v[j[u][q]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k0 = (base + (int64_t)u * kBlock) * KPT;
      if (VECOUT && k0 + KPT <= nidx) {
        struct alignas(16) Pack { E x[KPT]; };
        Pack p;
#pragma unroll
        for (int q = 0; q < KPT; ++q) p.x[q] = e[u][q];
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(*reinterpret_cast<u32x4 *>(&p), reinterpret_cast<u32x4 *>(res + k0));
      } else {
#pragma unroll
        for (int q = 0; q < KPT; ++q)
          if (k0 + q < nidx) res[k0 + q] = e[u][q];
      }
    }
  }
}

// res[k] = v[start0 + k*step]: each lane owns one 16-byte group of the CONTIGUOUS side (res) and issues one vector
// store; its VEC source elements are `step` apart. No integer division, 64-bit multiply once per group.
template <typename E>
__global__ void __launch_bounds__(kBlock)
gather_range_kernel(E *__restrict__ res, const E *__restrict__ v, int64_t start0, int64_t step,
                    int64_t len, int32_t res_aligned) {
  constexpr int VEC = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
  const int64_t ngroups = (len + VEC - 1) / VEC;
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * kBlock) {
    const int64_t k0 = g * VEC;
    const E *src = v + start0 + k0 * step;
    if (k0 + VEC <= len) {
      struct alignas(16) Pack { E x[VEC]; };
      Pack pk;
#pragma unroll
      for (int q = 0; q < VEC; ++q) pk.x[q] = src[(int64_t)q * step];
      if (res_aligned) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(*reinterpret_cast<u32x4 *>(&pk), reinterpret_cast<u32x4 *>(res + k0));
      } else {
#pragma unroll
        for (int q = 0; q < VEC; ++q) res[k0 + q] = pk.x[q];
      }
    } else {
      for (int q = 0; k0 + q < len; ++q) res[k0 + q] = src[(int64_t)q * step];
    }
  }
}

template <typename E>
__global__ void __launch_bounds__(kBlock)
scatter_idx_kernel(E *__restrict__ res, const E *__restrict__ u, const int64_t *__restrict__ idx,
                   const int64_t *__restrict__ pos, int64_t nidx) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nidx;
       k += (int64_t)gridDim.x * kBlock) {
    const int64_t src = pos ? pos[k] : k;
    res[idx[k] - 1] = u[src];
  }
}

// res[i] = u[(i-start0)/step] if i is on the range else 0 : every element written exactly once, 16 bytes per lane.
// No per-element division: a lane divides ONCE (its first group), then walks (q, r) = divmod(i - lo, |step|) forward
// with the precomputed divmod of the grid stride (sq, sr).
template <typename E>
__global__ void __launch_bounds__(kBlock)
extend_range_kernel(E *__restrict__ res, int64_t nres, const E *__restrict__ u, int64_t start0,
                    int64_t step, int64_t len, int64_t sq, int64_t sr, int32_t res_aligned) {
  constexpr int VEC = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
  const int64_t stop0 = start0 + (len - 1) * step;  // inclusive; step may be negative
  const int64_t lo = step > 0 ? start0 : stop0, hi = step > 0 ? stop0 : start0;
  const int64_t astep = step > 0 ? step : -step;
  const int64_t ngroups = (nres + VEC - 1) / VEC;
  int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= ngroups) return;
  // (q, r) for element i0 = g*VEC relative to lo; elements below lo get a NEGATIVE offset handled by the range test
  int64_t off = g * VEC - lo;
  int64_t q = off >= 0 ? off / astep : -((-off + astep - 1) / astep);   // floor division
  int64_t r = off - q * astep;                                          // 0 <= r < astep
  for (; g < ngroups; g += (int64_t)gridDim.x * kBlock) {
    const int64_t i0 = g * VEC;
    struct alignas(16) Pack { E x[VEC]; };
    Pack pk;
    int64_t qq = q, rr = r;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      E val;
      memset(&val, 0, sizeof(E));
      const int64_t i = i0 + j;
      if (rr == 0 && len > 0 && i >= lo && i <= hi) val = u[step > 0 ? qq : (len - 1) - qq];
      pk.x[j] = val;
      if (++rr == astep) { rr = 0; ++qq; }
    }
    if (res_aligned && i0 + VEC <= nres) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(*reinterpret_cast<u32x4 *>(&pk), reinterpret_cast<u32x4 *>(res + i0));
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (i0 + j < nres) res[i0 + j] = pk.x[j];
    }
    q += sq;
    r += sr;
    if (r >= astep) { r -= astep; ++q; }
  }
}

// ---- index extension with a SORTED plan: res is written exactly once, in full 16-byte vectors ------------------------
// `res .= 0; res[I] = u` (src/special-operators.jl:171-174) for a strictly increasing index list (the glue sorts the
// last-write-wins plan once; `pos` maps a sorted entry back to its position in u). Segment-owner formulation: a
// workgroup owns an output tile, finds the plan entries that fall into it with a wave-wide 64-ary lower bound (5 probe
// rounds for 5e7 entries; the first rounds hit the same few cache lines for every tile), builds the tile — zeros and
// values — in LDS, and streams it out with vector stores. Against memset + scatter this removes one full write of res
// and every partial-line write. Out-of-tile entries (a plan that is NOT strictly increasing) are dropped, never
// written out of bounds.
__device__ __forceinline__ int64_t wave_lower_bound(const int64_t *__restrict__ idx, int64_t lo, int64_t hi,
                                                    int64_t target, int lane) {
  // first p in [lo, hi] with idx[p] >= target (hi when none); all 64 lanes cooperate and return the same value
  while (true) {
    const int64_t n = hi - lo;
    if (n <= 0) return lo;
    if (n <= 64) {
      const int64_t p = lo + lane;
      const bool lt = p < hi && idx[p] < target;
      return lo + __popcll(__ballot(lt));
    }
    const int64_t p = lo + ((int64_t)(lane + 1) * n) / 65;          // 64 interior probes, strictly increasing, < hi
    const bool lt = idx[p] < target;
    const int c = __popcll(__ballot(lt));                           // sorted -> the `lt` lanes are a prefix
    const int64_t nlo = c == 0 ? lo : lo + ((int64_t)c * n) / 65 + 1;
    const int64_t nhi = c == 64 ? hi : lo + ((int64_t)(c + 1) * n) / 65;
    lo = nlo;
    hi = nhi;
  }
}

template <typename E, int TILE_BYTES>
__global__ void __launch_bounds__(kBlock)
extend_sorted_kernel(E *__restrict__ res, int64_t nres, const E *__restrict__ u, const int64_t *__restrict__ idx,
                     const int64_t *__restrict__ pos, int64_t nidx, int64_t ntiles, int32_t tiles_per_block,
                     int32_t res_aligned) {
  constexpr int TILE = TILE_BYTES / (int)sizeof(E);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __shared__ u32x4 smem[TILE_BYTES / 16];
  __shared__ int64_t bounds[2];
  E *tile = reinterpret_cast<E *>(smem);
  const int tid = threadIdx.x;
  const int64_t tb = (int64_t)blockIdx.x * tiles_per_block;
  const int64_t te = tb + tiles_per_block < ntiles ? tb + tiles_per_block : ntiles;
  const u32x4 z = {0u, 0u, 0u, 0u};
  int64_t s = 0;
  for (int64_t t = tb; t < te; ++t) {
    const int64_t t0 = t * TILE, t1 = t0 + TILE < nres ? t0 + TILE : nres;   // outputs [t0, t1); entry k is inside iff t0 < idx[k] <= t1
    if (tid < 64) {
      // ONE full search per workgroup; the following tiles start where the previous one ended. The end is bracketed:
      // strictly increasing indices put at most TILE entries into a tile (2 probe rounds, no over-read of the plan).
      if (t == tb) s = wave_lower_bound(idx, 0, nidx, t0 + 1, tid);
      const int64_t cap = s + TILE < nidx ? s + TILE : nidx;
      const int64_t e = wave_lower_bound(idx, s, cap, t1 + 1, tid);
      if (tid == 0) { bounds[0] = s; bounds[1] = e; }
    }
#pragma unroll
    for (int w = tid; w < TILE_BYTES / 16; w += kBlock) smem[w] = z;
    __syncthreads();
    s = bounds[0];
    const int64_t e = bounds[1];
    for (int64_t k = s + tid; k < e; k += kBlock) {
      const int64_t j = idx[k] - 1 - t0;
      const E val = u[pos ? pos[k] : k];
      if (j >= 0 && j < t1 - t0) tile[j] = val;
    }
    s = e;
    __syncthreads();
    const int64_t nbytes = (t1 - t0) * (int64_t)sizeof(E);
    if (res_aligned) {                       // t0 * sizeof(E) is a multiple of 16 (TILE_BYTES is): vectors stay aligned
      u32x4 *dst = reinterpret_cast<u32x4 *>(res + t0);
      const int nv = (int)(nbytes / 16);
      for (int w = tid; w < nv; w += kBlock) __builtin_nontemporal_store(smem[w], dst + w);
      for (int64_t j = (int64_t)nv * (16 / (int)sizeof(E)) + tid; j < t1 - t0; j += kBlock) res[t0 + j] = tile[j];
    } else {
      for (int64_t j = tid; j < t1 - t0; j += kBlock) res[t0 + j] = tile[j];
    }
    __syncthreads();
  }
}

template <typename F>
int32_t by_elem_size(int32_t es, F &&f) {
  switch (es) {
    case 4: return f.template operator()<u32>();
    case 8: return f.template operator()<u64>();
    case 16: return f.template operator()<u64x2>();
    default: set_error("element size %d not in {4,8,16}", es); return MXLO_EINVAL;
  }
}
}  // namespace

MXLO_API int32_t mxlo_gather(mxlo_ctx *ctx, int32_t elem_size, void *res, const void *v,
                             int64_t nv, const int64_t *idx, int64_t nidx) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_gather: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(nidx >= 0 && nv >= 0, MXLO_ESHAPE, "mxlo_gather: negative size");
  if (nidx == 0) return MXLO_OK;
  MXLO_REQUIRE(res && v && idx, MXLO_EINVAL, "mxlo_gather: NULL operand");
  return by_elem_size(elem_size, [&]<typename E>() -> int32_t {
    constexpr int KPT = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
    const int grid = grid_for(ctx, (nidx + KPT - 1) / KPT, kBlock * 4, ctx->tune.blocks_per_cu);
    if ((((uintptr_t)res) & 15u) == 0)
      hipLaunchKernelGGL((gather_idx_kernel<E, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, (E *)res,
                         (const E *)v, idx, nidx);
    else
      hipLaunchKernelGGL((gather_idx_kernel<E, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, (E *)res,
                         (const E *)v, idx, nidx);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

MXLO_API int32_t mxlo_gather_range(mxlo_ctx *ctx, int32_t elem_size, void *res, const void *v,
                                   int64_t nv, int64_t start, int64_t step, int64_t len) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_gather_range: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(len >= 0 && nv >= 0, MXLO_ESHAPE, "mxlo_gather_range: negative size");
  if (len == 0) return MXLO_OK;
  MXLO_REQUIRE(res && v, MXLO_EINVAL, "mxlo_gather_range: NULL operand");
  const int64_t last = start + (len - 1) * step;
  MXLO_REQUIRE(step != 0 && start >= 1 && start <= nv && last >= 1 && last <= nv, MXLO_ESHAPE,
               "mxlo_gather_range: %lld:%lld:%lld outside 1..%lld", (long long)start,
               (long long)step, (long long)last, (long long)nv);
  if (step == 1) {  // UnitRange: a contiguous copy through the streaming kernel (16-byte, nontemporal
                    // when large: ~1.8x the rate of hipMemcpyAsync D2D measured at 5e7 doubles);
                    // plain register moves, so every bit pattern (NaN payloads) survives
    const char *src = (const char *)v + (start - 1) * elem_size;
    if (elem_size == 4)
      return launch_map<float, 1, false, false>(ctx, (float *)res, (const float *)src, (const float *)nullptr,
                                                len, CopyBitsOp<float>{});
    const int64_t nd = len * (elem_size / 8);
    return launch_map<double, 1, false, false>(ctx, (double *)res, (const double *)src,
                                               (const double *)nullptr, nd, CopyBitsOp<double>{});
  }
  return by_elem_size(elem_size, [&]<typename E>() -> int32_t {
    constexpr int VEC = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
    const int grid = grid_for(ctx, (len + VEC - 1) / VEC, kBlock * 4, ctx->tune.blocks_per_cu);
    hipLaunchKernelGGL((gather_range_kernel<E>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (E *)res, (const E *)v, start - 1, step, len, (int32_t)((((uintptr_t)res) & 15u) == 0));
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

MXLO_API int32_t mxlo_scatter_zero(mxlo_ctx *ctx, int32_t elem_size, void *res, int64_t nres,
                                   const void *u, const int64_t *idx, const int64_t *pos,
                                   int64_t nidx) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_scatter_zero: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(nidx >= 0 && nres >= 0, MXLO_ESHAPE, "mxlo_scatter_zero: negative size");
  MXLO_REQUIRE(elem_size == 4 || elem_size == 8 || elem_size == 16, MXLO_EINVAL,
               "element size %d not in {4,8,16}", elem_size);
  if (nres == 0) return MXLO_OK;
  MXLO_REQUIRE(res, MXLO_EINVAL, "mxlo_scatter_zero: res is NULL");
  MXLO_HIP(hipMemsetAsync(res, 0, (size_t)nres * elem_size, ctx->stream));  // res .= 0
  if (nidx == 0) return MXLO_OK;
  MXLO_REQUIRE(u && idx, MXLO_EINVAL, "mxlo_scatter_zero: NULL operand");
  return by_elem_size(elem_size, [&]<typename E>() -> int32_t {
    const int grid = grid_for(ctx, nidx, kBlock * 4, ctx->tune.blocks_per_cu);
    hipLaunchKernelGGL((scatter_idx_kernel<E>), dim3(grid), dim3(kBlock), 0, ctx->stream, (E *)res,
                       (const E *)u, idx, pos, nidx);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

MXLO_API int32_t mxlo_scatter_zero_sorted(mxlo_ctx *ctx, int32_t elem_size, void *res, int64_t nres,
                                          const void *u, const int64_t *idx, const int64_t *pos, int64_t nidx) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_scatter_zero_sorted: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(nidx >= 0 && nres >= 0, MXLO_ESHAPE, "mxlo_scatter_zero_sorted: negative size");
  MXLO_REQUIRE(nidx <= nres, MXLO_ESHAPE, "mxlo_scatter_zero_sorted: %lld strictly increasing indices cannot fit 1..%lld",
               (long long)nidx, (long long)nres);
  if (nres == 0) return MXLO_OK;
  MXLO_REQUIRE(res && (nidx == 0 || (u && idx)), MXLO_EINVAL, "mxlo_scatter_zero_sorted: NULL operand");
  return by_elem_size(elem_size, [&]<typename E>() -> int32_t {
    constexpr int TB = 16384;
    const int64_t ntiles = (nres * (int64_t)sizeof(E) + TB - 1) / TB;
    // consecutive tiles per workgroup (one plan search per workgroup), keeping >= ~32 workgroups per CU for balance
    int64_t tpb = ctx->tune.extend_tiles_per_block > 0 ? ctx->tune.extend_tiles_per_block
                                                       : std::clamp<int64_t>(ntiles / ((int64_t)ctx->num_cu * 32), 1, 4);
    const int64_t nblocks = (ntiles + tpb - 1) / tpb;
    MXLO_REQUIRE(nblocks <= 0x7fffffffLL, MXLO_ESHAPE, "mxlo_scatter_zero_sorted: res too long");
    hipLaunchKernelGGL((extend_sorted_kernel<E, TB>), dim3((unsigned)nblocks), dim3(kBlock), 0, ctx->stream, (E *)res, nres,
                       (const E *)u, idx, pos, nidx, ntiles, (int32_t)tpb, (int32_t)((((uintptr_t)res) & 15u) == 0));
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

MXLO_API int32_t mxlo_scatter_zero_range(mxlo_ctx *ctx, int32_t elem_size, void *res, int64_t nres,
                                         const void *u, int64_t start, int64_t step, int64_t len) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_scatter_zero_range: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(len >= 0 && nres >= 0, MXLO_ESHAPE, "mxlo_scatter_zero_range: negative size");
  if (nres == 0) return MXLO_OK;
  MXLO_REQUIRE(res && (len == 0 || u), MXLO_EINVAL, "mxlo_scatter_zero_range: NULL operand");
  if (len > 0) {
    const int64_t last = start + (len - 1) * step;
    MXLO_REQUIRE(step != 0 && start >= 1 && start <= nres && last >= 1 && last <= nres,
                 MXLO_ESHAPE, "mxlo_scatter_zero_range: %lld:%lld:%lld outside 1..%lld",
                 (long long)start, (long long)step, (long long)last, (long long)nres);
  }
  return by_elem_size(elem_size, [&]<typename E>() -> int32_t {
    constexpr int VEC = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
    const int grid = grid_for(ctx, (nres + VEC - 1) / VEC, kBlock * 4, ctx->tune.blocks_per_cu);
    const int64_t astep = step > 0 ? step : -step;
    const int64_t stride = (int64_t)grid * kBlock * VEC;          // elements a lane advances per iteration
    hipLaunchKernelGGL((extend_range_kernel<E>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                       (E *)res, nres, (const E *)u, start - 1, len > 0 ? step : 1, len, len > 0 ? stride / astep : 0,
                       len > 0 ? stride % astep : 0, (int32_t)((((uintptr_t)res) & 15u) == 0));
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

// ---- index plans: a strictly increasing index set as BIT MASK + RANKS (round 5) -----------------------------------------
// A strictly increasing I ⊂ 1..n is one bit per element of the long vector plus, per 64-bit mask word, the number of set
// bits before it: k-th index <-> k-th set bit, so neither `res = v[I]` nor `res .= 0; res[I] = u` has to READ the index
// list (8 bytes per index, a third of the restriction's algorithmic bytes) — 1/8 + 1/8 byte per element of the long
// vector replaces it. Both applies become streaming passes over the LONG vector with the short one read / written through
// rank arithmetic (popcount of the mask bits below the lane's element):
//   extension  : every lane owns one 16-byte group of res: bits -> values from u[rank ..] (or u[pos[rank ..]]), zeros
//                elsewhere, ONE nontemporal store; res is written exactly once, no LDS tile, no search;
//   restriction: every lane owns one 16-byte group of v, loaded only when one of its bits is set (empty groups cost no
//                traffic), and stores its selected elements at res[rank ..] — neighbouring lanes write neighbouring
//                addresses, the memory pipeline merges them into full lines.
// Pure data movement: bit-exact (NaN payloads survive). Built ONCE at operator construction from host indices.
struct mxlo_index_plan {
  mxlo_ctx *ctx = nullptr;
  int64_t n = 0, nidx = 0, nwords = 0;
  // [nwords + 1] pairs (mask, rank): bit (i & 63) of mask word (i >> 6) <=> element i (0-based) is selected; rank = number
  // of selected elements before that word. Interleaved, so that a lane fetches both with ONE 16-byte load.
  unsigned long long *mr = nullptr;
};

namespace {
template <typename E, bool ALIGNED>
__global__ void __launch_bounds__(kBlock)
extend_mask_kernel(E *__restrict__ res, int64_t nres, const E *__restrict__ u, const int64_t *__restrict__ pos,
                   const u64x2 *__restrict__ mr) {
  constexpr int VEC = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
  constexpr int U = 4;
  struct alignas(16) Pack { E x[VEC]; };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int64_t ngroups = (nres + VEC - 1) / VEC;
  for (int64_t base = ((int64_t)blockIdx.x * U) * kBlock + threadIdx.x; base < ngroups; base += (int64_t)gridDim.x * U * kBlock) {
    unsigned long long w[U];
    int64_t r[U];
    bool ok[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t g = base + (int64_t)k * kBlock;
      ok[k] = g < ngroups;
      const u64x2 t = mr[ok[k] ? (g * VEC) >> 6 : 0];
      w[k] = t[0];
      r[k] = (int64_t)t[1];
    }
    Pack pk[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t i0 = (base + (int64_t)k * kBlock) * VEC;
      const int sh = (int)(i0 & 63);
      unsigned bits = ok[k] ? (unsigned)((w[k] >> sh) & ((1u << VEC) - 1u)) : 0u;
      int64_t rr = r[k] + __popcll(w[k] & ((1ull << sh) - 1ull));
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        E val;
        memset(&val, 0, sizeof(E));
        if ((bits >> j) & 1u) {
          val = u[pos ? pos[rr] : rr];
          ++rr;
        }
        pk[k].x[j] = val;
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (!ok[k]) continue;
      const int64_t i0 = (base + (int64_t)k * kBlock) * VEC;
      if (ALIGNED && i0 + VEC <= nres) {
        __builtin_nontemporal_store(*reinterpret_cast<u32x4 *>(&pk[k]), reinterpret_cast<u32x4 *>(res + i0));
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (i0 + j < nres) res[i0 + j] = pk[k].x[j];
      }
    }
  }
}

// DENSE: every 16-byte group of v is loaded unconditionally (at densities where nearly every 32-byte sector holds a
// selected element anyway, the branch around the load costs more than the few sectors it saves).
template <typename E, bool ALIGNED, bool DENSE>
__global__ void __launch_bounds__(kBlock)
gather_mask_kernel(E *__restrict__ res, const E *__restrict__ v, int64_t n, const u64x2 *__restrict__ mr) {
  constexpr int VEC = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
  constexpr int U = 4;
  struct alignas(16) Pack { E x[VEC]; };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int64_t ngroups = (n + VEC - 1) / VEC;
  const int64_t nfull = n / VEC;                                             // groups that lie entirely inside v
  for (int64_t base = ((int64_t)blockIdx.x * U) * kBlock + threadIdx.x; base < ngroups; base += (int64_t)gridDim.x * U * kBlock) {
    unsigned bits[U];
    int64_t r[U];
    Pack pk[U];
    const bool whole = base + (int64_t)(U - 1) * kBlock < nfull;             // all U groups of this lane are full groups
    if (ALIGNED && DENSE && whole) {
      u64x2 t[U];
#pragma unroll
      for (int k = 0; k < U; ++k) t[k] = mr[((base + (int64_t)k * kBlock) * VEC) >> 6];
#pragma unroll
      for (int k = 0; k < U; ++k) *reinterpret_cast<u32x4 *>(&pk[k]) = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(v + (base + (int64_t)k * kBlock) * VEC));
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int sh = (int)(((base + (int64_t)k * kBlock) * VEC) & 63);
        bits[k] = (unsigned)((t[k][0] >> sh) & ((1u << VEC) - 1u));
        r[k] = (int64_t)t[k][1] + __popcll(t[k][0] & ((1ull << sh) - 1ull));
      }
    } else {
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t g = base + (int64_t)k * kBlock;
        const bool ok = g < ngroups;
        const int64_t i0 = g * VEC;
        const u64x2 t = mr[ok ? i0 >> 6 : 0];
        const int sh = (int)(i0 & 63);
        bits[k] = ok ? (unsigned)((t[0] >> sh) & ((1u << VEC) - 1u)) : 0u;   // (bits beyond n are never set)
        r[k] = (int64_t)t[1] + __popcll(t[0] & ((1ull << sh) - 1ull));
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        if (!bits[k]) continue;                                               // an empty group is not read
        const int64_t i0 = (base + (int64_t)k * kBlock) * VEC;
        if (ALIGNED && i0 + VEC <= n) {
          *reinterpret_cast<u32x4 *>(&pk[k]) = *reinterpret_cast<const u32x4 *>(v + i0);
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            if (((bits[k] >> j) & 1u) && i0 + j < n) pk[k].x[j] = v[i0 + j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      int64_t rr = r[k];
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if ((bits[k] >> j) & 1u) res[rr++] = pk[k].x[j];
    }
  }
}
}  // namespace

// idx_host: HOST array of `nidx` 1-based indices, strictly increasing, all in 1..n (checked: MXLO_EDOMAIN otherwise).
MXLO_API int32_t mxlo_index_plan_create(mxlo_ctx *ctx, const int64_t *idx_host, int64_t nidx, int64_t n, mxlo_index_plan **out) {
  MXLO_REQUIRE(ctx && out && nidx >= 0 && n >= 0 && (nidx == 0 || idx_host), MXLO_EINVAL, "mxlo_index_plan_create: bad argument");
  MXLO_DEVICE_GUARD(ctx);
  *out = nullptr;
  const int64_t nwords = (n + 63) / 64;
  std::vector<unsigned long long> mr(2 * ((size_t)nwords + 1), 0ull);     // (mask, rank) pairs
  int64_t prev = 0;
  for (int64_t k = 0; k < nidx; ++k) {
    const int64_t i = idx_host[k];
    MXLO_REQUIRE(i > prev && i <= n, MXLO_EDOMAIN, "mxlo_index_plan_create: index %lld at position %lld is not strictly increasing within 1..%lld",
                 (long long)i, (long long)k, (long long)n);
    prev = i;
    mr[2 * (size_t)((i - 1) >> 6)] |= 1ull << ((i - 1) & 63);
  }
  for (int64_t w = 0; w < nwords; ++w) mr[2 * ((size_t)w + 1) + 1] = mr[2 * (size_t)w + 1] + (unsigned long long)__builtin_popcountll(mr[2 * (size_t)w]);
  mxlo_index_plan *p = new mxlo_index_plan();
  p->ctx = ctx;
  p->n = n;
  p->nidx = nidx;
  p->nwords = nwords;
  const size_t bytes = sizeof(unsigned long long) * mr.size();
  if (hipMalloc((void **)&p->mr, bytes) != hipSuccess) {
    (void)hipGetLastError();
    delete p;
    set_error("mxlo_index_plan_create: out of device memory (%zu bytes)", bytes);
    return MXLO_ENOMEM;
  }
  if (hipMemcpy(p->mr, mr.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(p->mr);
    delete p;
    set_error("mxlo_index_plan_create: upload failed");
    return MXLO_EHIP;
  }
  *out = p;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_index_plan_destroy(mxlo_index_plan *p) {
  if (!p) return MXLO_OK;
  MXLO_DEVICE_GUARD(p->ctx);
  (void)hipFree(p->mr);
  delete p;
  return MXLO_OK;
}

// res = v[I]  (mulRestrict!, src/special-operators.jl:167-169) for the plan's strictly increasing I; nv must be the plan's n
MXLO_API int32_t mxlo_gather_plan(mxlo_ctx *ctx, int32_t elem_size, void *res, const void *v, int64_t nv, const mxlo_index_plan *plan) {
  MXLO_REQUIRE(ctx && plan, MXLO_EINVAL, "mxlo_gather_plan: NULL argument");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(plan->ctx->device == ctx->device, MXLO_EINVAL, "mxlo_gather_plan: the plan lives on device %d, the ctx on %d", plan->ctx->device, ctx->device);
  MXLO_REQUIRE(nv == plan->n, MXLO_ESHAPE, "mxlo_gather_plan: v has %lld elements, the plan was built for %lld", (long long)nv, (long long)plan->n);
  if (plan->nidx == 0) return MXLO_OK;
  MXLO_REQUIRE(res && v, MXLO_EINVAL, "mxlo_gather_plan: NULL operand");
  return by_elem_size(elem_size, [&]<typename E>() -> int32_t {
    constexpr int VEC = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
    const int grid = grid_for(ctx, (nv + VEC - 1) / VEC, kBlock * 4, ctx->tune.blocks_per_cu);
    const u64x2 *mr = reinterpret_cast<const u64x2 *>(plan->mr);
    const bool dense = plan->nidx * 8 >= plan->n;           // >= 1/8: practically every 32-byte sector is touched
    if ((((uintptr_t)v) & 15u) != 0)
      hipLaunchKernelGGL((gather_mask_kernel<E, false, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, (E *)res, (const E *)v, nv, mr);
    else if (dense)
      hipLaunchKernelGGL((gather_mask_kernel<E, true, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, (E *)res, (const E *)v, nv, mr);
    else
      hipLaunchKernelGGL((gather_mask_kernel<E, true, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, (E *)res, (const E *)v, nv, mr);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

// res .= 0; res[I] = u[pos ? pos[k] : k]  (multRestrict!, src/special-operators.jl:171-174); nres must be the plan's n;
// pos (device, 0-based, or NULL) as in mxlo_scatter_zero_sorted
MXLO_API int32_t mxlo_scatter_zero_plan(mxlo_ctx *ctx, int32_t elem_size, void *res, int64_t nres, const void *u, const int64_t *pos,
                                        const mxlo_index_plan *plan) {
  MXLO_REQUIRE(ctx && plan, MXLO_EINVAL, "mxlo_scatter_zero_plan: NULL argument");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(plan->ctx->device == ctx->device, MXLO_EINVAL, "mxlo_scatter_zero_plan: the plan lives on device %d, the ctx on %d", plan->ctx->device, ctx->device);
  MXLO_REQUIRE(nres == plan->n, MXLO_ESHAPE, "mxlo_scatter_zero_plan: res has %lld elements, the plan was built for %lld", (long long)nres, (long long)plan->n);
  if (nres == 0) return MXLO_OK;
  MXLO_REQUIRE(res && (plan->nidx == 0 || u), MXLO_EINVAL, "mxlo_scatter_zero_plan: NULL operand");
  return by_elem_size(elem_size, [&]<typename E>() -> int32_t {
    constexpr int VEC = sizeof(E) >= 16 ? 1 : 16 / (int)sizeof(E);
    const int grid = grid_for(ctx, (nres + VEC - 1) / VEC, kBlock * 4, ctx->tune.blocks_per_cu);
    const u64x2 *mr = reinterpret_cast<const u64x2 *>(plan->mr);
    if ((((uintptr_t)res) & 15u) == 0)
      hipLaunchKernelGGL((extend_mask_kernel<E, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, (E *)res, nres, (const E *)u, pos, mr);
    else
      hipLaunchKernelGGL((extend_mask_kernel<E, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, (E *)res, nres, (const E *)u, pos, mr);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

// ---- row-shard staging for the collectives that move VECTORS (row-sharded dense / opHermitian) ----------------
// ShardPlan: shard r of an n-vector owns rows [lo(r), lo(r) + len(r)), lo(r) = r*q + min(r, rem), len(r) = q + (r < rem),
// q = n / world, rem = n % world. all-gather / reduce-scatter need equal counts per rank, so the wire format is
// `world` slots of pad = ceil(n / world) elements. ONE launch converts between the two layouts (blockIdx.y = shard, so
// no lane does an integer division):
//   PACK  : padded[r*pad + k] = k < len(r) && lo(r)+k < nvalid ? full[lo(r) + k] : 0      (zero pad slot; rows >= nvalid
//           are DEFINED as zero without being read — the caller's partial-sum vector need not be cleared beyond them)
//   UNPACK: full[lo(r) + k] = padded[r*pad + k],  k < len(r)
namespace {
template <typename E, bool PACK>
__global__ void __launch_bounds__(kBlock)
shard_stage_kernel(E *__restrict__ dst, const E *__restrict__ src, int64_t q, int64_t rem, int64_t pad, int64_t nvalid) {
  const int64_t r = blockIdx.y;
  const int64_t lo = r * q + (r < rem ? r : rem), len = q + (r < rem ? 1 : 0);
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < pad; k += (int64_t)gridDim.x * kBlock) {
    if constexpr (PACK) {
      E val{};
      if (k < len && lo + k < nvalid) val = src[lo + k];
      dst[r * pad + k] = val;
    } else {
      if (k < len) dst[lo + k] = src[r * pad + k];
    }
  }
}
}  // namespace

MXLO_API int32_t mxlo_shard_stage(mxlo_ctx *ctx, int32_t elem_size, void *dst, const void *src, int64_t n,
                                  int32_t world, int64_t nvalid, int32_t direction) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_shard_stage: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(n >= 0 && world >= 1 && world <= 65535, MXLO_ESHAPE, "mxlo_shard_stage: n = %lld, world = %d",
               (long long)n, world);
  MXLO_REQUIRE(direction == MXLO_SHARD_PACK || direction == MXLO_SHARD_UNPACK, MXLO_EINVAL,
               "mxlo_shard_stage: direction %d", direction);
  if (n == 0) return MXLO_OK;
  MXLO_REQUIRE(dst && src, MXLO_EINVAL, "mxlo_shard_stage: NULL operand");
  const int64_t q = n / world, rem = n % world, pad = (n + world - 1) / world;
  if (nvalid < 0 || nvalid > n) nvalid = n;
  return by_elem_size(elem_size, [&]<typename E>() -> int32_t {
    int64_t gx = (pad + kBlock - 1) / kBlock;
    const int64_t cap = std::max<int64_t>(1, (int64_t)ctx->num_cu * 8 / world);
    if (gx > cap) gx = cap;
    if (direction == MXLO_SHARD_PACK)
      hipLaunchKernelGGL((shard_stage_kernel<E, true>), dim3((unsigned)gx, (unsigned)world), dim3(kBlock), 0, ctx->stream,
                         (E *)dst, (const E *)src, q, rem, pad, nvalid);
    else
      hipLaunchKernelGGL((shard_stage_kernel<E, false>), dim3((unsigned)gx, (unsigned)world), dim3(kBlock), 0, ctx->stream,
                         (E *)dst, (const E *)src, q, rem, pad, nvalid);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}


// ---- kron of two diagonal / identity factors: fused row/col index decomposition ------------------------
// kron(A, B) with A = Diagonal(dA) (m x m) and B = Diagonal(dB) (p x p) (either may be the identity):
//   X = reshape(x, p, m);  (B * X * transpose(A))[r, c] = dB[r] * (X[r, c] * dA[c])      (src/kron.jl:14-22)
//   res[r + c*p] = alpha * that (+ beta * res[r + c*p])
// Output index i decomposes as (r, c) = (i % p, i / p); blockIdx.y walks c so no integer division is
// needed in the lanes. HBM-bound: 16 B/elt (+8 when beta != 0) plus the two small diagonals (cached).
namespace {
template <typename T, typename CA, typename CB, bool BETA0>
__global__ void __launch_bounds__(kBlock)
kron_diag_kernel(T *__restrict__ res, const T *__restrict__ dA, const T *__restrict__ dB,
                 const T *__restrict__ x, int64_t p, int64_t m, CA alpha, CB beta) {
  for (int64_t c = blockIdx.y; c < m; c += gridDim.y) {
    const T ac = dA ? dA[c] : T(1);
    T *rc = res + c * p;
    const T *xc = x + c * p;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < p; r += (int64_t)gridDim.x * kBlock) {
      const T br = dB ? dB[r] : T(1);
      const T inner = br * (xc[r] * ac);
      rc[r] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)inner, beta, BETA0 ? T(0) : rc[r]);
    }
  }
}

template <typename T>
int32_t kron_diag_t(mxlo_ctx *ctx, T *res, const T *dA, int64_t m, const T *dB, int64_t p, const T *x,
                    double alpha, double beta, int32_t flags) {
  if (m <= 0 || p <= 0) return MXLO_OK;
  int64_t gx = (p + kBlock - 1) / kBlock;
  if (gx > 1024) gx = 1024;
  int64_t gy = m < 65535 ? m : 65535;
  // keep the grid around 16 workgroups per CU
  const int64_t cap = (int64_t)ctx->num_cu * 16;
  if (gx * gy > cap) {
    gy = cap / gx;
    if (gy < 1) gy = 1;
  }
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    hipLaunchKernelGGL((kron_diag_kernel<T, CA, CB, B0>), dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0,
                       ctx->stream, res, dA, dB, x, p, m, (CA)alpha, (CB)beta);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}
}  // namespace

MXLO_API int32_t mxlo_kron_diag_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *dA, int64_t m,
                                    const void *dB, int64_t p, const void *x, double alpha, double beta,
                                    int32_t flags) {
  CHECK_COMMON("mxlo_kron_diag_mul");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(m >= 0 && p >= 0, MXLO_ESHAPE, "mxlo_kron_diag_mul: negative size");
  if (m == 0 || p == 0) return MXLO_OK;
  MXLO_REQUIRE(res && x, MXLO_EINVAL, "mxlo_kron_diag_mul: NULL operand");
  alpha = eff_alpha(dtype, flags, alpha);
  beta = eff_beta(dtype, flags, beta);
  if (dtype == MXLO_F64)
    return kron_diag_t<double>(ctx, (double *)res, (const double *)dA, m, (const double *)dB, p,
                               (const double *)x, alpha, beta, flags);
  return kron_diag_t<float>(ctx, (float *)res, (const float *)dA, m, (const float *)dB, p, (const float *)x,
                            alpha, beta, flags);
}
