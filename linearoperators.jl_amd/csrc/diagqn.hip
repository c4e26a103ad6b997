// diagqn.hip — push! of the diagonal quasi-Newton operators (src/DiagonalHessianApproximation.jl):
// DiagonalPSB (:45-64), DiagonalAndrei (:117-139), SpectralGradient (:190-199), DiagonalBFGS (:236-249).
// Their mul! IS mxlo_diag_mul (:37,112,179,226; SpectralGradient through MXLO_D_SCALAR).
//
// The reference walks s / y / d once per statement (norm, three or four dots, the broadcast update:
// 72-88 B/elt). Here ONE reduction pass reads s, y (and d) once and produces every sum the update needs
// (fixed-order, deterministic, all-reduce hook for row-sharded vectors), the scalars are evaluated on the
// host in the reference's statement order and eltype, and ONE map pass applies the update: 48 B/elt for
// PSB / Andrei, 32 B/elt for BFGS, 16 B/elt for SpectralGradient.
#include "common.h"
#include "stream_kernels.h"

#include <cmath>

namespace mxlo {
namespace {

constexpr int kSums = 6;  // sum s^2, sum s^4, sum s*y, sum s^2*d, sum |y|, #(s != 0)

__device__ __forceinline__ double wave_sum6(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <typename T, int VEC, bool USE_D, bool NT>
__global__ void __launch_bounds__(kBlock)
dqn_reduce_kernel(const T *__restrict__ s, const T *__restrict__ y, const T *__restrict__ d, int64_t head,
                  int64_t nvec, int64_t n, double *__restrict__ partials) {
  using V = typename std::conditional<VEC == 1, T, typename Vec16<T>::type>::type;
  double acc[kSums] = {0, 0, 0, 0, 0, 0};
  auto add = [&](T se, T ye, T de) {
    const double sd = (double)se, s2 = sd * sd;
    acc[0] += s2;
    acc[1] = fma(s2, s2, acc[1]);
    acc[2] = fma(sd, (double)ye, acc[2]);
    if constexpr (USE_D) acc[3] = fma(s2, (double)de, acc[3]);
    acc[4] += fabs((double)ye);
    acc[5] += se != (T)0 ? 1.0 : 0.0;
  };
  auto ld = [&](const T *p, int64_t i) -> V {
    if constexpr (VEC == 1) return p[head + i];
    else return NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(p + head + i * VEC))
                   : *reinterpret_cast<const V *>(p + head + i * VEC);
  };
  const int64_t stride = (int64_t)gridDim.x * kBlock;
#pragma unroll 2
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    const V sv = ld(s, i), yv = ld(y, i);
    V dv = sv;
    if constexpr (USE_D) dv = ld(d, i);
    if constexpr (VEC == 1) add(sv, yv, dv);
    else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) add(sv[e], yv[e], dv[e]);
    }
  }
  if (blockIdx.x == 0) {  // unaligned head and the tail past the last full vector
    const int64_t tail0 = head + nvec * VEC, cnt = head + (n - tail0);
    for (int64_t t = threadIdx.x; t < cnt; t += kBlock) {
      const int64_t i = t < head ? t : tail0 + (t - head);
      add(s[i], y[i], USE_D ? d[i] : (T)0);
    }
  }
  __shared__ double lds[kBlock / kWave][kSums];
#pragma unroll
  for (int c = 0; c < kSums; ++c) {
    const double w = wave_sum6(acc[c]);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6][c] = w;
  }
  __syncthreads();
  if (threadIdx.x < kSums)
    partials[(int64_t)threadIdx.x * kMaxRedBlocks + blockIdx.x] =
        (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
}

// B.d .+= q / sNorm2 .* s .^ 2          (:62)   c = q / sNorm2
template <typename T>
struct PsbOp {
  T c;
  __device__ void init() {}
  __device__ T operator()(T s, T, T d) const { return d + (c * (s * s)); }
};
// B.d .+= q / sNorm2 .* s .^ 2 .- 1     (:137)
template <typename T>
struct AndreiOp {
  T c;
  __device__ void init() {}
  __device__ T operator()(T s, T, T d) const { return d + ((c * (s * s)) - (T)1); }
};
// B.d .= abs.(y); B.d .*= sum(B.d) / sT_y   (:246-247)
template <typename T>
struct DiagBfgsOp {
  T c;
  __device__ void init() {}
  __device__ T operator()(T y, T, T) const { return fabs(y) * c; }
};

template <typename T>
inline double rT(double v) { return sizeof(T) == 4 ? (double)(float)v : v; }

template <typename T>
int32_t diagqn_push_t(mxlo_ctx *ctx, int32_t kind, T *d, const T *s, const T *y, int64_t n, int32_t *status) {
  *status = 0;
  constexpr int VEC = Vec16<T>::N;
  const bool use_d = kind == MXLO_DQN_PSB || kind == MXLO_DQN_ANDREI;
  double *out = ctx->scalars;
  if (n > 0) {
    const int64_t head0 = common_head<T>({s, y, use_d ? d : nullptr});
    const bool vec = head0 >= 0 && n >= 4 * VEC;
    const int64_t head = vec ? (head0 < n ? head0 : n) : 0;
    const int64_t nvec = vec ? (n - head) / VEC : n;
    int grid = grid_for(ctx, nvec, (int64_t)kBlock * 2, ctx->tune.red_blocks_per_cu);
    if (grid > kMaxRedBlocks) grid = kMaxRedBlocks;   // one partial slot per workgroup and sum
    const bool nt = (int64_t)sizeof(T) * n * (use_d ? 3 : 2) >= ctx->tune.nt_min_bytes;
#define DQN_GO(VEC_, USE_D_, NT_)                                                                          \
  hipLaunchKernelGGL((dqn_reduce_kernel<T, VEC_, USE_D_, NT_>), dim3(grid), dim3(kBlock), 0, ctx->stream, s, \
                     y, d, head, nvec, n, ctx->partials)
    if (vec) {
      if (use_d) { if (nt) DQN_GO(VEC, true, true); else DQN_GO(VEC, true, false); }
      else { if (nt) DQN_GO(VEC, false, true); else DQN_GO(VEC, false, false); }
    } else {
      if (use_d) DQN_GO(1, true, false); else DQN_GO(1, false, false);
    }
#undef DQN_GO
    MXLO_LAUNCH_CHECK();
    MXLO_TRY(finalize_and_reduce(ctx, kSums, grid, out));
  } else {
    MXLO_HIP(hipMemsetAsync(out, 0, sizeof(double) * kSums, ctx->stream));
  }
  MXLO_TRY(allreduce_hook(ctx, out, kSums));
  double h[kSums];
  MXLO_HIP(hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  MXLO_HIP(hipStreamSynchronize(ctx->stream));  // the s == 0 error and the scalars are host control flow
  if (h[5] == 0) {  // norm(s) == 0 (:51-53,123-125,242-244) / all(x -> x == 0, s) (:195-197)
    *status = 1;
    return MXLO_OK;
  }
  const double S2 = rT<T>(h[0]), S4 = rT<T>(h[1]), SY = rT<T>(h[2]), S2D = rT<T>(h[3]), AY = rT<T>(h[4]);
  if (kind == MXLO_DQN_SPECTRAL) {  // B.d[1] = dot(s, y) / dot(s, s)  (:198)
    const T v = (T)rT<T>(SY / S2);
    MXLO_HIP(hipMemcpyAsync(d, &v, sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    MXLO_HIP(hipStreamSynchronize(ctx->stream));  // `v` lives on this stack frame
    return MXLO_OK;
  }
  const double sNorm = rT<T>(std::sqrt(h[0]));          // norm(s, 2)
  const double sNorm2 = rT<T>(sNorm * sNorm);           // sNorm^2
  const double sT_y = rT<T>(SY / sNorm2);
  if (kind == MXLO_DQN_BFGS) {
    const T c = (T)rT<T>(AY / sT_y);                    // sum(B.d) / sT_y with B.d = abs.(y)
    return launch_map<T, 1, false, false>(ctx, d, y, (const T *)nullptr, n, DiagBfgsOp<T>{c});
  }
  const double trA2 = rT<T>(S4 / rT<T>(sNorm2 * sNorm2));  // dot(s2, s2) / sNorm2^2
  const double sT_B_s = rT<T>(S2D / sNorm2);
  double q = rT<T>(sT_y - sT_B_s);
  if (kind == MXLO_DQN_ANDREI) q = rT<T>(q + rT<T>(S2 / sNorm2));  // q += dot(s, s) / sNorm2
  q = rT<T>(q / trA2);
  const T c = (T)rT<T>(q / sNorm2);
  if (kind == MXLO_DQN_PSB) return launch_map<T, 1, true, false>(ctx, d, s, (const T *)nullptr, n, PsbOp<T>{c});
  return launch_map<T, 1, true, false>(ctx, d, s, (const T *)nullptr, n, AndreiOp<T>{c});
}

}  // namespace
}  // namespace mxlo

using namespace mxlo;

MXLO_API int32_t mxlo_diagqn_push(mxlo_ctx *ctx, int32_t dtype, int32_t kind, void *d, const void *s,
                                  const void *y, int64_t n, int32_t *status) {
  MXLO_REQUIRE(ctx && status && n >= 0 && (n == 0 || (s && y)) && d, MXLO_EINVAL, "mxlo_diagqn_push: bad argument");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(kind >= MXLO_DQN_PSB && kind <= MXLO_DQN_SPECTRAL, MXLO_EINVAL, "mxlo_diagqn_push: kind %d", kind);
  MXLO_REQUIRE(dtype == MXLO_F64 || dtype == MXLO_F32, MXLO_EINVAL, "bad dtype %d", dtype);
  if (dtype == MXLO_F64) return diagqn_push_t<double>(ctx, kind, (double *)d, (const double *)s, (const double *)y, n, status);
  return diagqn_push_t<float>(ctx, kind, (float *)d, (const float *)s, (const float *)y, n, status);
}
