// reductions.hip — deterministic global reductions: skinny-panel dots.
//
// panel_dots computes out[c] = dot(col_c, x) for NC columns in ONE pass that reads x
// once and every column once (the L-BFGS n x 2m panel pass; NC = 1 is the Householder
// h'v). Each lane keeps NC f64 accumulators in registers (f32 data is accumulated in
// f64 too: the kernel is HBM-bound, the FMAs are free), wave64 __shfl_down tree ->
// LDS across the 4 waves -> one partial per (column, workgroup). A second tiny kernel
// sums the partials in a FIXED order, so results are bit-reproducible run to run and
// independent of dispatch order (no float atomics). When the ctx has an all-reduce
// hook (row-sharded vectors) it is invoked on the finalized scalars.
#include "common.h"

namespace mxlo {

constexpr int kFuseMaxCols = 4;   // in-kernel finalize is serial over columns: only worth it for a few
// ... and for few workgroups: every workgroup pays an agent-scope release fence (L2 write-back + invalidate across the
// 8 XCDs) before taking its ticket. Measured (tools/sweep_fuse.py): 32 workgroups -2.5 us per reduction, 512
// workgroups +24 us, 1024 workgroups +180 us -> large reductions keep the separate finalize launch.
constexpr int kFuseMaxGrid = 64;

template <typename T, int NC>
struct ColPtrs {
  const T *p[NC];
};

// Sum over the wave in LANE 0 (other lanes: unspecified), on the tree of the shuffle-down loop this replaces —
//   v[l] += v[l + 32]; v[l] += v[l + 16]; ... ; v[l] += v[l + 1]
// — so every reduction of the library keeps its bits. A 64-bit __shfl_down is two ds_bpermute_b32 with an LDS round trip
// each and six dependent levels per sum: the 2 NC + 4 sums at the end of a push pass cost 1.8 us (5 columns) to 5.5 us (20)
// at launch-bound sizes. Here the upper half arrives with v_permlane32_swap, the odd rows with v_permlane16_swap (gfx950),
// the rest with DPP row shifts: register-speed VALU work that independent sums overlap freely.
__device__ __forceinline__ double wave_sum(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto slo = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);   // [1]: lanes 0..31 hold lanes 32..63
    const auto shi = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v += __hiloint2double((int)shi[1], (int)slo[1]);
  }
  {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto slo = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);   // [1]: rows 0, 2 hold rows 1, 3
    const auto shi = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v += __hiloint2double((int)shi[1], (int)slo[1]);
  }
  v += dpp_shift_or_zero<0x108, 0xf>(v);   // row_shl:8: lane l reads lane l + 8 of its row
  v += dpp_shift_or_zero<0x104, 0xf>(v);   // row_shl:4
  v += dpp_shift_or_zero<0x102, 0xf>(v);   // row_shl:2
  v += dpp_shift_or_zero<0x101, 0xf>(v);   // row_shl:1
#endif
  return v;
}

// agent-scope relaxed accesses for data handed from one workgroup to another INSIDE a kernel (the partial sums
// read by the last-arriving workgroup): coherent across the 8 XCDs' L2s, unlike plain loads/stores.
__device__ __forceinline__ void store_agent(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// FUSE (launch-bound sizes only, see kFuseMaxGrid): the workgroup that arrives LAST (atomic ticket) sums the
// per-workgroup partials in the same fixed order as finalize_kernel and writes out[c] — one dependent kernel fewer. The order of
// summation does not depend on which workgroup is last, so results stay bit-reproducible.
template <typename T, int VEC, int NC, int UNROLL, bool XVEC, bool NT, bool FUSE = false>
__global__ void __launch_bounds__(kBlock)
panel_dots_kernel(ColPtrs<T, NC> cols, const T *__restrict__ x, int64_t head, int64_t nvec,
                  int64_t n, double *__restrict__ partials, unsigned *__restrict__ ticket = nullptr,
                  double *__restrict__ out = nullptr) {
  using V = typename std::conditional<VEC == 1, T, typename Vec16<T>::type>::type;
  const int tid = threadIdx.x;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;

  constexpr int64_t CHUNK = (int64_t)kBlock * UNROLL;
  const int64_t nchunks = (nvec + CHUNK - 1) / CHUNK;
  auto load_x = [&](int64_t i, T (&xe)[VEC]) {
    if constexpr (VEC == 1) {
      xe[0] = x[head + i];
    } else if constexpr (XVEC) {
      const V xv = NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(x + head + i * VEC))
                                  : *reinterpret_cast<const V *>(x + head + i * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) xe[e] = xv[e];
    } else {  // x has a different 16-byte phase than the panel: element loads for x only
#pragma unroll
      for (int e = 0; e < VEC; ++e) xe[e] = x[head + i * VEC + e];
    }
  };
  auto load_c = [&](int c, int64_t i) -> V {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(cols.p[c] + head + i * VEC))
                          : *reinterpret_cast<const V *>(cols.p[c] + head + i * VEC);
  };
  auto accumulate = [&](const V (&cv)[NC], const T (&xe)[VEC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        T ce;
        if constexpr (VEC == 1) ce = cv[c];
        else ce = cv[c][e];
        acc[c] = fma((double)ce, (double)xe[e], acc[c]);
      }
    }
  };
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t base = ch * CHUNK + tid;
    if (base + (int64_t)(UNROLL - 1) * kBlock < nvec) {
      // whole chunk in range: issue every load of the chunk before the first FMA
      T xe[UNROLL][VEC];
      V cv[UNROLL][NC];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * kBlock;
        load_x(i, xe[u]);
#pragma unroll
        for (int c = 0; c < NC; ++c) cv[u][c] = load_c(c, i);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) accumulate(cv[u], xe[u]);
    } else {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * kBlock;
        if (i < nvec) {
          T xe[VEC];
          V cv[NC];
          load_x(i, xe);
#pragma unroll
          for (int c = 0; c < NC; ++c) cv[c] = load_c(c, i);
          accumulate(cv, xe);
        }
      }
    }
  }
  if constexpr (VEC > 1) {  // scalar head/tail
    if (blockIdx.x == gridDim.x - 1) {
      const int64_t tail0 = head + nvec * VEC;
      const int64_t cnt = head + (n - tail0);
      if (tid < cnt) {
        const int64_t i = tid < head ? tid : tail0 + (tid - head);
        const double xe = (double)x[i];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = fma((double)cols.p[c][i], xe, acc[c]);
      }
    }
  }
  __shared__ double lds[kBlock / kWave][NC];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const double s = wave_sum(acc[c]);
    if (lane == 0) lds[wave][c] = s;
  }
  __syncthreads();
  if (tid < NC) {
    const double s = ((lds[0][tid] + lds[1][tid]) + (lds[2][tid] + lds[3][tid]));
    if constexpr (FUSE) store_agent(partials + (int64_t)tid * kMaxRedBlocks + blockIdx.x, s);
    else partials[(int64_t)tid * kMaxRedBlocks + blockIdx.x] = s;
  }
  if constexpr (FUSE) {
    __shared__ int last;
    __threadfence();                       // release: this workgroup's partials before its ticket
    __syncthreads();
    if (tid == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();                       // acquire: every other workgroup's partials after the last ticket
    const int nblocks = (int)gridDim.x;
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {         // same order as finalize_kernel: lane t adds t, t+256, ..., fixed tree
      const double *p = partials + (int64_t)c * kMaxRedBlocks;
      double s = 0.0;
      for (int i = tid; i < nblocks; i += kBlock) s += load_agent(p + i);
      s = wave_sum(s);
      __syncthreads();                     // lds reuse
      if (lane == 0) lds[wave][0] = s;
      __syncthreads();
      if (tid == 0) out[c] = (lds[0][0] + lds[1][0]) + (lds[2][0] + lds[3][0]);
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
  }
}

// One workgroup per column; lane t sums partials t, t+256, ... sequentially, then a fixed tree.
__global__ void __launch_bounds__(kBlock)
finalize_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ out) {
  const int c = blockIdx.x, tid = threadIdx.x;
  const double *p = partials + (int64_t)c * kMaxRedBlocks;
  double s = 0.0;
  for (int i = tid; i < nblocks; i += kBlock) s += p[i];
  s = wave_sum(s);
  __shared__ double lds[kBlock / kWave];
  if ((tid & 63) == 0) lds[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) out[c] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

int32_t finalize_and_reduce(mxlo_ctx *ctx, int ncols, int nblocks, double *out_dev) {
  hipLaunchKernelGGL(finalize_kernel, dim3(ncols), dim3(kBlock), 0, ctx->stream, ctx->partials,
                     nblocks, out_dev);
  MXLO_LAUNCH_CHECK();
  return MXLO_OK;
}

// finalize_kernel that also POSTS its results to mapped pinned host memory as (value, sequence number) pairs
// `extra_n` .. `extra_n + ncols - 1`; workgroup 0 posts `extra_n` doubles an earlier kernel left in `extra_src` as pairs
// 0 .. extra_n - 1 (the L-SR1 push!'s six decision doubles come from two finalize launches: this is the second).
__global__ void __launch_bounds__(kBlock)
finalize_post_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ out, double *post,
                     unsigned long long seq, const double *__restrict__ extra_src, int extra_n) {
  const int c = blockIdx.x, tid = threadIdx.x;
  auto post_pair = [&](int i, double v) {
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(post) + 2 * i;
    __hip_atomic_store(slot, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(slot + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  const double *p = partials + (int64_t)c * kMaxRedBlocks;
  double s = 0.0;
  for (int i = tid; i < nblocks; i += kBlock) s += p[i];
  s = wave_sum(s);
  __shared__ double lds[kBlock / kWave];
  if ((tid & 63) == 0) lds[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    const double v = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    out[c] = v;
    post_pair(extra_n + c, v);
  }
  if (c == 0 && tid >= kWave && tid < kWave + extra_n) post_pair(tid - kWave, extra_src[tid - kWave]);
}

int32_t finalize_and_post(mxlo_ctx *ctx, int ncols, int nblocks, double *out_dev, double *post_dev, unsigned long long seq,
                          const double *extra_src, int extra_n) {
  MXLO_REQUIRE(post_dev && extra_n >= 0 && extra_n + ncols <= 8 && extra_n <= kWave, MXLO_EINVAL, "finalize_and_post: bad arguments");
  hipLaunchKernelGGL(finalize_post_kernel, dim3(ncols), dim3(kBlock), 0, ctx->stream, ctx->partials, nblocks, out_dev,
                     post_dev, seq, extra_src, extra_n);
  MXLO_LAUNCH_CHECK();
  return MXLO_OK;
}

int32_t allreduce_hook(mxlo_ctx *ctx, double *dev, int64_t count) {
  if (ctx->allreduce && count > 0) {
    int32_t s = ctx->allreduce(ctx->allreduce_user, dev, count, (void *)ctx->stream);
    MXLO_REQUIRE(s == 0, MXLO_EREDUCE, "all-reduce hook returned %d", s);
  }
  return MXLO_OK;
}

template <typename T, int VEC, int NC, bool XVEC>
static int32_t launch_dots(mxlo_ctx *ctx, const T *const *cols, const T *x, int64_t head,
                           int64_t nvec, int64_t n, int *nblocks_out, double *fused_out = nullptr) {
  ColPtrs<T, NC> cp;
  for (int c = 0; c < NC; ++c) cp.p[c] = cols[c];
  // fewer columns -> more chunks in flight per lane to keep ~the same bytes in flight
  constexpr int UNROLL = NC <= 2 ? 4 : (NC <= 6 ? 2 : 1);
  const bool nt = (int64_t)sizeof(T) * n * (NC + 1) >= ctx->tune.nt_min_bytes;
  // HBM-sized dots of one or two columns (the Householder dot pass): ONE workgroup per CU — 595 vs 606 us for the whole
  // n = 1e8 apply against the 4 per CU the wide panels want (tools/sweep_red_grid.py, profiles/r05_sweep_red_grid.txt;
  // a read-only pass of 1-2 streams does not gain from more workgroups, profiles/r05_tune_read.txt)
  const int per_cu = (nt && NC <= 2 && ctx->tune.red_blocks_per_cu > 1) ? 1 : ctx->tune.red_blocks_per_cu;
  const int grid = grid_for(ctx, nvec, (int64_t)kBlock * UNROLL, per_cu);
  if constexpr (NC <= kFuseMaxCols) {
    if (fused_out && grid <= kFuseMaxGrid) {
      if (nt)
        hipLaunchKernelGGL((panel_dots_kernel<T, VEC, NC, UNROLL, XVEC, true, true>), dim3(grid), dim3(kBlock), 0,
                           ctx->stream, cp, x, head, nvec, n, ctx->partials, ctx->ticket, fused_out);
      else
        hipLaunchKernelGGL((panel_dots_kernel<T, VEC, NC, UNROLL, XVEC, false, true>), dim3(grid), dim3(kBlock), 0,
                           ctx->stream, cp, x, head, nvec, n, ctx->partials, ctx->ticket, fused_out);
      MXLO_LAUNCH_CHECK();
      *nblocks_out = 0;                     // finalized in the kernel
      return MXLO_OK;
    }
  }
  if (nt)
    hipLaunchKernelGGL((panel_dots_kernel<T, VEC, NC, UNROLL, XVEC, true>), dim3(grid), dim3(kBlock), 0,
                       ctx->stream, cp, x, head, nvec, n, ctx->partials);
  else
    hipLaunchKernelGGL((panel_dots_kernel<T, VEC, NC, UNROLL, XVEC, false>), dim3(grid), dim3(kBlock), 0,
                       ctx->stream, cp, x, head, nvec, n, ctx->partials);
  MXLO_LAUNCH_CHECK();
  *nblocks_out = grid;
  return MXLO_OK;
}

#define DOTS_CASE(NCV)                                                                           \
  case NCV:                                                                                      \
    if (vec && xvec) return launch_dots<T, Vec16<T>::N, NCV, true>(ctx, cols, x, head, nvec, n, nb, fo); \
    if (vec) return launch_dots<T, Vec16<T>::N, NCV, false>(ctx, cols, x, head, nvec, n, nb, fo); \
    break;

template <typename T>
static int32_t dots_chunk(mxlo_ctx *ctx, const T *const *cols, int nc, const T *x, int64_t n,
                          int *nb, double *fo = nullptr) {
  constexpr int VEC = Vec16<T>::N;
  // the panel columns must share a 16-byte phase for the vector path
  int64_t head = 0;
  bool vec = n >= 4 * VEC;
  if (vec) {
    int64_t mis = -1;
    for (int c = 0; c < nc && vec; ++c) {
      int64_t m = (int64_t)((uintptr_t)cols[c] & 15u);
      if (m % (int64_t)sizeof(T)) vec = false;
      else if (mis < 0) mis = m;
      else if (mis != m) vec = false;
    }
    if (vec) head = mis == 0 ? 0 : (16 - mis) / (int64_t)sizeof(T);
  }
  const int64_t nvec = vec ? (n - head) / VEC : n;
  const bool xvec = vec && ((((uintptr_t)(x + head)) & 15u) == 0);
  switch (nc) {
    DOTS_CASE(1) DOTS_CASE(2) DOTS_CASE(3) DOTS_CASE(4) DOTS_CASE(5) DOTS_CASE(6) DOTS_CASE(7)
    DOTS_CASE(8) DOTS_CASE(9) DOTS_CASE(10) DOTS_CASE(11) DOTS_CASE(12) DOTS_CASE(13)
    DOTS_CASE(14) DOTS_CASE(15) DOTS_CASE(16) DOTS_CASE(17) DOTS_CASE(18) DOTS_CASE(19)
    DOTS_CASE(20)
    default: break;
  }
  // scalar fallback (mis-phased columns): at most 4 columns per launch
  switch (nc) {
    case 1: return launch_dots<T, 1, 1, false>(ctx, cols, x, 0, n, n, nb, fo);
    case 2: return launch_dots<T, 1, 2, false>(ctx, cols, x, 0, n, n, nb, fo);
    case 3: return launch_dots<T, 1, 3, false>(ctx, cols, x, 0, n, n, nb, fo);
    case 4: return launch_dots<T, 1, 4, false>(ctx, cols, x, 0, n, n, nb, fo);
    default: break;
  }
  set_error("dots_chunk: unsupported column count %d", nc);
  return MXLO_EINVAL;
}

// out_dev[c] = dot(cols[c], x), c < ncols (ncols <= kMaxRedCols); f64 results.
template <typename T>
int32_t panel_dots(mxlo_ctx *ctx, const T *const *cols, int ncols, const T *x, int64_t n,
                   double *out_dev) {
  MXLO_REQUIRE(ncols >= 0 && ncols <= kMaxRedCols, MXLO_EINVAL, "panel_dots: %d columns", ncols);
  if (ncols == 0) return MXLO_OK;
  if (n <= 0) {
    MXLO_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * ncols, ctx->stream));
    return allreduce_hook(ctx, out_dev, ncols);
  }
  constexpr int VEC = Vec16<T>::N;
  int done = 0;
  while (done < ncols) {
    int nc = ncols - done;
    if (nc > ctx->tune.dots_max_nc) nc = ctx->tune.dots_max_nc;
    // decide whether the vector path is available for this chunk; if not, cap at 4 columns
    bool vec = n >= 4 * VEC;
    int64_t mis = -1;
    for (int c = 0; c < nc && vec; ++c) {
      int64_t m = (int64_t)((uintptr_t)cols[done + c] & 15u);
      if (m % (int64_t)sizeof(T) || (mis >= 0 && mis != m)) vec = false;
      mis = m;
    }
    if (!vec && nc > 4) nc = 4;
    int nb = 0;
    // few columns in ONE launch: the last workgroup finalizes in-kernel (no separate finalize launch)
    const bool fuse = ctx->tune.fuse_finalize && done == 0 && nc == ncols && nc <= kFuseMaxCols;
    MXLO_TRY(dots_chunk<T>(ctx, cols + done, nc, x, n, &nb, fuse ? out_dev : nullptr));
    if (nb > 0) MXLO_TRY(finalize_and_reduce(ctx, nc, nb, out_dev + done));
    done += nc;
  }
  return allreduce_hook(ctx, out_dev, ncols);
}

// The dots kernel only: per-workgroup partial sums of dot(cols[c], x) are left in ctx->partials ([c][kMaxRedBlocks],
// *nblocks of them per column) for a consumer that adds them up itself (ncols <= dots_max_nc: one chunk; n >= 1).
template <typename T>
int32_t panel_dots_partials(mxlo_ctx *ctx, const T *const *cols, int ncols, const T *x, int64_t n, int *nblocks) {
  MXLO_REQUIRE(ncols >= 1 && ncols <= 4 && n >= 1, MXLO_EINVAL, "panel_dots_partials: bad arguments");
  return dots_chunk<T>(ctx, cols, ncols, x, n, nblocks, nullptr);
}
template int32_t panel_dots_partials<double>(mxlo_ctx *, const double *const *, int, const double *, int64_t, int *);
template int32_t panel_dots_partials<float>(mxlo_ctx *, const float *const *, int, const float *, int64_t, int *);

// ---- dual-x panel dots: out1[c] = dot(col_c, x1), out2[c] = dot(col_c, x2) in ONE pass over the columns.
// push! needs S's_new and S'y_new (forward) / Y'y_new and Y's_new (inverse): reading the panel once instead of
// twice. Operands are panel columns (16-byte aligned, zero-padded to a whole number of vectors), so there is no
// head/tail handling: `nvec` covers the padding, which contributes 0*0.
template <typename T, int NC, int UNROLL, bool NT>
__global__ void __launch_bounds__(kBlock)
panel_dots2_kernel(ColPtrs<T, NC> cols, const T *__restrict__ x1, const T *__restrict__ x2, int64_t nvec,
                   double *__restrict__ partials) {
  constexpr int VEC = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const int tid = threadIdx.x;
  double a1[NC], a2[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) a1[c] = a2[c] = 0.0;
  auto ld = [&](const T *p, int64_t i) -> V {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(p) + i) : reinterpret_cast<const V *>(p)[i];
  };
  constexpr int64_t CHUNK = (int64_t)kBlock * UNROLL;
  const int64_t nchunks = (nvec + CHUNK - 1) / CHUNK;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t base = ch * CHUNK + tid;
    V xv1[UNROLL], xv2[UNROLL], cv[UNROLL][NC];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = base + (int64_t)u * kBlock;
      ok[u] = i < nvec;
      if (ok[u]) {
        xv1[u] = ld(x1, i);
        xv2[u] = ld(x2, i);
#pragma unroll
        for (int c = 0; c < NC; ++c) cv[u][c] = ld(cols.p[c], i);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (!ok[u]) continue;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const double ce = (double)cv[u][c][e];
          a1[c] = fma(ce, (double)xv1[u][e], a1[c]);
          a2[c] = fma(ce, (double)xv2[u][e], a2[c]);
        }
      }
    }
  }
  __shared__ double lds[kBlock / kWave][2 * NC];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const double s1 = wave_sum(a1[c]), s2 = wave_sum(a2[c]);
    if (lane == 0) {
      lds[wave][c] = s1;
      lds[wave][NC + c] = s2;
    }
  }
  __syncthreads();
  if (tid < 2 * NC)
    partials[(int64_t)tid * kMaxRedBlocks + blockIdx.x] = (lds[0][tid] + lds[1][tid]) + (lds[2][tid] + lds[3][tid]);
}

// cols, x1, x2: 16-byte aligned panel columns of padded length `nvec_elems` (a multiple of the vector width).
template <typename T>
int32_t panel_dots2(mxlo_ctx *ctx, const T *const *cols, int ncols, const T *x1, const T *x2, int64_t n_padded,
                    double *out1, double *out2) {
  constexpr int VEC = Vec16<T>::N;
  MXLO_REQUIRE(ncols >= 0 && ncols <= kMaxRedCols / 2 && n_padded % VEC == 0, MXLO_EINVAL, "panel_dots2: bad arguments");
  bool aligned = (((uintptr_t)x1 | (uintptr_t)x2) & 15u) == 0;
  for (int c = 0; c < ncols; ++c) aligned = aligned && (((uintptr_t)cols[c]) & 15u) == 0;
  MXLO_REQUIRE(aligned, MXLO_EINVAL, "panel_dots2: operands must be 16-byte aligned panel columns");
  const int64_t nvec = n_padded / VEC;
  int done = 0;
  while (done < ncols) {
    const int nc = ncols - done >= 10 ? 10 : ncols - done;
    const bool nt = (int64_t)sizeof(T) * n_padded * (nc + 2) >= ctx->tune.nt_min_bytes;
    int grid = 0;
    auto go = [&]<int NC>() {
      ColPtrs<T, NC> cp;
      for (int c = 0; c < NC; ++c) cp.p[c] = cols[done + c];
      constexpr int UNROLL = NC <= 2 ? 4 : (NC <= 5 ? 2 : 1);
      grid = grid_for(ctx, nvec, (int64_t)kBlock * UNROLL, ctx->tune.red_blocks_per_cu);
      if (grid > kMaxRedBlocks) grid = kMaxRedBlocks;
      if (nt)
        hipLaunchKernelGGL((panel_dots2_kernel<T, NC, UNROLL, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, cp, x1,
                           x2, nvec, ctx->partials);
      else
        hipLaunchKernelGGL((panel_dots2_kernel<T, NC, UNROLL, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, cp, x1,
                           x2, nvec, ctx->partials);
    };
    switch (nc) {
      case 1: go.template operator()<1>(); break;
      case 2: go.template operator()<2>(); break;
      case 3: go.template operator()<3>(); break;
      case 4: go.template operator()<4>(); break;
      case 5: go.template operator()<5>(); break;
      case 6: go.template operator()<6>(); break;
      case 7: go.template operator()<7>(); break;
      case 8: go.template operator()<8>(); break;
      case 9: go.template operator()<9>(); break;
      default: go.template operator()<10>(); break;
    }
    MXLO_LAUNCH_CHECK();
    // partial columns [0, nc) belong to x1, [nc, 2nc) to x2
    hipLaunchKernelGGL(finalize_kernel, dim3(nc), dim3(kBlock), 0, ctx->stream, ctx->partials, grid, out1 + done);
    MXLO_LAUNCH_CHECK();
    hipLaunchKernelGGL(finalize_kernel, dim3(nc), dim3(kBlock), 0, ctx->stream,
                       ctx->partials + (int64_t)nc * kMaxRedBlocks, grid, out2 + done);
    MXLO_LAUNCH_CHECK();
    done += nc;
  }
  MXLO_TRY(allreduce_hook(ctx, out1, ncols));
  return allreduce_hook(ctx, out2, ncols);
}

// ---- push! pass: dual-x dots over one panel WITH the new pair held per lane ----------------------------------------
// One-pass push!: the two-kernel schedule copied s -> S[:,ins], y -> Y[:,ins] (two device-to-device
// copies, 4 vector passes) and then took the Gram rows as dual-x dots against the freshly copied panel columns,
// re-reading them. Here a pass over a panel streams every OTHER column once with s_new (x1) and y_new (x2) in
// registers; the column being replaced is never loaded — its values ARE x1 (S panel) or x2 (Y panel) — and the pass
// that runs after push!'s accept/reject decision stores x1 / x2 (/ b = x2 ./ sqrt(ys), src/lbfgs.jl:232) into the
// slot while they are in registers. The decision scalars y's and y'y come out of the FIRST pass (x1·x2, x2·x2 from
// the same registers), so the separate 2-vector dots pass disappears as well. Same chunk / lane / tree decomposition
// as panel_dots2_kernel: the Gram rows are bit-identical to the two-kernel schedule's (tested).
template <typename T, int NC>
struct PushPassArgs {
  const T *cols[NC];
  const T *x1, *x2;        // caller vectors (length n, 16-byte aligned; NOT padded)
  T *st1, *st2, *stb;      // optional stores: x1 -> st1, x2 -> st2, x2 ./ sq -> stb (padded panel columns)
  T sq;
  int slot, slot_src;      // column index replaced by x1 (slot_src 1) / x2 (2); -1: none
  int64_t nvec;            // vectors per padded column
  int64_t n;               // valid elements of x1, x2
};

// FAST: the launch-bound instantiation (every workgroup has at most one chunk) with the branch-free chunk loads below. It
// needs ~45 more VGPRs (5 columns: 102 -> 148, one wave per SIMD less), which costs the HBM-bound sizes 6 % (inverse
// m = 10, n = 5e7: 1.64 -> 1.77 ms) — those keep the guarded loads, whose latency their occupancy hides.
template <typename T, int NC, int UNROLL, bool NT, bool STORE, bool FAST = false>
__global__ void __launch_bounds__(kBlock)
push_pass_kernel(PushPassArgs<T, NC> A, double *__restrict__ partials) {
  constexpr int VEC = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const int tid = threadIdx.x;
  double a1[NC], a2[NC], e12 = 0.0, e22 = 0.0, ebb = 0.0, e11 = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) a1[c] = a2[c] = 0.0;
  auto ld = [&](const T *p, int64_t i) -> V {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(p) + i) : reinterpret_cast<const V *>(p)[i];
  };
  const int64_t nfull = A.n / VEC;   // whole vectors of x1 / x2; vector nfull (if any) is partial, the rest padding
  auto ldx = [&](const T *p, int64_t i) -> V {
    if (i < nfull) return ld(p, i);
    V v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = (i * VEC + e < A.n) ? p[i * VEC + e] : T(0);   // zero padding, as in the panel
    return v;
  };
  constexpr int64_t CHUNK = (int64_t)kBlock * UNROLL;
  const int64_t nchunks = (A.nvec + CHUNK - 1) / CHUNK;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t base = ch * CHUNK + tid;
    V xv1[UNROLL], xv2[UNROLL], cv[UNROLL][NC];
    bool ok[UNROLL];
    if (FAST && (ch + 1) * CHUNK <= nfull) {
      // Whole chunk inside the valid range (every chunk but the last): branch-free, so the (NC + 2) * UNROLL loads of a
      // lane really are in flight together. (The guarded form below waits for each load before issuing the next —
      // the ISA had an s_waitcnt vmcnt(0) in front of every one of them; at launch-bound sizes that was 14 memory
      // round trips for 5 columns, 44 for 20.) The column being replaced is still never read: its pointer is
      // swapped for the caller vector that stands for it (a cache hit, and the accumulate step selects that vector anyway).
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * kBlock;
        ok[u] = true;
        xv1[u] = ld(A.x1, i);
        xv2[u] = ld(A.x2, i);
#pragma unroll
        for (int c = 0; c < NC; ++c) cv[u][c] = ld(c == A.slot ? (A.slot_src == 1 ? A.x1 : A.x2) : A.cols[c], i);
      }
    } else {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * kBlock;
        ok[u] = i < A.nvec;
        if (ok[u]) {
          xv1[u] = ldx(A.x1, i);
          xv2[u] = ldx(A.x2, i);
#pragma unroll
          for (int c = 0; c < NC; ++c)
            if (c != A.slot) cv[u][c] = ld(A.cols[c], i);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (!ok[u]) continue;
      const int64_t i = base + (int64_t)u * kBlock;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const V col = (c == A.slot) ? (A.slot_src == 1 ? xv1[u] : xv2[u]) : cv[u][c];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const double ce = (double)col[e];
          a1[c] = fma(ce, (double)xv1[u][e], a1[c]);
          a2[c] = fma(ce, (double)xv2[u][e], a2[c]);
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        e12 = fma((double)xv1[u][e], (double)xv2[u][e], e12);
        e22 = fma((double)xv2[u][e], (double)xv2[u][e], e22);
        e11 = fma((double)xv1[u][e], (double)xv1[u][e], e11);
      }
      if constexpr (STORE) {
        if (A.st1) __builtin_nontemporal_store(xv1[u], reinterpret_cast<V *>(A.st1) + i);
        if (A.st2) __builtin_nontemporal_store(xv2[u], reinterpret_cast<V *>(A.st2) + i);
        if (A.stb) {
          V b;
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            b[e] = xv2[u][e] / A.sq;                       // b[insert] = y ./ sqrt(ys): a true division per element
            ebb = fma((double)b[e], (double)b[e], ebb);
          }
          __builtin_nontemporal_store(b, reinterpret_cast<V *>(A.stb) + i);
        }
      }
    }
  }
  constexpr int NP = 2 * NC + 4;
  __shared__ double lds[kBlock / kWave][NP];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const double s1 = wave_sum(a1[c]), s2 = wave_sum(a2[c]);
    if (lane == 0) {
      lds[wave][c] = s1;
      lds[wave][NC + c] = s2;
    }
  }
  {
    const double s12 = wave_sum(e12), s22 = wave_sum(e22), sbb = wave_sum(ebb), s11 = wave_sum(e11);
    if (lane == 0) {
      lds[wave][2 * NC] = s12;
      lds[wave][2 * NC + 1] = s22;
      lds[wave][2 * NC + 2] = sbb;
      lds[wave][2 * NC + 3] = s11;
    }
  }
  __syncthreads();
  if (tid < NP)
    partials[(int64_t)tid * kMaxRedBlocks + blockIdx.x] = (lds[0][tid] + lds[1][tid]) + (lds[2][tid] + lds[3][tid]);
}

// one workgroup per partial column, destination per column from a table (NULL: result not wanted)
constexpr int kPushMaxNC = 20;
struct FinalizeMap {
  double *dst[2 * kPushMaxNC + 4];
  PushExtras ex;          // common.h: a small copy (workgroup 0) and the posting of columns post_col, post_col + 1
  int post_col = -1;
};
__global__ void __launch_bounds__(kBlock)
finalize_map_kernel(const double *__restrict__ partials, int nblocks, FinalizeMap M) {
  const int c = blockIdx.x, tid = threadIdx.x;
  if (c == 0)
    for (int i = tid; i < M.ex.cp_n; i += kBlock) M.ex.cp_dst[i] = M.ex.cp_src[i];
  double *dst = M.dst[c];
  if (!dst) return;
  const double *p = partials + (int64_t)c * kMaxRedBlocks;
  double s = 0.0;
  for (int i = tid; i < nblocks; i += kBlock) s += p[i];
  s = wave_sum(s);
  __shared__ double lds[kBlock / kWave];
  if ((tid & 63) == 0) lds[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    const double v = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    *dst = v;
    if (M.ex.post && (c == M.post_col || c == M.post_col + 1)) {   // value first, its sequence number behind it
      unsigned long long *slot = reinterpret_cast<unsigned long long *>(M.ex.post) + 2 * (c - M.post_col);
      __hip_atomic_store(slot, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      __hip_atomic_store(slot + 1, M.ex.post_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// One pass over `ncols` (1..10, or exactly 20) padded panel columns. out1[c] = dot(col_c, x1), out2[c] = dot(col_c, x2) with column
// `slot` standing for x1 (slot_src 1) or x2 (2); out_x1x2 / out_x2x2 / out_bb / out_x1x1 (each may be NULL) receive
// x1·x2, x2·x2, |x2 ./ sq|^2 and x1·x1. The caller runs the all-reduce hook on whatever it keeps (results are LOCAL sums here).
template <typename T>
int32_t panel_push_pass(mxlo_ctx *ctx, const T *const *cols, int ncols, int slot, int slot_src, const T *x1,
                        const T *x2, int64_t n, int64_t n_padded, T *st1, T *st2, T *stb, double sq, double *out1,
                        double *out2, double *out_x1x2, double *out_x2x2, double *out_bb, double *out_x1x1,
                        const PushExtras *extras) {
  constexpr int VEC = Vec16<T>::N;
  MXLO_REQUIRE(!extras || !extras->post || (out_x1x2 && out_x2x2), MXLO_EINVAL, "panel_push_pass: posting needs x1.x2 and x2.x2");
  MXLO_REQUIRE(((ncols >= 1 && ncols <= 10) || ncols == kPushMaxNC) && n_padded % VEC == 0 && n <= n_padded &&
                   n > n_padded - VEC, MXLO_EINVAL, "panel_push_pass: bad arguments");
  bool aligned = (((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)st1 | (uintptr_t)st2 | (uintptr_t)stb) & 15u) == 0;
  for (int c = 0; c < ncols; ++c) aligned = aligned && (((uintptr_t)cols[c]) & 15u) == 0;
  MXLO_REQUIRE(aligned, MXLO_EINVAL, "panel_push_pass: operands must be 16-byte aligned");
  const int64_t nvec = n_padded / VEC;
  const bool store = st1 || st2 || stb;
  const bool nt = (int64_t)sizeof(T) * n_padded * (ncols + 2) >= ctx->tune.nt_min_bytes;
  int grid = 0;
  auto go = [&]<int NC>() {
    PushPassArgs<T, NC> A;
    for (int c = 0; c < NC; ++c) A.cols[c] = cols[c];
    A.x1 = x1; A.x2 = x2; A.st1 = st1; A.st2 = st2; A.stb = stb; A.sq = (T)sq;
    A.slot = slot; A.slot_src = slot_src; A.nvec = nvec; A.n = n;
    constexpr int UNROLL = NC <= 2 ? 4 : (NC <= 5 ? 2 : 1);      // = panel_dots2: identical per-lane summation order
    grid = grid_for(ctx, nvec, (int64_t)kBlock * UNROLL, ctx->tune.red_blocks_per_cu);
    if (grid > kMaxRedBlocks) grid = kMaxRedBlocks;
    auto launch = [&]<bool NTV, bool ST, bool FAST>() {
      hipLaunchKernelGGL((push_pass_kernel<T, NC, UNROLL, NTV, ST, FAST>), dim3(grid), dim3(kBlock), 0, ctx->stream, A,
                         ctx->partials);
    };
    const int64_t nchunks = (nvec + (int64_t)kBlock * UNROLL - 1) / ((int64_t)kBlock * UNROLL);
    if (nchunks <= grid && !nt) {   // launch-bound: one chunk per workgroup
      if (store) launch.template operator()<false, true, true>(); else launch.template operator()<false, false, true>();
    } else if (nt) {
      if (store) launch.template operator()<true, true, false>(); else launch.template operator()<true, false, false>();
    } else {
      if (store) launch.template operator()<false, true, false>(); else launch.template operator()<false, false, false>();
    }
    FinalizeMap M;
    for (int c = 0; c < 2 * kPushMaxNC + 4; ++c) M.dst[c] = nullptr;
    for (int c = 0; c < NC; ++c) {
      M.dst[c] = out1 ? out1 + c : nullptr;
      M.dst[NC + c] = out2 ? out2 + c : nullptr;
    }
    M.dst[2 * NC] = out_x1x2;
    M.dst[2 * NC + 1] = out_x2x2;
    M.dst[2 * NC + 2] = out_bb;
    M.dst[2 * NC + 3] = out_x1x1;
    if (extras) {
      M.ex = *extras;
      M.post_col = 2 * NC;
    }
    hipLaunchKernelGGL(finalize_map_kernel, dim3(2 * NC + 4), dim3(kBlock), 0, ctx->stream, ctx->partials, grid, M);
  };
  switch (ncols) {
    case 1: go.template operator()<1>(); break;
    case 2: go.template operator()<2>(); break;
    case 3: go.template operator()<3>(); break;
    case 4: go.template operator()<4>(); break;
    case 5: go.template operator()<5>(); break;
    case 6: go.template operator()<6>(); break;
    case 7: go.template operator()<7>(); break;
    case 8: go.template operator()<8>(); break;
    case 9: go.template operator()<9>(); break;
    case 10: go.template operator()<10>(); break;
    default: go.template operator()<kPushMaxNC>(); break;
  }
  MXLO_LAUNCH_CHECK();
  return MXLO_OK;
}

template int32_t panel_push_pass<double>(mxlo_ctx *, const double *const *, int, int, int, const double *, const double *,
                                         int64_t, int64_t, double *, double *, double *, double, double *, double *,
                                         double *, double *, double *, double *, const PushExtras *);
template int32_t panel_push_pass<float>(mxlo_ctx *, const float *const *, int, int, int, const float *, const float *,
                                        int64_t, int64_t, float *, float *, float *, double, double *, double *, double *,
                                        double *, double *, double *, const PushExtras *);

template int32_t panel_dots2<double>(mxlo_ctx *, const double *const *, int, const double *, const double *, int64_t,
                                     double *, double *);
template int32_t panel_dots2<float>(mxlo_ctx *, const float *const *, int, const float *, const float *, int64_t,
                                    double *, double *);

template int32_t panel_dots<double>(mxlo_ctx *, const double *const *, int, const double *, int64_t,
                                    double *);
template int32_t panel_dots<float>(mxlo_ctx *, const float *const *, int, const float *, int64_t,
                                   double *);

}  // namespace mxlo
