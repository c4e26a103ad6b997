// dense.hip — matrix-carrying leaves: dense LinearOperator(M) (GEMV), opHermitian, kron(A,B).
//
// kron(A,B)*x = vec(B * X * A^T) is two GEMMs on the matrix cores: f64 uses
// v_mfma_f64_16x16x4_f64 (one f64 of A and B per lane; C/D layout col = lane&15,
// row = (lane>>4) + 4*reg — NOT the f32 map), f32 uses v_mfma_f32_16x16x4_f32
// (row = 4*(lane>>4) + reg). The production kernel is gemm_glds.h (LDS-DMA ring, swizzled LDS images, counted
// vmcnt, mid-slab barrier); `gemm_kernel` below is the fallback for operands the 16-byte DMA path cannot take
// (odd leading dimensions / extents, misaligned views).
#include "common.h"
#include "gemm_glds.h"
#include "stream_kernels.h"

using namespace mxlo;

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LDT = 80;  // padded tile row (elements): 80*8 B = 640 B -> second k-row lands 32 banks away

template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
  using Acc = f64x4;
  static __device__ __forceinline__ Acc run(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float> {
  using Acc = f32x4v;
  static __device__ __forceinline__ Acc run(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// C (M x N, ldc) = alpha * opA(A) (M x K) * opB(B) (K x N) (+ beta * C); column-major.
// TA: A is stored K x M (we need its transpose); TB: B is stored N x K.
template <typename T, typename CA, typename CB, bool TA, bool TB, bool BETA0>
__global__ void __launch_bounds__(kBlock)
gemm_kernel(T *__restrict__ C, int64_t ldc, const T *__restrict__ A, int64_t lda,
            const T *__restrict__ B, int64_t ldb, int M, int N, int K, CA alpha, CB beta) {
  __shared__ T sA[BK][LDT];  // sA[k][i]
  __shared__ T sB[BK][LDT];  // sB[k][j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bm = blockIdx.x * BM, bn = blockIdx.y * BN;
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  using Acc = typename Mfma<T>::Acc;
  Acc acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0;

  for (int k0 = 0; k0 < K; k0 += BK) {
    // ---- stage A tile (BM x BK) and B tile (BK x BN): 1024 elements each, 4 per thread
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = tid + t * kBlock;  // 0..1023
      {  // A
        int i, k;
        if constexpr (!TA) { i = e & 63; k = e >> 6; }   // i contiguous in memory
        else { k = e & 15; i = e >> 4; }                 // k contiguous in memory
        const int gi = bm + i, gk = k0 + k;
        T v = 0;
        if (gi < M && gk < K) v = TA ? A[gk + (int64_t)gi * lda] : A[gi + (int64_t)gk * lda];
        sA[k][i] = v;
      }
      {  // B
        int j, k;
        if constexpr (!TB) { k = e & 15; j = e >> 4; }   // k contiguous in memory
        else { j = e & 63; k = e >> 6; }                 // j contiguous in memory
        const int gj = bn + j, gk = k0 + k;
        T v = 0;
        if (gj < N && gk < K) v = TB ? B[gj + (int64_t)gk * ldb] : B[gk + (int64_t)gj * ldb];
        sB[k][j] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      const int kr = kk + (lane >> 4);
      T a0 = sA[kr][wm + (lane & 15)], a1 = sA[kr][wm + 16 + (lane & 15)];
      T b0 = sB[kr][wn + (lane & 15)], b1 = sB[kr][wn + 16 + (lane & 15)];
      acc[0][0] = Mfma<T>::run(a0, b0, acc[0][0]);
      acc[0][1] = Mfma<T>::run(a0, b1, acc[0][1]);
      acc[1][0] = Mfma<T>::run(a1, b0, acc[1][0]);
      acc[1][1] = Mfma<T>::run(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
  // ---- epilogue: res = (alpha*acc) (+ beta*C), each product rounded separately (no FMA)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = bm + wm + a * 16 + Mfma<T>::row(lane, r);
        const int gj = bn + wn + b * 16 + (lane & 15);
        if (gi < M && gj < N) {
          T *p = C + gi + (int64_t)gj * ldc;
          *p = fin_ab<T, CA, CB, BETA0>(alpha * (CA)acc[a][b][r], beta, BETA0 ? T(0) : *p);
        }
      }
}

// C (M x N) = alpha * opA(A) * opB(B) (+ beta*C). tb == true (B' stored N x K, N-contiguous) with either A layout is
// the DMA kernel's shape — both products of kron's prod! AND of its tprod!/ctprod! have it. Tile choice
// (tools/tune_gemm.hip, profiles/r02_tune_gemm.txt): the largest tile that still gives every CU a workgroup.
template <typename T>
int32_t gemm(mxlo_ctx *ctx, T *C, int64_t ldc, const T *A, int64_t lda, bool ta, const T *B,
             int64_t ldb, bool tb, int64_t M, int64_t N, int64_t K, double alpha, double beta,
             int32_t flags) {
  if (M <= 0 || N <= 0) return MXLO_OK;
  MXLO_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), MXLO_ESHAPE, "gemm dims too large");
  // per-lane DMA offsets are 32-bit: one K-slab of either operand must span < 4 GiB
  const bool ld_ok = lda < (1LL << 22) && ldb < (1LL << 22) && ldc > 0;
  if (tb && K > 0 && ld_ok && ctx->tune.gemm_tile >= 0 && gemm_glds_ok<T>(A, lda, ta, B, ldb, M, N, K)) {
    return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
      auto tiles = [&](int tm, int tn) { return ((M + tm - 1) / tm) * ((N + tn - 1) / tn); };
      int tile = ctx->tune.gemm_tile;   // 0 = auto, else 32 / 64 / 128
      if (tile == 0) {
        if (tiles(128, 128) >= ctx->num_cu) tile = 128;
        else if (tiles(64, 64) * 5 >= (int64_t)ctx->num_cu * 3) tile = 64;   // >= 0.6 workgroups per CU
        else tile = 32;
      }
#define GLDS(AK_, TM_, TN_, WM_, WN_, BK_, NST_, PAIR_, PFD_, UNR_)                                               \
  {                                                                                                               \
    GlShape S{(int)M, (int)N, (int)K, (int)((M + TM_ - 1) / TM_), (int)((N + TN_ - 1) / TN_)};                    \
    hipLaunchKernelGGL((gemm_glds_kernel<T, CA, CB, B0, AK_, TM_, TN_, WM_, WN_, BK_, NST_, true, true, PAIR_, PFD_, true, false, false, UNR_>), \
                       dim3(S.gx * S.gy), dim3(WM_ * WN_ * 64), 0, ctx->stream, C, ldc, A, lda, B, ldb, S,        \
                       (CA)alpha, (CB)beta);                                                                      \
  }
#define GLDS_BY_A(TM_, TN_, WM_, WN_, BK_, NST_, PAIR_, PFD_, UNR_AM_)                                            \
  if (ta) GLDS(true, TM_, TN_, WM_, WN_, BK_, NST_, PAIR_, PFD_, true) else GLDS(false, TM_, TN_, WM_, WN_, BK_, NST_, PAIR_, PFD_, UNR_AM_)
      // PAIR (gemm_glds.h): tile pairs interleaved so that two fragments arrive with one 16-byte LDS read. Measured
      // (profiles/r03_tune_gemm_pair.txt): 64-tiles -2.0 % (f64) / -2.7 % (f32), f64 128-tiles -0.6 %, f32 128-tiles +0.4 %.
      // SWAPC (always on): the MFMA runs with its operands exchanged so that the accumulator holds the transposed tile and
      // 16 lanes store 16 consecutive rows of the column-major C (whole 128-byte lines in f64): 1024^2 f64 72.6 -> 70.4 us,
      // f32 43.1 -> 41.0 us, 2048^2 -1 ... -4 % (profiles/r03_tune_gemm_swapc.txt).
      // UNR (on except for the M-contiguous f32 64- / 128-tiles, where it measured +1 ... 2 %): the steady state runs NST slabs per trip with the ring index a compile-time constant, so every LDS
      // address is base + immediate and the vector ALU does no address work between the MFMAs: 1024^2 70.2 -> 68.9 us, the
      // transpose mode (XOR-swizzled K-contiguous A) 79.0 -> 69.8 us, 32-tiles -6 ... -8 %, 128-tiles at 2048^2 -2.9 %.
      // PFD 2 (fragments read two k-steps ahead): the 32-tile workgroup has ONE wave per SIMD, nothing else hides its LDS
      // latency: -0.5 ... -3 % there, +1 % on the 64-tiles (two waves per SIMD) — profiles/r03_tune_gemm_bk.txt
      if constexpr (sizeof(T) == 8) {
        if (tile == 128) GLDS_BY_A(128, 128, 4, 4, 16, 3, true, 1, true)
        else if (tile == 64) GLDS_BY_A(64, 64, 4, 2, 32, 3, true, 1, true)
        else GLDS_BY_A(32, 32, 2, 2, 32, 4, false, 2, true)
      } else {
        if (tile == 128) GLDS_BY_A(128, 128, 4, 4, 32, 3, false, 1, false)
        else if (tile == 64) GLDS_BY_A(64, 64, 4, 2, 64, 3, true, 1, false)
        else GLDS_BY_A(32, 32, 2, 2, 64, 4, false, 2, true)
      }
#undef GLDS_BY_A
#undef GLDS
      MXLO_LAUNCH_CHECK();
      return MXLO_OK;
    });
  }
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
#define GO(TA_, TB_)                                                                                  \
  hipLaunchKernelGGL((gemm_kernel<T, CA, CB, TA_, TB_, B0>), grid, dim3(kBlock), 0, ctx->stream, C,  \
                     ldc, A, lda, B, ldb, (int)M, (int)N, (int)K, (CA)alpha, (CB)beta)
    if (!ta && !tb) GO(false, false);
    else if (!ta && tb) GO(false, true);
    else if (ta && !tb) GO(true, false);
    else GO(true, true);
#undef GO
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

// ---- GEMV --------------------------------------------------------------------------------
// T mode (res[j] = alpha * dot(M[:,j], v) + beta*res[j]): one wave per column, coalesced down the column.
// PAIR: 16 bytes of rows per lane per load (2 rows f64, 4 rows f32), 4 loads of M in flight per lane,
// nontemporal (M is streamed once; v stays cached).
template <typename T, typename CA, typename CB, bool BETA0, bool PAIR, bool NT = true>
__global__ void __launch_bounds__(kBlock)
gemv_t_kernel(T *__restrict__ res, const T *__restrict__ M, int64_t m, int64_t n, int64_t ld,
              const T *__restrict__ v, CA alpha, CB beta) {
  constexpr int VR = 16 / (int)sizeof(T);     // rows per lane per 16-byte load: 2 (f64) or 4 (f32)
  typedef T VV __attribute__((ext_vector_type(VR)));
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / kWave);
  for (int64_t j = wave; j < n; j += nwaves) {
    const T *colp = M + j * ld;
    double acc0 = 0.0, acc1 = 0.0;
    int64_t i = 0;
    if constexpr (PAIR) {
      constexpr int U = 4;
      const int64_t mp = m / VR;  // 16-byte row groups
      const VV *cp = reinterpret_cast<const VV *>(colp);
      const VV *vp = reinterpret_cast<const VV *>(v);
      int64_t p = lane;
      for (; p + (int64_t)(U - 1) * 64 < mp; p += (int64_t)U * 64) {
        VV a[U], x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          a[u] = NT ? __builtin_nontemporal_load(cp + p + u * 64) : cp[p + u * 64];
          x[u] = vp[p + u * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int e = 0; e < VR; e += 2) {
            acc0 = fma((double)a[u][e], (double)x[u][e], acc0);
            acc1 = fma((double)a[u][e + 1], (double)x[u][e + 1], acc1);
          }
        }
      }
      for (; p < mp; p += 64) {
        const VV a = cp[p], x = vp[p];
#pragma unroll
        for (int e = 0; e < VR; e += 2) {
          acc0 = fma((double)a[e], (double)x[e], acc0);
          acc1 = fma((double)a[e + 1], (double)x[e + 1], acc1);
        }
      }
      i = mp * VR;
    }
    for (int64_t r = i + lane; r < m; r += 64) acc0 = fma((double)colp[r], (double)v[r], acc0);
    double acc = acc0 + acc1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) {
      res[j] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)acc, beta, BETA0 ? T(0) : res[j]);
    }
  }
}

// N mode partials: a lane owns one row (or a PAIR of rows: 16-byte loads for f64), a chunk of columns per
// blockIdx.y, 8 columns in flight; part[chunk][row].
template <typename T, bool PAIR>
__global__ void __launch_bounds__(kBlock)
gemv_n_partial_kernel(double *__restrict__ part, const T *__restrict__ M, int64_t m, int64_t n,
                      int64_t ld, const T *__restrict__ v, int64_t cols_per_chunk) {
  typedef T V2 __attribute__((ext_vector_type(2)));
  constexpr int R = PAIR ? 2 : 1;
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * R;
  const int64_t j0 = (int64_t)blockIdx.y * cols_per_chunk;
  int64_t j1 = j0 + cols_per_chunk;
  if (j1 > n) j1 = n;
  if (i >= m) return;
  double acc0 = 0.0, acc1 = 0.0;
  if (PAIR && i + 1 < m) {
    constexpr int U = 8;
    int64_t j = j0;
    for (; j + U <= j1; j += U) {
      V2 a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(M + i + (j + u) * ld));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double vj = (double)v[j + u];
        acc0 = fma((double)a[u][0], vj, acc0);
        acc1 = fma((double)a[u][1], vj, acc1);
      }
    }
    for (; j < j1; ++j) {
      const V2 a = *reinterpret_cast<const V2 *>(M + i + j * ld);
      const double vj = (double)v[j];
      acc0 = fma((double)a[0], vj, acc0);
      acc1 = fma((double)a[1], vj, acc1);
    }
    part[(int64_t)blockIdx.y * m + i] = acc0;
    part[(int64_t)blockIdx.y * m + i + 1] = acc1;
    return;
  }
  for (int64_t j = j0; j < j1; ++j) acc0 = fma((double)M[i + j * ld], (double)v[j], acc0);
  part[(int64_t)blockIdx.y * m + i] = acc0;
}

// 32 rows per workgroup, 8 lanes per row: lane `sub` adds chunks sub, sub+8, ... (independent loads in flight),
// the 8 sub-sums are combined in a fixed order -> deterministic, and m/32 workgroups instead of m/256.
template <typename T, typename CA, typename CB, bool BETA0>
__global__ void __launch_bounds__(kBlock)
gemv_n_finish_kernel(T *__restrict__ res, const double *__restrict__ part, int64_t m, int nchunks,
                     CA alpha, CB beta, T *__restrict__ raw_out) {
  const int r = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + r;
  __shared__ double sred[8][32];
  double a = 0.0;
  if (i < m)
    for (int c = sub; c < nchunks; c += 8) a += part[(int64_t)c * m + i];
  sred[sub][r] = a;
  __syncthreads();
  if (sub != 0 || i >= m) return;
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) acc += sred[q][r];
  if (raw_out) {
    raw_out[i] = (T)acc;
    return;
  }
  res[i] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)acc, beta, BETA0 ? T(0) : res[i]);
}

// N mode, ROW BANDS (round 5): a workgroup owns RB consecutive rows across ALL columns, so the sum over the columns
// never leaves the workgroup — one launch, no partial workspace, no dependent finish launch (at n = 4096 the finish was
// 5 of 28 us). LPR = RB / VR lanes cover the band's RB rows of one column with 16-byte loads (RB = 16 rows of f64 = one
// 128-byte line per column), the other lanes of the workgroup take other columns (NCL = 512 / LPR column lanes), 8 loads
// in flight per lane (64 KiB per workgroup). The NCL column-lane partials are added in a fixed order through LDS:
// deterministic. Grid = m / RB >= #CU / 2 (gemv_n picks RB = 16 / 32 / 64); at one moment the workgroups read the same
// few columns, each its own 128-byte-aligned piece: whole columns stream from HBM contiguously.
constexpr int kGemvRowsBlock = 512;
template <typename T, typename CA, typename CB, bool BETA0, int RB, bool NT = true>
__global__ void __launch_bounds__(kGemvRowsBlock)
gemv_n_rows_kernel(T *__restrict__ res, const T *__restrict__ M, int64_t m, int64_t n, int64_t ld,
                   const T *__restrict__ v, CA alpha, CB beta) {
  constexpr int VR = 16 / (int)sizeof(T);
  typedef T VV __attribute__((ext_vector_type(VR)));
  constexpr int LPR = RB / VR, NCL = kGemvRowsBlock / LPR, U = 8;
  static_assert(RB % VR == 0 && kGemvRowsBlock % LPR == 0 && RB <= kGemvRowsBlock, "bad row band");
  __shared__ double sred[NCL][RB];
  const int tid = threadIdx.x, seg = tid % LPR, cl = tid / LPR;
  const int64_t row = (int64_t)blockIdx.x * RB + (int64_t)seg * VR;
  double acc[VR];
#pragma unroll
  for (int e = 0; e < VR; ++e) acc[e] = 0.0;
  if (row < m) {                                  // m % VR == 0 (gemv_n): a vector is wholly inside or wholly outside
    const T *base = M + row;
    int64_t j = cl;
    for (; j + (int64_t)(U - 1) * NCL < n; j += (int64_t)U * NCL) {
      VV a[U];
      T x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const VV *q = reinterpret_cast<const VV *>(base + (j + (int64_t)u * NCL) * ld);
        a[u] = NT ? __builtin_nontemporal_load(q) : *q;
        x[u] = v[j + (int64_t)u * NCL];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < VR; ++e) acc[e] = fma((double)a[u][e], (double)x[u], acc[e]);
    }
    for (; j < n; j += NCL) {
      const VV a = *reinterpret_cast<const VV *>(base + j * ld);
      const double x = (double)v[j];
#pragma unroll
      for (int e = 0; e < VR; ++e) acc[e] = fma((double)a[e], x, acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < VR; ++e) sred[cl][seg * VR + e] = acc[e];
  __syncthreads();
  // RB * Q threads: thread (q, r) adds column lanes q, q + Q, ... of row r; the Q sub-sums are added in a fixed order
  constexpr int Q = RB * 8 <= kGemvRowsBlock ? 8 : kGemvRowsBlock / RB;
  static_assert(Q >= 1 && NCL % Q == 0 && RB * Q <= kGemvRowsBlock, "bad finish shape");
  double part = 0.0;
  const int r = tid % RB, q = tid / RB;
  if (tid < RB * Q) {
#pragma unroll 8
    for (int c = q; c < NCL; c += Q) part += sred[c][r];
  }
  __syncthreads();
  if (tid < RB * Q) sred[q][r] = part;
  __syncthreads();
  if (tid < RB) {
    const int64_t i = (int64_t)blockIdx.x * RB + tid;
    if (i < m) {
      double s = 0.0;
#pragma unroll
      for (int qq = 0; qq < Q; ++qq) s += sred[qq][tid];
      res[i] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)s, beta, BETA0 ? T(0) : res[i]);
    }
  }
}

// the band height gemv_n uses for an m x n operand (0: keep the column-chunk schedule)
template <typename T>
int gemv_rows_band(const mxlo_ctx *ctx, const T *M, int64_t m, int64_t n, int64_t ld) {
  constexpr int VR = 16 / (int)sizeof(T);
  const bool vec = (((uintptr_t)M & 15u) == 0) && ld % VR == 0 && m % VR == 0;
  // tune gemv_n_rows: 0 = off, 1 = auto, 8*VR / 16*VR / 32*VR = that band height (sweeps)
  int rb = 0;
  const int want = ctx->tune.gemv_n_rows;
  if (want > 1) {
    if (want == 8 * VR || want == 16 * VR || want == 32 * VR) rb = want;
  } else if (want == 1 && n >= 1024) {
    // the tallest band that still gives every CU a workgroup: a band reads RB * sizeof(T) contiguous bytes per column
    // (128 B at 8 * VR), and the longer the pieces the better HBM streams them (tools/bench_gemv_n.py)
    if (m >= (int64_t)32 * VR * ctx->num_cu) rb = 32 * VR;
    else if (m >= (int64_t)16 * VR * ctx->num_cu) rb = 16 * VR;
    else if (m >= (int64_t)8 * VR * ctx->num_cu) rb = 8 * VR;
    // measured (profiles/r05_bench_gemv_n.txt): 512-byte pieces win everywhere (0.84 - 0.88 of peak); 256-byte pieces
    // win (0.78 - 0.84) except against the column-chunk schedule on very wide matrices, whose finish launch is amortised
    // there; 128-byte pieces top out at 0.68 - 0.73: better than the two launches up to n = 16384 columns only
    if (rb == 16 * VR && n >= 16384) rb = 0;
    if (rb == 8 * VR && n > 16384) rb = 0;
  }
  return (rb != 0 && vec && m >= rb) ? rb : 0;
}

template <typename T>
int32_t gemv_n(mxlo_ctx *ctx, T *res, const T *M, int64_t m, int64_t n, int64_t ld, const T *v,
               double alpha, double beta, int32_t flags) {
  {
    constexpr int VR = 16 / (int)sizeof(T);
    const int rb = gemv_rows_band<T>(ctx, M, m, n, ld);
    if (rb != 0) {
      const bool nt = (int64_t)sizeof(T) * m * n >= ctx->tune.nt_min_bytes;   // see gemv_t: cache-sized matrices keep default loads
      return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
#define ROWS(RB_)                                                                                                     \
  if (nt) hipLaunchKernelGGL((gemv_n_rows_kernel<T, CA, CB, B0, RB_, true>), dim3((unsigned)((m + (RB_) - 1) / (RB_))),   \
                             dim3(kGemvRowsBlock), 0, ctx->stream, res, M, m, n, ld, v, (CA)alpha, (CB)beta);         \
  else hipLaunchKernelGGL((gemv_n_rows_kernel<T, CA, CB, B0, RB_, false>), dim3((unsigned)((m + (RB_) - 1) / (RB_))),     \
                          dim3(kGemvRowsBlock), 0, ctx->stream, res, M, m, n, ld, v, (CA)alpha, (CB)beta)
        if (rb == 32 * VR) { ROWS(32 * VR); }
        else if (rb == 16 * VR) { ROWS(16 * VR); }
        else { ROWS(8 * VR); }
#undef ROWS
        MXLO_LAUNCH_CHECK();
        return MXLO_OK;
      });
    }
  }
  const int64_t cap = (int64_t)kMaxRedCols * kMaxRedBlocks;  // doubles in ctx->partials
  MXLO_REQUIRE(m <= cap, MXLO_ESHAPE, "gemv: m = %lld exceeds the partial workspace", (long long)m);
  // pairs of rows per lane need 16-byte (8-byte for f32) aligned column starts
  const bool pair = m >= 2 && (((uintptr_t)M % (2 * sizeof(T))) == 0) && (ld % 2 == 0);
  const int64_t rows_per_block = (int64_t)kBlock * (pair ? 2 : 1);
  const int64_t row_blocks = (m + rows_per_block - 1) / rows_per_block;
  int64_t nchunks = (n + 63) / 64;              // >= 64 columns per chunk
  const int64_t want = (int64_t)ctx->num_cu * 8 / (row_blocks > 0 ? row_blocks : 1) + 1;
  if (nchunks > want) nchunks = want;
  if (nchunks > cap / (m > 0 ? m : 1)) nchunks = cap / (m > 0 ? m : 1);
  if (nchunks < 1) nchunks = 1;
  if (nchunks > 65535) nchunks = 65535;
  const int64_t cpc = (n + nchunks - 1) / nchunks;
  nchunks = n > 0 ? (n + cpc - 1) / cpc : 1;
  dim3 grid((unsigned)row_blocks, (unsigned)nchunks);
  if (pair)
    hipLaunchKernelGGL((gemv_n_partial_kernel<T, true>), grid, dim3(kBlock), 0, ctx->stream, ctx->partials, M, m,
                       n, ld, v, cpc > 0 ? cpc : 1);
  else
    hipLaunchKernelGGL((gemv_n_partial_kernel<T, false>), grid, dim3(kBlock), 0, ctx->stream, ctx->partials, M, m,
                       n, ld, v, cpc > 0 ? cpc : 1);
  MXLO_LAUNCH_CHECK();
  const unsigned fin_blocks = (unsigned)((m + 31) / 32);
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    hipLaunchKernelGGL((gemv_n_finish_kernel<T, CA, CB, B0>), dim3(fin_blocks), dim3(kBlock), 0, ctx->stream, res,
                       ctx->partials, m, (int)nchunks, (CA)alpha, (CB)beta, (T *)nullptr);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

template <typename T>
int32_t gemv_t(mxlo_ctx *ctx, T *res, const T *M, int64_t m, int64_t n, int64_t ld, const T *v,
               double alpha, double beta, int32_t flags) {
  if (n <= 0) return MXLO_OK;
  int64_t blocks = (n + 3) / 4;
  const int64_t cap = (int64_t)ctx->num_cu * 16;
  if (blocks > cap) blocks = cap;
  constexpr int VR = 16 / (int)sizeof(T);     // 16-byte loads down the column need 16-byte aligned column starts
  const bool pair = m >= VR && (((uintptr_t)M & 15u) == 0) && (ld % VR == 0) && (((uintptr_t)v & 15u) == 0);
  // a matrix that fits the 256 MiB Infinity Cache is re-read from it by the next apply of the same operator (Krylov loops):
  // the nontemporal hint (worth 6.3 -> 7.1 TB/s on an HBM-sized read) is kept for footprints beyond nt_min_bytes only
  const bool nt = (int64_t)sizeof(T) * m * n >= ctx->tune.nt_min_bytes;
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    if (pair && !nt)
      hipLaunchKernelGGL((gemv_t_kernel<T, CA, CB, B0, true, false>), dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                         res, M, m, n, ld, v, (CA)alpha, (CB)beta);
    else if (pair)
      hipLaunchKernelGGL((gemv_t_kernel<T, CA, CB, B0, true>), dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                         res, M, m, n, ld, v, (CA)alpha, (CB)beta);
    else
      hipLaunchKernelGGL((gemv_t_kernel<T, CA, CB, B0, false>), dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                         res, M, m, n, ld, v, (CA)alpha, (CB)beta);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

template <typename T>
int32_t gemv_any(mxlo_ctx *ctx, T *res, const T *M, int64_t m, int64_t n, int64_t ld, const T *v,
                 double alpha, double beta, int32_t mode, int32_t flags) {
  const int64_t nres = mode == MXLO_OP_N ? m : n;
  const int64_t nin = mode == MXLO_OP_N ? n : m;
  if (nres == 0) return MXLO_OK;
  if (nin == 0) {  // empty sum: res = beta*res (or 0)
    if (beta == 0) {
      MXLO_HIP(hipMemsetAsync(res, 0, sizeof(T) * nres, ctx->stream));
      return MXLO_OK;
    }
    return mxlo_scale(ctx, sizeof(T) == 8 ? MXLO_F64 : MXLO_F32, res, nres, beta, flags);
  }
  if (mode == MXLO_OP_N) return gemv_n<T>(ctx, res, M, m, n, ld, v, alpha, beta, flags);
  return gemv_t<T>(ctx, res, M, m, n, ld, v, alpha, beta, flags);
}

// ---- block GEMV: res (.. x k) = alpha * op(M) * V (.. x k) + beta * res, M read ONCE for KB columns of V ---------
// `mul!(res::Matrix, LinearOperator(M), V::Matrix, α, β)` (src/operations.jl:34-36 with the closure of
// src/constructors.jl:19-29, a GEMM in the reference) for the tall-skinny blocks of block Krylov methods: k is a
// handful, so the product is HBM-bound on M — one pass over M for the whole block instead of one per column.
// N mode: a lane owns one row (a pair with 16-byte loads) and KB accumulators per row; V[j, c] is uniform over the
// wave (scalar loads). Column chunks per blockIdx.y as in gemv_n_partial_kernel; part[(chunk*KB + c)*m + row].
template <typename T, int KB, bool PAIR>
__global__ void __launch_bounds__(kBlock)
gemvb_n_partial_kernel(double *__restrict__ part, const T *__restrict__ M, int64_t m, int64_t n, int64_t ld,
                       const T *__restrict__ V, int64_t ldv, int64_t cols_per_chunk) {
  typedef T V2 __attribute__((ext_vector_type(2)));
  constexpr int R = PAIR ? 2 : 1;
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * R;
  const int64_t j0 = (int64_t)blockIdx.y * cols_per_chunk;
  int64_t j1 = j0 + cols_per_chunk;
  if (j1 > n) j1 = n;
  if (i >= m) return;
  double acc0[KB], acc1[KB];
#pragma unroll
  for (int c = 0; c < KB; ++c) acc0[c] = acc1[c] = 0.0;
  const bool two = PAIR && i + 1 < m;
  int64_t j = j0;
  if (two) {
    constexpr int U = 4;
    for (; j + U <= j1; j += U) {
      V2 a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(M + i + (j + u) * ld));
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int c = 0; c < KB; ++c) {
          const double vj = (double)V[(j + u) + (int64_t)c * ldv];
          acc0[c] = fma((double)a[u][0], vj, acc0[c]);
          acc1[c] = fma((double)a[u][1], vj, acc1[c]);
        }
      }
    }
    for (; j < j1; ++j) {
      const V2 a = *reinterpret_cast<const V2 *>(M + i + j * ld);
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const double vj = (double)V[j + (int64_t)c * ldv];
        acc0[c] = fma((double)a[0], vj, acc0[c]);
        acc1[c] = fma((double)a[1], vj, acc1[c]);
      }
    }
  } else {
    for (; j < j1; ++j) {
      const double a = (double)M[i + j * ld];
#pragma unroll
      for (int c = 0; c < KB; ++c) acc0[c] = fma(a, (double)V[j + (int64_t)c * ldv], acc0[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < KB; ++c) {
    double *p = part + ((int64_t)blockIdx.y * KB + c) * m + i;
    p[0] = acc0[c];
    if (two) p[1] = acc1[c];
  }
}

// blockIdx.y = column of the block; 32 rows per workgroup, 8 lanes per row, fixed-order sums (as gemv_n_finish_kernel)
template <typename T, typename CA, typename CB, bool BETA0>
__global__ void __launch_bounds__(kBlock)
gemvb_n_finish_kernel(T *__restrict__ res, int64_t ldr, const double *__restrict__ part, int64_t m, int nchunks, int kb,
                      CA alpha, CB beta) {
  const int r = threadIdx.x & 31, sub = threadIdx.x >> 5, c = (int)blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 32 + r;
  __shared__ double sred[8][32];
  double a = 0.0;
  if (i < m)
    for (int ch = sub; ch < nchunks; ch += 8) a += part[((int64_t)ch * kb + c) * m + i];
  sred[sub][r] = a;
  __syncthreads();
  if (sub != 0 || i >= m) return;
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) acc += sred[q][r];
  T *o = res + i + (int64_t)c * ldr;
  *o = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)acc, beta, BETA0 ? T(0) : *o);
}

// N mode, ROW BANDS for a block of KB vectors (round 6): gemv_n_rows_kernel's schedule — a 512-thread workgroup owns RB
// rows across ALL columns, LPR lanes cover the band's piece of one column with 16-byte loads, so the sum over the columns
// never leaves the workgroup: no partial workspace (the column-chunk form above writes and re-reads nchunks * KB * m
// doubles, 69 MB next to a 2.1 GB matrix at n = 16384, k = 8, and adds ONE write stream to the read streams) and no
// dependent finish launch — with KB accumulators per row and lane. A wave's lanes span 64 / LPR columns of M, so V[j, c]
// is not wave-uniform: the block V is staged in LDS once per WORKGROUP, CH columns x KB at a time in its own layout
// (double-buffered: the next chunk travels global -> registers while the current one is consumed; one barrier per chunk),
// and every lane reads its column's KB values from there (ds_read_b64, the lanes of one column share the address). The
// loads of M for the next step are issued BEFORE the current step is consumed (two register sets): a barrier never finds
// the memory pipe empty. Column assignment, order of the fma's per lane, tail and the fixed-order sum over the column
// lanes through LDS are gemv_n_rows_kernel's, per vector: column c of the block has the BITS of the single apply.
template <typename T, typename CA, typename CB, bool BETA0, int KB, int RB, bool NT>
__global__ void __launch_bounds__(kGemvRowsBlock)
gemvb_n_rows_kernel(T *__restrict__ res, int64_t ldr, const T *__restrict__ M, int64_t m, int64_t n, int64_t ld,
                    const T *__restrict__ V, int64_t ldv, CA alpha, CB beta) {
  constexpr int VR = 16 / (int)sizeof(T);
  typedef T VV __attribute__((ext_vector_type(VR)));
  // U loads of M per lane and register set (two sets): 8 + 8 in flight up to 4 vectors, 4 + 4 with 8 (the accumulators take the room)
  constexpr int LPR = RB / VR, NCL = kGemvRowsBlock / LPR, U = (KB >= 8 || RB == 8 * VR) ? 4 : 8, STEP = NCL * U;
  constexpr int ITS = STEP >= 512 ? 1 : 512 / STEP, CH = STEP * ITS;      // 512 columns of V per staged chunk
  static_assert(RB % VR == 0 && kGemvRowsBlock % LPR == 0 && RB <= kGemvRowsBlock, "bad row band");
  constexpr int SL = KB * CH / kGemvRowsBlock;                            // staging loads per thread and chunk
  static_assert(KB * CH % kGemvRowsBlock == 0, "bad staging shape");
  constexpr int kRedDoubles = NCL * RB, kStageDoubles = (int)((size_t)2 * KB * CH * sizeof(T) / 8);
  __shared__ double lds[kRedDoubles > kStageDoubles ? kRedDoubles : kStageDoubles];
  T *vlds = reinterpret_cast<T *>(lds);                                   // [2][KB][CH]
  const int tid = threadIdx.x, seg = tid % LPR, cl = tid / LPR;
  const int64_t row = (int64_t)blockIdx.x * RB + (int64_t)seg * VR;
  const bool live = row < m;                       // m % VR == 0: a vector is wholly inside or wholly outside
  const T *base = M + (live ? row : 0);
  double acc[KB][VR];
#pragma unroll
  for (int c = 0; c < KB; ++c)
#pragma unroll
    for (int e = 0; e < VR; ++e) acc[c][e] = 0.0;
  const int64_t nmain = n / STEP;                  // whole steps of U columns per column lane
  const int64_t nfull = nmain / ITS;               // whole chunks of ITS steps
  T sreg[SL];
  auto stage_load = [&](int64_t ch) {              // global -> registers: V[ch * CH + j, c], zero past the end (masked, never skipped)
#pragma unroll
    for (int k = 0; k < SL; ++k) {
      const int idx = tid + k * kGemvRowsBlock, c = idx / CH, jl = idx % CH;
      const int64_t j = ch * CH + jl;
      const T x = V[(j < n ? j : n - 1) + (int64_t)c * ldv];     // clamped, unconditional load + select: no branch
      sreg[k] = j < n ? x : T(0);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int k = 0; k < SL; ++k) vlds[(size_t)buf * KB * CH + tid + k * kGemvRowsBlock] = sreg[k];
  };
  auto load_step = [&](VV (&a)[U], int64_t it) {
    const int64_t j = it * STEP + cl;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const VV *q = reinterpret_cast<const VV *>(base + (j + (int64_t)u * NCL) * ld);
      a[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
  };
  auto consume = [&](const VV (&a)[U], int64_t it) {
    const T *x = vlds + (size_t)((it / ITS) & 1) * KB * CH + (it % ITS) * STEP + cl;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const double xv = (double)x[c * CH + u * NCL];
#pragma unroll
        for (int e = 0; e < VR; ++e) acc[c][e] = fma((double)a[u][e], xv, acc[c][e]);
      }
  };
  stage_load(0);
  stage_store(0);
  __syncthreads();
  // (lanes past the last row — only in the last band — read row 0's pieces instead: every lane of the workgroup walks
  //  the same schedule and meets the same barriers; their sums are never stored)
  // The steady state is STRAIGHT-LINE code per chunk — every load of it is issued unconditionally (past the end: a clamped,
  // valid step whose data is not used; staging past n: masked to zero) — so that the counted s_waitcnt before each consume
  // is exact: a prefetch under a condition makes the compiler wait for vmcnt(0), i.e. serialises load and use (measured:
  // 735 us instead of 330 at n = 16384, k = 2).
  if (nmain > 0) {
    static_assert(ITS % 2 == 0, "two register sets");
    VV a0[U], a1[U];
    load_step(a0, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int64_t ch = 0; ch < nfull; ++ch) {
      stage_load(ch + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sidx = 0; sidx < ITS; sidx += 2) {
        const int64_t it = ch * ITS + sidx;
        // (scheduling fences: left alone, the compiler sinks each prefetch into the consume that follows it to save
        //  registers and waits for it with vmcnt(0) — the pipelining is the point of the two register sets)
        load_step(a1, it + 1);                                     // inside the chunk: always a whole step
        __builtin_amdgcn_sched_barrier(0);
        consume(a0, it);
        __builtin_amdgcn_sched_barrier(0);
        load_step(a0, it + 2 < nmain ? it + 2 : nmain - 1);        // the step after the last: a valid one, unused
        __builtin_amdgcn_sched_barrier(0);
        consume(a1, it + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      stage_store((int)((ch + 1) & 1));
      __syncthreads();
    }
    int64_t it = nfull * ITS;                                      // the steps of the last, partial chunk (already staged)
    if (it < nmain) {
      consume(a0, it);                                             // requested by the loop above (or before it)
      for (++it; it < nmain; ++it) {
        load_step(a0, it);
        consume(a0, it);
      }
    }
  }
  if (live) {                                      // columns past the last whole step, one per column lane and trip
    for (int64_t j = nmain * STEP + cl; j < n; j += NCL) {
      const VV a = *reinterpret_cast<const VV *>(base + j * ld);
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const double xv = (double)V[j + (int64_t)c * ldv];
#pragma unroll
        for (int e = 0; e < VR; ++e) acc[c][e] = fma((double)a[e], xv, acc[c][e]);
      }
    }
  }
  // per vector of the block: the fixed-order sum over the NCL column lanes of gemv_n_rows_kernel
  constexpr int Q = RB * 8 <= kGemvRowsBlock ? 8 : kGemvRowsBlock / RB;
  static_assert(Q >= 1 && NCL % Q == 0 && RB * Q <= kGemvRowsBlock, "bad finish shape");
  double(*sred)[RB] = reinterpret_cast<double(*)[RB]>(lds);
  const int r = tid % RB, q = tid / RB;
#pragma unroll
  for (int c = 0; c < KB; ++c) {
    __syncthreads();                               // the staging buffers / the previous vector's sums are dead
#pragma unroll
    for (int e = 0; e < VR; ++e) sred[cl][seg * VR + e] = acc[c][e];
    __syncthreads();
    double part = 0.0;
    if (tid < RB * Q) {
#pragma unroll 8
      for (int cc = q; cc < NCL; cc += Q) part += sred[cc][r];
    }
    __syncthreads();
    if (tid < RB * Q) sred[q][r] = part;
    __syncthreads();
    if (tid < RB) {
      const int64_t i = (int64_t)blockIdx.x * RB + tid;
      if (i < m) {
        double s = 0.0;
#pragma unroll
        for (int qq = 0; qq < Q; ++qq) s += sred[qq][tid];
        T *o = res + i + (int64_t)c * ldr;
        *o = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)s, beta, BETA0 ? T(0) : *o);
      }
    }
  }
}

// T mode: a wave owns JB = 4 consecutive columns of M and forms their dots with all KB columns of U at once: the four
// columns of M are streamed once, and each 16-byte piece of U (L2 / L1 resident) is loaded once per FOUR columns of M
// instead of once per column (one column per wave ran at 0.29 of HBM peak for KB = 8: 9 vector loads per M load).
//   res[j + c*ldr] = alpha * dot(M[:, j], U[:, c]) + beta * res[j + c*ldr]
template <typename T, typename CA, typename CB, bool BETA0, int KB, bool PAIR>
__global__ void __launch_bounds__(kBlock)
gemvb_t_kernel(T *__restrict__ res, int64_t ldr, const T *__restrict__ M, int64_t m, int64_t n, int64_t ld,
               const T *__restrict__ Um, int64_t ldu, CA alpha, CB beta) {
  constexpr int VR = 16 / (int)sizeof(T);
  constexpr int JB = 4;
  typedef T VV __attribute__((ext_vector_type(VR)));
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / kWave);
  const int64_t ngroups = (n + JB - 1) / JB;
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    const int64_t j0 = g * JB;
    const int jn = (int)(n - j0 < JB ? n - j0 : JB);           // columns of this group inside the matrix
    const T *colp[JB];
#pragma unroll
    for (int q = 0; q < JB; ++q) colp[q] = M + (j0 + (q < jn ? q : 0)) * ld;   // past the edge: alias column j0, result dropped
    double acc[JB][KB];
#pragma unroll
    for (int q = 0; q < JB; ++q)
#pragma unroll
      for (int c = 0; c < KB; ++c) acc[q][c] = 0.0;
    int64_t i = 0;
    if constexpr (PAIR) {
      const int64_t mp = m / VR;
      int64_t p = lane;
      for (; p + 64 < mp; p += 128) {                     // 8 loads of M in flight per lane
        VV a0[JB], a1[JB];
#pragma unroll
        for (int q = 0; q < JB; ++q) {
          a0[q] = __builtin_nontemporal_load(reinterpret_cast<const VV *>(colp[q]) + p);
          a1[q] = __builtin_nontemporal_load(reinterpret_cast<const VV *>(colp[q]) + p + 64);
        }
#pragma unroll
        for (int c = 0; c < KB; ++c) {
          const VV x0 = *reinterpret_cast<const VV *>(Um + (int64_t)c * ldu + p * VR);
          const VV x1 = *reinterpret_cast<const VV *>(Um + (int64_t)c * ldu + (p + 64) * VR);
#pragma unroll
          for (int q = 0; q < JB; ++q)
#pragma unroll
            for (int e = 0; e < VR; ++e) {
              acc[q][c] = fma((double)a0[q][e], (double)x0[e], acc[q][c]);
              acc[q][c] = fma((double)a1[q][e], (double)x1[e], acc[q][c]);
            }
        }
      }
      for (; p < mp; p += 64) {
        VV a[JB];
#pragma unroll
        for (int q = 0; q < JB; ++q) a[q] = __builtin_nontemporal_load(reinterpret_cast<const VV *>(colp[q]) + p);
#pragma unroll
        for (int c = 0; c < KB; ++c) {
          const VV x = *reinterpret_cast<const VV *>(Um + (int64_t)c * ldu + p * VR);
#pragma unroll
          for (int q = 0; q < JB; ++q)
#pragma unroll
            for (int e = 0; e < VR; ++e) acc[q][c] = fma((double)a[q][e], (double)x[e], acc[q][c]);
        }
      }
      i = mp * VR;
    }
    for (int64_t r = i + lane; r < m; r += 64) {
      double a[JB];
#pragma unroll
      for (int q = 0; q < JB; ++q) a[q] = (double)colp[q][r];
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const double x = (double)Um[(int64_t)c * ldu + r];
#pragma unroll
        for (int q = 0; q < JB; ++q) acc[q][c] = fma(a[q], x, acc[q][c]);
      }
    }
#pragma unroll
    for (int q = 0; q < JB; ++q) {
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        double v = acc[q][c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && q < jn) {
          T *o = res + (j0 + q) + (int64_t)c * ldr;
          *o = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)v, beta, BETA0 ? T(0) : *o);
        }
      }
    }
  }
}

// T mode, LDS-staged (round 5): gemvb_t_kernel above fetches every 16-byte piece of U from L1/L2 once per FOUR columns of
// M — 2 vector-memory instructions on U per load of M at KB = 8, and the vector memory pipe, not HBM, sets the pace
// (0.52 of peak at n = 16384, k = 8). Here a workgroup of 512 threads owns 32 columns of M (4 per wave) and walks the rows in chunks of
// 128 vectors: the chunk of U (KB x 128 vectors, 16 KiB at KB = 8) is staged in LDS once per WORKGROUP (double-buffered:
// the next chunk travels global -> registers while the current one is consumed), every wave reads it from there
// (ds_read_b128: 4x the L1 rate), and the only vector-memory traffic of the inner loop is M itself, 16 loads in flight per
// lane. U is re-read from L2 once per 32 columns of M (n/32 x |U| instead of n/4 x |U| through L1).
// Preconditions (checked by the caller): M, U 16-byte aligned, ld and ldu multiples of the vector width.
// kGemvbLdsBlock threads: 8 waves x 4 columns of M = 32 columns per workgroup (256 threads: 16 columns, for matrices with
// fewer than one 32-column group per CU)
template <typename T, typename CA, typename CB, bool BETA0, int KB, int kGemvbLdsBlock>
__global__ void __launch_bounds__(kGemvbLdsBlock)
gemvb_t_lds_kernel(T *__restrict__ res, int64_t ldr, const T *__restrict__ M, int64_t m, int64_t n, int64_t ld,
                   const T *__restrict__ Um, int64_t ldu, CA alpha, CB beta) {
  constexpr int VR = 16 / (int)sizeof(T);
  constexpr int JB = 4, RCV = 128, NW = kGemvbLdsBlock / kWave;
  typedef T VV __attribute__((ext_vector_type(VR)));
  __shared__ VV ulds[2][KB][RCV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t j0 = ((int64_t)blockIdx.x * NW + wave) * JB;
  const int64_t mp = m / VR;                                   // whole vectors per column
  const int64_t nchunks = (mp + RCV - 1) / RCV;
  const T *colp[JB];
#pragma unroll
  for (int q = 0; q < JB; ++q) colp[q] = M + (j0 + q < n ? j0 + q : (j0 < n ? j0 : 0)) * ld;   // past the edge: a valid column, result dropped
  double acc[JB][KB];
#pragma unroll
  for (int q = 0; q < JB; ++q)
#pragma unroll
    for (int c = 0; c < KB; ++c) acc[q][c] = 0.0;
  constexpr int UL = KB * RCV / kGemvbLdsBlock;                        // staging loads per thread and chunk
  VV ureg[UL];
  auto stage_load = [&](int64_t ch) {                          // global -> registers (vectors past the end: zero)
#pragma unroll
    for (int k = 0; k < UL; ++k) {
      const int idx = tid + k * kGemvbLdsBlock, c = idx / RCV, pv = idx % RCV;
      const int64_t p = ch * RCV + pv;
      VV z;
#pragma unroll
      for (int e = 0; e < VR; ++e) z[e] = T(0);
      ureg[k] = p < mp ? *reinterpret_cast<const VV *>(Um + (int64_t)c * ldu + p * VR) : z;
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int k = 0; k < UL; ++k) {
      const int idx = tid + k * kGemvbLdsBlock;
      ulds[buf][idx / RCV][idx % RCV] = ureg[k];
    }
  };
  if (nchunks > 0) {
    stage_load(0);
    stage_store(0);
  }
  __syncthreads();
  for (int64_t ch = 0; ch < nchunks; ++ch) {
    const int buf = (int)(ch & 1);
    if (ch + 1 < nchunks) stage_load(ch + 1);                  // in flight while this chunk is consumed
    VV a[2][JB];
    const int64_t p0 = ch * RCV + lane, p1 = p0 + 64;
    const int64_t q0 = p0 < mp ? p0 : mp - 1, q1 = p1 < mp ? p1 : mp - 1;   // clamped: the value is masked below
#pragma unroll
    for (int q = 0; q < JB; ++q) {
      a[0][q] = __builtin_nontemporal_load(reinterpret_cast<const VV *>(colp[q]) + q0);
      a[1][q] = __builtin_nontemporal_load(reinterpret_cast<const VV *>(colp[q]) + q1);
    }
    if (p1 >= mp) {                                            // only the last chunk: rows past the end contribute nothing
#pragma unroll
      for (int q = 0; q < JB; ++q)
#pragma unroll
        for (int e = 0; e < VR; ++e) {
          if (p0 >= mp) a[0][q][e] = T(0);
          a[1][q][e] = T(0);
        }
    }
#pragma unroll
    for (int c = 0; c < KB; ++c) {
      const VV x0 = ulds[buf][c][lane], x1 = ulds[buf][c][lane + 64];
#pragma unroll
      for (int q = 0; q < JB; ++q)
#pragma unroll
        for (int e = 0; e < VR; ++e) {
          acc[q][c] = fma((double)a[0][q][e], (double)x0[e], acc[q][c]);
          acc[q][c] = fma((double)a[1][q][e], (double)x1[e], acc[q][c]);
        }
    }
    if (ch + 1 < nchunks) stage_store(buf ^ 1);
    __syncthreads();
  }
  for (int64_t r = mp * VR + lane; r < m; r += 64) {           // rows past the last whole vector
#pragma unroll
    for (int c = 0; c < KB; ++c) {
      const double x = (double)Um[(int64_t)c * ldu + r];
#pragma unroll
      for (int q = 0; q < JB; ++q) acc[q][c] = fma((double)colp[q][r], x, acc[q][c]);
    }
  }
#pragma unroll
  for (int q = 0; q < JB; ++q) {
#pragma unroll
    for (int c = 0; c < KB; ++c) {
      const double v = wave_allsum(acc[q][c]);
      if (lane == 0 && j0 + q < n) {
        T *o = res + (j0 + q) + (int64_t)c * ldr;
        *o = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)v, beta, BETA0 ? T(0) : *o);
      }
    }
  }
}

inline int32_t ensure_scratch(mxlo_ctx *ctx, size_t need, const char *what) {
  if (ctx->scratch_bytes < need) {            // stream-ordered users only: drain before the buffer is replaced
    if (ctx->scratch) {
      MXLO_HIP(hipStreamSynchronize(ctx->stream));
      MXLO_HIP(hipFree(ctx->scratch));
    }
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    hipError_t e = hipMalloc(&ctx->scratch, need);
    MXLO_REQUIRE(e == hipSuccess, MXLO_ENOMEM, "%s scratch: %s", what, hipGetErrorString(e));
    ctx->scratch_bytes = need;
    ++ctx->scratch_generation;     // graphs that recorded the old workspace pointer are stale now
  }
  if (ctx->capturing) ctx->scratch_used_in_capture = true;
  return MXLO_OK;
}

// one chunk of KB <= 8 columns
template <typename T, int KB>
int32_t gemv_block_chunk(mxlo_ctx *ctx, T *res, int64_t ldr, const T *M, int64_t m, int64_t n, int64_t ld, const T *V,
                         int64_t ldv, double alpha, double beta, int32_t mode, int32_t flags) {
  if (mode == MXLO_OP_N) {
    // Row bands with the block staged in LDS per WORKGROUP (round 6; the round-5 attempt staged it per wave and step and was
    // slower everywhere): taken where the single apply takes its 512- / 256-byte bands, so that every column of the block
    // has the bits of gemv_n on that column.
    {
      constexpr int VR = 16 / (int)sizeof(T);
      // the tallest band that still gives every CU a workgroup (gemv_rows_band's rule without its exceptions for very wide
      // matrices: those weigh a finish launch that the block form's column-chunk schedule pays KB-fold)
      const bool vec = (((uintptr_t)M & 15u) == 0) && ld % VR == 0 && m % VR == 0;
      int rb = 0;
      if (ctx->tune.gemvb_n_rows && vec && n >= 1024) {
        if (m >= (int64_t)32 * VR * ctx->num_cu) rb = 32 * VR;
        else if (m >= (int64_t)16 * VR * ctx->num_cu) rb = 16 * VR;
        else if (m >= (int64_t)8 * VR * ctx->num_cu) rb = 8 * VR;
      }
      if (rb != 0) {
        const bool nt = (int64_t)sizeof(T) * m * n >= ctx->tune.nt_min_bytes;
        return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
#define BROWS(RB_, NT_)                                                                                                  \
  hipLaunchKernelGGL((gemvb_n_rows_kernel<T, CA, CB, B0, KB, RB_, NT_>), dim3((unsigned)((m + (RB_) - 1) / (RB_))),      \
                     dim3(kGemvRowsBlock), 0, ctx->stream, res, ldr, M, m, n, ld, V, ldv, (CA)alpha, (CB)beta)
          if (rb == 32 * VR) { if (nt) BROWS(32 * VR, true); else BROWS(32 * VR, false); }
          else if (rb == 16 * VR) { if (nt) BROWS(16 * VR, true); else BROWS(16 * VR, false); }
          else { if (nt) BROWS(8 * VR, true); else BROWS(8 * VR, false); }
#undef BROWS
          MXLO_LAUNCH_CHECK();
          return MXLO_OK;
        });
      }
    }
    const bool pair = m >= 2 && (((uintptr_t)M % (2 * sizeof(T))) == 0) && (ld % 2 == 0);
    const int64_t rows_per_block = (int64_t)kBlock * (pair ? 2 : 1);
    const int64_t row_blocks = (m + rows_per_block - 1) / rows_per_block;
    int64_t nchunks = (n + 63) / 64;
    const int64_t want = (int64_t)ctx->num_cu * 8 / (row_blocks > 0 ? row_blocks : 1) + 1;
    if (nchunks > want) nchunks = want;
    if (nchunks > 65535) nchunks = 65535;
    if (nchunks < 1) nchunks = 1;
    const int64_t cpc = (n + nchunks - 1) / nchunks;
    nchunks = (n + cpc - 1) / cpc;
    MXLO_TRY(ensure_scratch(ctx, sizeof(double) * (size_t)nchunks * KB * (size_t)m, "block GEMV"));
    double *part = (double *)ctx->scratch;
    dim3 grid((unsigned)row_blocks, (unsigned)nchunks);
    if (pair)
      hipLaunchKernelGGL((gemvb_n_partial_kernel<T, KB, true>), grid, dim3(kBlock), 0, ctx->stream, part, M, m, n, ld, V,
                         ldv, cpc);
    else
      hipLaunchKernelGGL((gemvb_n_partial_kernel<T, KB, false>), grid, dim3(kBlock), 0, ctx->stream, part, M, m, n, ld,
                         V, ldv, cpc);
    MXLO_LAUNCH_CHECK();
    return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
      hipLaunchKernelGGL((gemvb_n_finish_kernel<T, CA, CB, B0>), dim3((unsigned)((m + 31) / 32), (unsigned)KB),
                         dim3(kBlock), 0, ctx->stream, res, ldr, part, m, (int)nchunks, KB, (CA)alpha, (CB)beta);
      MXLO_LAUNCH_CHECK();
      return MXLO_OK;
    });
  }
  int64_t blocks = ((n + 3) / 4 + 3) / 4;          // 4 columns per wave, 4 waves per workgroup
  const int64_t cap = (int64_t)ctx->num_cu * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  constexpr int VR = 16 / (int)sizeof(T);
  const bool pair = m >= VR && (((uintptr_t)M & 15u) == 0) && (ld % VR == 0) && (((uintptr_t)V & 15u) == 0) &&
                    (ldv % VR == 0);
  // >= 4 columns of U on a matrix tall enough for a few row chunks: the LDS-staged form (32 columns of M per workgroup)
  const bool lds_form = pair && KB >= 4 && m >= 4 * 128 * VR && ctx->tune.gemvb_t_lds;
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    if constexpr (KB >= 4) {
      if (lds_form) {
        if ((n + 31) / 32 >= ctx->num_cu)
          hipLaunchKernelGGL((gemvb_t_lds_kernel<T, CA, CB, B0, KB, 512>), dim3((unsigned)((n + 31) / 32)), dim3(512), 0, ctx->stream,
                             res, ldr, M, m, n, ld, V, ldv, (CA)alpha, (CB)beta);
        else
          hipLaunchKernelGGL((gemvb_t_lds_kernel<T, CA, CB, B0, KB, 256>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, ctx->stream,
                             res, ldr, M, m, n, ld, V, ldv, (CA)alpha, (CB)beta);
        MXLO_LAUNCH_CHECK();
        return MXLO_OK;
      }
    }
    if (pair)
      hipLaunchKernelGGL((gemvb_t_kernel<T, CA, CB, B0, KB, true>), dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                         res, ldr, M, m, n, ld, V, ldv, (CA)alpha, (CB)beta);
    else
      hipLaunchKernelGGL((gemvb_t_kernel<T, CA, CB, B0, KB, false>), dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                         res, ldr, M, m, n, ld, V, ldv, (CA)alpha, (CB)beta);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

template <typename T>
int32_t gemv_block(mxlo_ctx *ctx, T *res, int64_t ldr, const T *M, int64_t m, int64_t n, int64_t ld, const T *V,
                   int64_t ldv, int64_t k, double alpha, double beta, int32_t mode, int32_t flags) {
  const int64_t nres = mode == MXLO_OP_N ? m : n, nin = mode == MXLO_OP_N ? n : m;
  if (nres == 0 || k == 0) return MXLO_OK;
  int64_t done = 0;
  while (done < k) {
    const int64_t left = k - done;
    T *r = res + done * ldr;
    const T *v = V + done * ldv;
    if (nin == 0) {                      // empty sum: column by column res = beta*res (or 0), as gemv_any
      MXLO_TRY(gemv_any<T>(ctx, r, M, m, n, ld, v, alpha, beta, mode, flags));
      done += 1;
    } else if (left >= 8) {
      MXLO_TRY((gemv_block_chunk<T, 8>(ctx, r, ldr, M, m, n, ld, v, ldv, alpha, beta, mode, flags)));
      done += 8;
    } else if (left >= 4) {
      MXLO_TRY((gemv_block_chunk<T, 4>(ctx, r, ldr, M, m, n, ld, v, ldv, alpha, beta, mode, flags)));
      done += 4;
    } else if (left >= 2) {
      MXLO_TRY((gemv_block_chunk<T, 2>(ctx, r, ldr, M, m, n, ld, v, ldv, alpha, beta, mode, flags)));
      done += 2;
    } else {
      MXLO_TRY(gemv_any<T>(ctx, r, M, m, n, ld, v, alpha, beta, mode, flags));   // a single column: the GEMV itself
      done += 1;
    }
  }
  return MXLO_OK;
}

// ---- opHermitian, single pass over the strict lower triangle -------------------------------------
// The triangle is cut into tiles of 256 rows x 32 columns (row group G, column tile J <= 8G+7); one workgroup
// owns a STRIP of 8 consecutive tiles of one row group (a 256 x 256 block; strip s <= G) and reads it ONCE —
// a wave-load is 1 KiB contiguous down one column. Per tile every wave produces, without any barrier,
//   cols  : Pcol[2G + half][32J + c] = sum over its 128 rows of L[r][c] * v[256G + r]      (part of L'*v)
// (halving butterfly across the 64 lanes) and keeps accumulating in registers
//   rows  : Prow[s][256G + r] = sum over the strip's columns of L[r][c] * v[c]             (part of L*v)
// which goes through LDS once per strip. Because waves run barrier-free over 8 tiles they drift apart and
// the reductions of one wave overlap the loads of the others (per-tile workgroups: 203 us, strips: 177 us
// at n = 16384, `tune_herm2.hip`). A second tiny kernel adds the partials in a FIXED order (deterministic,
// no float atomics) and applies   res = alpha*((d.*v + L*v) + L'*v) (+ beta*res)   (src/linalg.jl:99-101).
// HBM traffic: 4n^2 B for the triangle + ~0.1n^2 B of partials, vs 8n^2 for two triangular GEMVs and
// 16n^2 for the reference (two full GEMVs over tril(A,-1) stored with explicit zeros).
constexpr int HC = 32;             // tile columns
// A lane holds 16 bytes of each column: RPL = 2 rows (f64) or 4 rows (f32); 128 lanes span a row group of
// HR = 128*RPL rows, whose diagonal block is DT = HR/HC tiles wide.
template <typename T>
struct HermCfg {
  static constexpr int RPL = 16 / (int)sizeof(T);
  static constexpr int HR = 128 * RPL;
  static constexpr int DT = HR / HC;
};

// Partials live in row-group blocks, so that everything the finish kernel adds for one output row sits within a few
// hundred KiB (slot stride = HR doubles) instead of being strided by n doubles across tens of MiB:
//   Prow: row group G owns q*G + DT slots of HR doubles at herm_row_base(G);
//   Pcol: column group Gc owns the 2*(ng - Gc) row halves at / below it, HR doubles each, at herm_col_base(Gc).
template <int HR_, int DT_>
__host__ __device__ __forceinline__ int64_t herm_row_base(int64_t G, int q) {
  return (int64_t)HR_ * ((int64_t)q * (G * (G - 1) / 2) + (int64_t)DT_ * G);
}
template <int HR_>
__host__ __device__ __forceinline__ int64_t herm_col_base(int64_t Gc, int64_t ng) {
  return (int64_t)HR_ * (2 * ng * Gc - Gc * (Gc - 1));
}

// C = tiles per strip (8, 2 or 1: fewer when the triangle is too small to fill the chip with 256x256 strips),
// qint = 8/C_strip strips per 256 columns. Row group G owns qint*G + 8 slots of row partials: one per strip
// left of its diagonal block, one per tile of the diagonal block.
//   EDGE = false: strips of full row groups, every tile strictly below the diagonal and inside the matrix
//                 (no masking code at all, 16-byte loads);
//   EDGE = true : masked loads. mode 0: the same strips when A is not 16-byte aligned; mode 1: the strips of
//                 the ragged last row group; mode 2 (instantiated with C = 1): the 8 tiles of a diagonal block,
//                 one workgroup each, so that the short diagonal pass still fills the chip.
//   DSEL (with EDGE = false, C = 1): a tile of the diagonal block of a FULL row group of an aligned matrix — the same
//                 unmasked 16-byte loads (every address is inside the matrix), then elements at or above the diagonal
//                 are replaced by zero with a select (whatever the caller keeps up there, NaN included, never enters
//                 a product). Register footprint of the unmasked path, so these tiles ride in the interior launch.
// SLOT: the partials are handed to finisher workgroups of the SAME launch (herm_single_kernel): self-validating 64-bit
// agent-scope stores (an empty slot is a NaN payload no arithmetic produces; NaN partials are canonicalised), no fence.
__device__ __forceinline__ void herm_put(double *p, double v, bool slot) {
  if (slot) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (v != v) bits = kCanonicalNaN;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}

// KV > 1 (round 5, the block apply mxlo_hermitian_mul_block): the tile is loaded ONCE and applied to KV vectors — vector kk
// is v + kk * ldv, its partials live KV buffers apart (pstride doubles). Per vector the arithmetic, the butterfly and the
// order of every addition are those of KV = 1: a block apply is bit-identical to KV single applies.
template <typename T, int C, bool EDGE, bool DSEL = false, bool SLOT = false, int KV = 1, bool NT = true>
__device__ __forceinline__ void
herm_strip_body(const T *__restrict__ A, int64_t lda, const T *__restrict__ v, int64_t n,
                double *__restrict__ Prow, double *__restrict__ Pcol, int64_t ng, int qint, int mode, int64_t t,
                int64_t ldv = 0, int64_t pstride = 0, int colmajor_g = 0) {
  constexpr int RPL = HermCfg<T>::RPL, HR = HermCfg<T>::HR, DT = HermCfg<T>::DT;
  typedef T VR __attribute__((ext_vector_type(RPL)));
  constexpr int HS = C;
  static_assert(!DSEL || (!EDGE && C == 1), "DSEL: one unmasked-load tile of a diagonal block");
  int64_t G, tile0, slot;                    // row group, first column tile, row-partial slot
  if constexpr (DSEL) {                      // diagonal block of row group t/DT, tile t%DT
    G = t / DT;
    tile0 = DT * G + t % DT;
    slot = (int64_t)qint * G + t % DT;
  } else if (!EDGE || mode == 0) {           // triangular enumeration u = G'(G'+1)/2 + r, r <= G'; G = G'+1
    constexpr int Q = DT / C;
    // colmajor_g = g > 0 (the interior strips of g full row groups; tune key herm_order): the (row group, column block) pairs
    // are walked COLUMN BLOCK by column block — workgroups that run at the same time then read the same 256 columns in
    // consecutive row groups, i.e. long contiguous runs down each column (what the row-band GEMV streams), instead of 2-KiB
    // pieces 128 KiB apart along one row group. The enumeration is the triangular one read backwards; (G, slot) and with
    // them every partial index are the same function of the pair: results do not depend on the order.
    const int64_t u0 = t / Q;
    const int64_t u = colmajor_g > 0 ? (int64_t)colmajor_g * (colmajor_g - 1) / 2 - 1 - u0 : u0;
    int64_t Gp = (int64_t)((sqrt(8.0 * (double)u + 1.0) - 1.0) * 0.5);
    while (Gp * (Gp + 1) / 2 > u) --Gp;
    while ((Gp + 1) * (Gp + 2) / 2 <= u) ++Gp;
    const int64_t rem = u - Gp * (Gp + 1) / 2;            // <= Gp
    int64_t cb;                                           // column block (of 256 columns) < G
    if (colmajor_g > 0) {
      cb = colmajor_g - 2 - Gp;                           // column block cb holds the colmajor_g - 1 - cb pairs G = cb + 1 ..
      G = cb + 1 + (Gp - rem);
    } else {
      G = Gp + 1;
      cb = rem;
    }
    slot = cb * Q + t % Q;                                // strip s < Q*G
    tile0 = slot * C;
  } else if (mode == 1) {                    // strips left of the diagonal block of the last row group
    G = ng - 1;
    slot = t;
    tile0 = slot * C;
  } else {                                   // diagonal block of row group t/DT, tile t%DT
    G = t / DT;
    tile0 = DT * G + t % DT;
    slot = (int64_t)qint * G + t % DT;
  }
  const int64_t i0 = G * HR;
  const int tid = threadIdx.x, lane = tid & 63, rp = tid & 127;   // rows RPL*rp .. RPL*rp+RPL-1 of the row group
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 7);          // columns cg + 2k, k < 16, of each tile
  const int half = __builtin_amdgcn_readfirstlane((tid >> 6) & 1);  // which 64*RPL rows this wave covers
  const int64_t gr = i0 + RPL * rp;
  double vr[KV][RPL], prow[KV][RPL];
#pragma unroll
  for (int kk = 0; kk < KV; ++kk)
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      vr[kk][r] = (!EDGE || gr + r < n) ? (double)v[gr + r + kk * ldv] : 0.0;
      prow[kk][r] = 0.0;
    }
  // One tile = load_tile (16 loads of 16 bytes per lane) + compute_tile (FMAs, column butterfly, column-partial stores).
  auto load_tile = [&](VR (&e)[16], const int64_t j0) __attribute__((always_inline)) {
    if constexpr (!EDGE) {                   // every element is strictly below the diagonal and inside
      const T *base = A + (j0 + cg) * lda + gr;
      // NT: a triangle of about the size of the Infinity Cache is loaded with the default policy — the next apply of the same
      // operator (a Krylov loop) finds part of it there —, a larger one with the nontemporal hint (as the dense GEMVs). A
      // template parameter: under a run-time flag the compiler merges the two load sequences and drops the hint from both.
      if constexpr (DSEL) {
        // A wave whose 64*RPL rows all lie at or above the tile's first column requests nothing (wave-uniform test; a per-lane
        // test costs 16 exec-mask branches in front of the loads of the tiles that end the launch, and measured slower).
        const int below = (int)(gr - (j0 + cg));      // row - column of this lane's element of column k = 0
        const bool wave_above = i0 + (int64_t)(half + 1) * (HR / 2) <= j0;
#pragma unroll
        for (int k = 0; k < 16; ++k) e[k] = VR(T(0));
        if (!wave_above) {
#pragma unroll
          for (int k = 0; k < 16; ++k)
            e[k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const VR *>(base + (int64_t)(2 * k) * lda))
                      : *reinterpret_cast<const VR *>(base + (int64_t)(2 * k) * lda);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
          for (int r = 0; r < RPL; ++r) e[k][r] = below + r > 2 * k ? e[k][r] : T(0);   // strict lower triangle only
        }
      } else if constexpr (NT) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          e[k] = __builtin_nontemporal_load(reinterpret_cast<const VR *>(base + (int64_t)(2 * k) * lda));
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) e[k] = *reinterpret_cast<const VR *>(base + (int64_t)(2 * k) * lda);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int64_t gc = j0 + cg + 2 * k;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          T x = 0;
          if (gc < n && gr + r > gc && gr + r < n) x = A[gr + r + gc * lda];   // strict lower triangle only
          e[k][r] = x;
        }
      }
    }
  };
  auto compute_tile = [&](const VR (&e)[16], const int64_t j0) __attribute__((always_inline)) {
    // column partials of this tile: block of column group Gc = j0/HR, row half 2G + half (offset so that + gc indexes it)
    const int64_t Gc = j0 / HR;
    const int64_t pcol0 = herm_col_base<HR>(Gc, ng) + (2 * G + half - 2 * Gc) * HR - Gc * HR;
#pragma unroll
   for (int kk = 0; kk < KV; ++kk) {          // fully unrolled: kk is a compile-time index into prow / vr (KV = 1: a single trip)
    const T *vk = v + kk * ldv;
    double *Pck = Pcol + kk * pstride;
    // FMAs + first butterfly stage, column pair (q, q+8) at a time (keeps the live set small)
    double w8[8], w4[4], w2[2], w1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t ca = j0 + cg + 2 * q, cb = ca + 16;
      const double vca = (!EDGE || ca < n) ? (double)vk[ca] : 0.0, vcb = (!EDGE || cb < n) ? (double)vk[cb] : 0.0;
      double pa = 0.0, pb = 0.0;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const double a = (double)e[q][r], b = (double)e[q + 8][r];
        prow[kk][r] = fma(a, vca, prow[kk][r]);
        prow[kk][r] = fma(b, vcb, prow[kk][r]);
        pa = r == 0 ? a * vr[kk][0] : fma(a, vr[kk][r], pa);
        pb = r == 0 ? b * vr[kk][0] : fma(b, vr[kk][r], pb);
      }
      w8[q] = halve_step32(pa, pb);           // lanes < 32: column q, rows of lanes l and l + 32; lanes >= 32: column q + 8
      __builtin_amdgcn_sched_barrier(0);
    }
    // remaining halving steps (permlane16 swap, then DPP row moves: no LDS round trips — common.h): 4 + 2 + 1 exchanges leave
    // one column per group of 4 lanes, two quad permutes finish. Operands and order of every add are those of the
    // xor-shuffle butterfly this replaces: same bits.
#pragma unroll
    for (int q = 0; q < 4; ++q) w4[q] = halve_step16(w8[q], w8[4 + q]);
#pragma unroll
    for (int q = 0; q < 2; ++q) w2[q] = halve_step8(w4[q], w4[2 + q]);
    w1 = halve_step4(w2[0], w2[1]);
    w1 = pair_step2(w1);
    w1 = pair_step1(w1);
    if ((lane & 3) == 0) {
      const int k = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
      const int64_t gc = j0 + cg + 2 * k;
      if (!EDGE || gc < n) herm_put(Pck + pcol0 + gc, w1, SLOT);
    }
   }
  };
  // (Round 6, negative: two register sets — the loads of tile jt + 1 issued before tile jt is consumed — change nothing for the
  //  single apply and cost the KV = 4 block form its registers, 195 -> 317 us at n = 16384: profiles/r06_herm_policy.txt. The
  //  strips are not latency-bound; occupancy 5 -> 2 workgroups per CU had shown the same.)
#pragma unroll 1
  for (int jt = 0; jt < HS; ++jt) {
    const int64_t j0 = (tile0 + jt) * HC;
    if (EDGE && j0 >= n) break;              // ragged last row group: tiles past the matrix
    VR e[16];
    load_tile(e, j0);
    compute_tile(e, j0);
  }
  __shared__ double rowred[KV][2][HR];
#pragma unroll
  for (int kk = 0; kk < KV; ++kk)
#pragma unroll
    for (int r = 0; r < RPL; ++r) rowred[kk][cg][RPL * rp + r] = prow[kk][r];
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < KV; ++kk)
    for (int tt = tid; tt < HR; tt += kBlock) {
      const int64_t row = i0 + tt;
      if (row < n) herm_put(Prow + kk * pstride + herm_row_base<HR, DT>(G, qint) + slot * HR + tt, rowred[kk][0][tt] + rowred[kk][1][tt], SLOT);
    }
}

// ONE launch for everything that can use unmasked 16-byte loads (aligned A): workgroups [0, n_int) run the interior
// strips of the full row groups, the rest the tiles of their diagonal blocks (DSEL). Both bodies fit the unmasked
// path's registers, so at n = 4096 all 1088 workgroups are resident at once (the round-2 merged kernel also carried
// the masked bodies — 195 VGPRs, half the occupancy — and took 14.8 us of the 20.6 us apply).
template <typename T, int C, bool NT>
__global__ void __launch_bounds__(kBlock)
herm_pass_kernel(const T *__restrict__ A, int64_t lda, const T *__restrict__ v, int64_t n,
                 double *__restrict__ Prow, double *__restrict__ Pcol, int64_t ng, int qint, int64_t n_int, int colmajor_g) {
  const int64_t t = blockIdx.x;
  if (t < n_int) return herm_strip_body<T, C, false, false, false, 1, NT>(A, lda, v, n, Prow, Pcol, ng, qint, 0, t, 0, 0, colmajor_g);
  herm_strip_body<T, 1, false, true, false, 1, NT>(A, lda, v, n, Prow, Pcol, ng, qint, 2, t - n_int);
}

// The masked remainder, one launch: every strip when A is not 16-byte aligned (n_all, mode 0), the strips of the
// ragged last row group (n_last, mode 1), and the diagonal tiles of the row groups from g0 on (mode 2).
template <typename T, int C>
__global__ void __launch_bounds__(kBlock)
herm_edge_kernel(const T *__restrict__ A, int64_t lda, const T *__restrict__ v, int64_t n,
                 double *__restrict__ Prow, double *__restrict__ Pcol, int64_t ng, int qint, int64_t n_all,
                 int64_t n_last, int64_t g0) {
  int64_t t = blockIdx.x;
  if (t < n_all) return herm_strip_body<T, C, true>(A, lda, v, n, Prow, Pcol, ng, qint, 0, t);
  t -= n_all;
  if (t < n_last) return herm_strip_body<T, C, true>(A, lda, v, n, Prow, Pcol, ng, qint, 1, t);
  t -= n_last;
  herm_strip_body<T, 1, true>(A, lda, v, n, Prow, Pcol, ng, qint, 2, t + HermCfg<T>::DT * g0);
}

// The same two launches for a BLOCK of KV vectors (mxlo_hermitian_mul_block): every tile of the triangle is read once.
template <typename T, int C, int KV, bool NT>
__global__ void __launch_bounds__(kBlock)
herm_pass_block_kernel(const T *__restrict__ A, int64_t lda, const T *__restrict__ v, int64_t ldv, int64_t n,
                       double *__restrict__ Prow, double *__restrict__ Pcol, int64_t pstride, int64_t ng, int qint, int64_t n_int,
                       int colmajor_g) {
  const int64_t t = blockIdx.x;
  if (t < n_int) return herm_strip_body<T, C, false, false, false, KV, NT>(A, lda, v, n, Prow, Pcol, ng, qint, 0, t, ldv, pstride, colmajor_g);
  herm_strip_body<T, 1, false, true, false, KV, NT>(A, lda, v, n, Prow, Pcol, ng, qint, 2, t - n_int, ldv, pstride);
}
template <typename T, int C, int KV>
__global__ void __launch_bounds__(kBlock)
herm_edge_block_kernel(const T *__restrict__ A, int64_t lda, const T *__restrict__ v, int64_t ldv, int64_t n,
                       double *__restrict__ Prow, double *__restrict__ Pcol, int64_t pstride, int64_t ng, int qint, int64_t n_all,
                       int64_t n_last, int64_t g0) {
  int64_t t = blockIdx.x;
  if (t < n_all) return herm_strip_body<T, C, true, false, false, KV>(A, lda, v, n, Prow, Pcol, ng, qint, 0, t, ldv, pstride);
  t -= n_all;
  if (t < n_last) return herm_strip_body<T, C, true, false, false, KV>(A, lda, v, n, Prow, Pcol, ng, qint, 1, t, ldv, pstride);
  t -= n_last;
  herm_strip_body<T, 1, true, false, false, KV>(A, lda, v, n, Prow, Pcol, ng, qint, 2, t + HermCfg<T>::DT * g0, ldv, pstride);
}

// 32 rows per workgroup, 8 lanes per row: lane `sub` adds partials sub, sub+8, ... (independent loads in
// flight), the 8 sub-sums are combined in a fixed order -> deterministic, and n/32 workgroups fill the chip.
// POLL (herm_single_kernel): the partials are slots written by strip workgroups of the SAME launch — each slot is read by
// exactly one lane, which waits for it (bounded, poll_slot) and then re-arms it for the next apply. Same order of
// additions as the separate finish launch: the two forms are bit-identical.
template <typename T, typename CA, typename CB, bool BETA0, int FR, bool POLL>
__device__ __forceinline__ void
herm_finish_body(T *__restrict__ res, const T *__restrict__ d, const T *__restrict__ v, double *__restrict__ Prow,
                 double *__restrict__ Pcol, int64_t n, int ng, int q, CA alpha, CB beta, int64_t blk,
                 unsigned long long ticks, unsigned *__restrict__ fault, int poll_sleep = 32) {
  // FR rows per workgroup, 256/FR lanes per row; a lane's partials (sub, sub+FS, ...) are loaded 8 (4) at a time with
  // nothing between the loads, and the finishing lane's d, v (res) are requested before them: the kernel is pure
  // latency (3 MB of partials at n = 4096) — ~2.3 us of its own plus the ~2.2 us every dependent launch costs.
  constexpr int FS = kBlock / FR;
  const int r = threadIdx.x % FR, sub = threadIdx.x / FR;
  const int64_t i = blk * FR + r;
  __shared__ double s1[FS][FR], s2[FS][FR];
  double t1 = 0.0, t2 = 0.0;
  T di = T(0), vi = T(0), ri = T(0);           // the finishing lane's operands, requested before the partials
  if (sub == 0 && i < n) {
    di = d[i];
    vi = v[i];
    if constexpr (!BETA0) ri = res[i];
  }
  auto take = [&](double *p) -> double {       // POLL: first look (all of a batch in flight), the wait happens in settle()
    if constexpr (POLL) return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else return *p;
  };
  auto settle = [&](double *p, double x) -> double {
    if constexpr (POLL) {
      unsigned long long bits = (unsigned long long)__double_as_longlong(x);
      if (bits == kSlotEmpty) bits = poll_slot(reinterpret_cast<unsigned long long *>(p), ticks, fault, kFaultHermitian);
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), kSlotEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
      return __longlong_as_double((long long)bits);
    } else {
      return x;
    }
  };
  if constexpr (POLL) {
    // Before the general sweep: two lanes per row wait — with long sleeps, so that 128 waiting workgroups do not load the
    // memory system the strips are streaming through — for the two partials that are produced LAST for this row: the row
    // partial of the last tile of its diagonal block and the column partial from the last row half of the matrix (the
    // diagonal tiles are the highest-numbered strip workgroups, the last row group's the highest of those).
    if (i < n && sub < 2) {
      constexpr int HR = HermCfg<T>::HR, DT = HermCfg<T>::DT;
      const int G = (int)(i / HR);
      const double *w = sub == 0 ? Prow + herm_row_base<HR, DT>(G, q) + (i - (int64_t)G * HR) + (int64_t)(q * G + DT - 1) * HR
                                 : Pcol + herm_col_base<HR>(G, ng) + (i - (int64_t)G * HR) + (int64_t)(2 * ng - 1 - 2 * G) * HR;
      unsigned long long t0 = 0;
      unsigned it = 0;
      while (__hip_atomic_load(reinterpret_cast<const unsigned long long *>(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kSlotEmpty) {
        for (int k = 0; k < poll_sleep; ++k) __builtin_amdgcn_s_sleep(1);      // (tune key herm_poll_sleep x 64 clocks between two looks)
        if ((++it & 63u) == 0) {                // bounded like poll_slot; the sweep below reports the fault
          const unsigned long long now = (unsigned long long)wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > ticks) break;
        }
      }
    }
    __syncthreads();
  }
  if (i < n) {
    constexpr int HR = HermCfg<T>::HR, DT = HermCfg<T>::DT;
    const int G = (int)(i / HR);
    const int c1 = q * G + DT;                 // L*v : q*G strips + DT diagonal tiles
    double *prow = Prow + herm_row_base<HR, DT>(G, q) + (i - (int64_t)G * HR);
    double *pcol = Pcol + herm_col_base<HR>(G, ng) + (i - (int64_t)G * HR);
    // Round 6: the FIRST batch of row partials (16 per lane) and of column partials (8 per lane) are requested together, before
    // anything is added — at n = 4096 that is every partial of the row, one memory round trip instead of three dependent ones
    // (the partials come from the other XCDs' strips: each batch is a miss to the Infinity Cache). The order of the additions
    // (slots sub, sub + FS, ... ascending; + 0.0 for absent ones, which cannot change a sum that started at + 0.0) is unchanged.
    // The single-launch form (POLL) keeps batches of 8 / 4 under their conditions: its loads are agent-scope atomics to
    // memory, and the 14 duplicate ones of the clamped form cost 0.5 us at n = 2048.
    constexpr int RB = POLL ? 8 : 16, CBN = POLL ? 4 : 8;
    const int h1 = 2 * ng;                     // L'*v: 128-row halves at/below i
    double xr[RB], xc[CBN];
    // (unconditional loads from clamped slots: under a condition the compiler folds `0.0 + first partial` into the load's
    //  branch and waits for it there — one more round trip per sum)
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int sx = sub + FS * u;
      if constexpr (POLL) xr[u] = sx < c1 ? take(prow + (int64_t)sx * HR) : 0.0;
      else xr[u] = take(prow + (int64_t)(sx < c1 ? sx : c1 - 1) * HR);
    }
#pragma unroll
    for (int u = 0; u < CBN; ++u) {
      const int h = 2 * G + sub + FS * u;
      if constexpr (POLL) xc[u] = h < h1 ? take(pcol + (int64_t)(h - 2 * G) * HR) : 0.0;
      else xc[u] = take(pcol + (int64_t)((h < h1 ? h : h1 - 1) - 2 * G) * HR);
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int sx = sub + FS * u;
      t1 += sx < c1 ? settle(prow + (int64_t)sx * HR, xr[u]) : 0.0;
    }
    for (int base = sub + FS * RB; base < c1; base += FS * 8) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sx = base + FS * u;
        x[u] = sx < c1 ? take(prow + (int64_t)sx * HR) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sx = base + FS * u;
        t1 += sx < c1 ? settle(prow + (int64_t)sx * HR, x[u]) : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < CBN; ++u) {
      const int h = 2 * G + sub + FS * u;
      t2 += h < h1 ? settle(pcol + (int64_t)(h - 2 * G) * HR, xc[u]) : 0.0;
    }
    for (int base = 2 * G + sub + FS * CBN; base < h1; base += FS * 4) {
      double x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int h = base + FS * u;
        x[u] = h < h1 ? take(pcol + (int64_t)(h - 2 * G) * HR) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int h = base + FS * u;
        t2 += h < h1 ? settle(pcol + (int64_t)(h - 2 * G) * HR, x[u]) : 0.0;
      }
    }
  }
  s1[sub][r] = t1;
  s2[sub][r] = t2;
  __syncthreads();
  if (sub == 0 && i < n) {
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int k = 0; k < FS; ++k) {
      a1 += s1[k][r];
      a2 += s2[k][r];
    }
    const T inner = ((di * vi) + (T)a1) + (T)a2;
    res[i] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)inner, beta, ri);
  }
}

template <typename T, typename CA, typename CB, bool BETA0, int FR>
__global__ void __launch_bounds__(kBlock)
herm_finish_kernel(T *__restrict__ res, const T *__restrict__ d, const T *__restrict__ v,
                   double *__restrict__ Prow, double *__restrict__ Pcol, int64_t n, int ng,
                   int q, CA alpha, CB beta) {
  herm_finish_body<T, CA, CB, BETA0, FR, false>(res, d, v, Prow, Pcol, n, ng, q, alpha, beta, (int64_t)blockIdx.x, 0ull, nullptr);
}

template <typename T, typename CA, typename CB, bool BETA0, int FR>
__global__ void __launch_bounds__(kBlock)
herm_finish_block_kernel(T *__restrict__ res, int64_t ldr, const T *__restrict__ d, const T *__restrict__ v, int64_t ldv,
                         double *__restrict__ Prow, double *__restrict__ Pcol, int64_t pstride, int64_t n, int ng, int q, CA alpha,
                         CB beta) {
  const int64_t kk = blockIdx.y;
  herm_finish_body<T, CA, CB, BETA0, FR, false>(res + kk * ldr, d, v + kk * ldv, Prow + kk * pstride, Pcol + kk * pstride, n, ng, q, alpha,
                                                beta, (int64_t)blockIdx.x, 0ull, nullptr);
}

// The WHOLE apply in one launch (round 5; n a multiple of the row-group height, aligned A, n <= herm_single_max_n):
// workgroups [0, n_int) interior strips, [n_int, n_light) diagonal tiles, [n_light, n_light + n/FR) FINISHERS. A finisher
// is dispatched after every strip workgroup (in-order dispatch of a 1-D grid), waits for exactly the slots its rows
// need and re-arms them; the dependent finish launch (~2.2 us of launch + 2.3 us of latency at n = 4096, of 17.8 us) and
// the fences / tickets of the forms tried in rounds 2-3 are gone: store -> load through the self-validating slots is all
// that connects producer and consumer.
template <typename T, int C, typename CA, typename CB, bool BETA0, int FR>
__global__ void __launch_bounds__(kBlock)
herm_single_kernel(T *__restrict__ res, const T *__restrict__ d, const T *__restrict__ A, int64_t lda, const T *__restrict__ v,
                   int64_t n, double *__restrict__ Prow, double *__restrict__ Pcol, int64_t ng, int qint, int64_t n_int,
                   int64_t n_light, CA alpha, CB beta, unsigned long long ticks, unsigned *__restrict__ fault, int colmajor_g, int poll_sleep) {
  const int64_t t = blockIdx.x;
  if (t < n_int) return herm_strip_body<T, C, false, false, true>(A, lda, v, n, Prow, Pcol, ng, qint, 0, t, 0, 0, colmajor_g);
  if (t < n_light) return herm_strip_body<T, 1, false, true, true>(A, lda, v, n, Prow, Pcol, ng, qint, 2, t - n_int);
  herm_finish_body<T, CA, CB, BETA0, FR, true>(res, d, v, Prow, Pcol, n, (int)ng, qint, alpha, beta, t - n_light, ticks, fault, poll_sleep);
}

__global__ void __launch_bounds__(kBlock) herm_slots_fill_kernel(unsigned long long *__restrict__ p, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock) p[i] = kSlotEmpty;
}

// Cache policy of the strip loads (tune key herm_nt: -1 = this rule, 0 / 1 force): nontemporal, EXCEPT for triangles of about
// the size of the 256 MiB Infinity Cache — [herm_dp_min_bytes, herm_nt_min_bytes) = [96, 384) MiB — where the next apply of
// the same operator (a Krylov loop) finds part of a default-policy stream still there. Measured, old vs new library in one
// run (profiles/r06_herm_policy.txt): f64 n = 6144 29.8 -> 28.2 us, 8192 49.9 -> 47.3; 4096 / 5120 neutral; below (3072) the
// default policy is 10 % SLOWER, above (16384) 9 % slower.
template <typename T>
inline int herm_nt_policy(const mxlo_ctx *ctx, int64_t n) {
  if (ctx->tune.herm_nt >= 0) return ctx->tune.herm_nt;
  const int64_t tri = (int64_t)sizeof(T) * n * (n / 2);
  return tri >= ctx->tune.herm_dp_min_bytes && tri < ctx->tune.herm_nt_min_bytes ? 0 : 1;
}

// Tiles per strip: 8-tile strips once there are two of them per CU, 2-tile strips with two per CU, else single tiles. Triangles
// small enough for the single-launch form (herm_small) take 2-tile strips from 1.25 per CU on — measured with the one launch
// (profiles/r06_herm_policy.txt): f64 n = 4096 14.8 -> 14.4 us, f32 n = 5120 14.1 -> 13.6; f64 n = 3072 (1.03 per CU) 10.1 -> 11.0:
// stays single tiles. The rule depends on (n, T, CUs) only, so the single launch and the two launches share their partial
// layout and stay bit-identical.
template <typename T>
inline bool herm_small(const mxlo_ctx *ctx, int64_t n) {
  if (ctx->tune.herm_single_max_n > 0) return n <= ctx->tune.herm_single_max_n;
  return (int64_t)sizeof(T) * n * (n / 2) <= ctx->tune.herm_single_max_bytes;
}
template <typename T>
inline int herm_strip_tiles(const mxlo_ctx *ctx, int64_t n, int64_t pairs) {
  constexpr int DT = HermCfg<T>::DT;
  if (ctx->tune.herm_strip) return ctx->tune.herm_strip;
  if ((DT / 8) * pairs >= 2 * ctx->num_cu) return 8;
  if ((DT / 2) * pairs * 4 >= (herm_small<T>(ctx, n) ? 5 : 8) * (int64_t)ctx->num_cu) return 2;
  return 1;
}

template <typename T>
int32_t hermitian_t(mxlo_ctx *ctx, T *res, const T *d, const T *A, int64_t lda, const T *v, int64_t n,
                    double alpha, double beta, int32_t flags) {
  if (n == 0) return MXLO_OK;
  constexpr int HR = HermCfg<T>::HR, DT = HermCfg<T>::DT, RPL = HermCfg<T>::RPL;
  const int64_t ng = (n + HR - 1) / HR, ngf = n / HR;
  MXLO_REQUIRE(4 * ng * (ng + 1) < (1LL << 31), MXLO_ESHAPE, "opHermitian: n too large");
  // tiles per strip: 8-tile strips once there are at least two of them per CU, thinner strips below that
  const int64_t pairs = ng * (ng - 1) / 2;                       // (row group, strip column block) pairs left of the diagonal
  const int C = herm_strip_tiles<T>(ctx, n, pairs), Q = DT / C;
  const int64_t prow_len = herm_row_base<HR, DT>(ng, Q), pcol_len = herm_col_base<HR>(ng, ng);
  const size_t need = sizeof(double) * (size_t)(prow_len + pcol_len);
  const bool aligned = (((uintptr_t)A & 15u) == 0) && (lda % RPL == 0);
  const int nt = herm_nt_policy<T>(ctx, n);
  // ---- the whole apply in ONE launch: full row groups of an aligned matrix, slots in their own (always re-armed) buffer
  if (ctx->tune.herm_single && aligned && n % HR == 0 && herm_small<T>(ctx, n) && ctx->fault_dev && !ctx->capturing) {
    int32_t fst = fused_fault_check(ctx);
    if (fst != MXLO_OK) return fst;
    if (ctx->tune.herm_single) {              // (the fault check switches the single-launch forms off)
      if (ctx->herm_slots_bytes < need || ctx->herm_slots_layout != (int64_t)n * 64 + C * 2 + (sizeof(T) == 8 ? 1 : 0) || ctx->herm_slots_dirty) {
        if (ctx->herm_slots_bytes < need) {
          if (ctx->herm_slots) {
            MXLO_HIP(hipStreamSynchronize(ctx->stream));
            MXLO_HIP(hipFree(ctx->herm_slots));
          }
          ctx->herm_slots = nullptr;
          ctx->herm_slots_bytes = 0;
          hipError_t e = hipMalloc((void **)&ctx->herm_slots, need);
          MXLO_REQUIRE(e == hipSuccess, MXLO_ENOMEM, "opHermitian slots: %s", hipGetErrorString(e));
          ctx->herm_slots_bytes = need;
        }
        // every slot empty: a new buffer, another (n, strip shape) than the last apply used (the slots an apply fills and
        // the slots its finishers re-arm are the same set only for one layout), or after a timed-out apply
        hipLaunchKernelGGL(herm_slots_fill_kernel, dim3(ctx->num_cu * 4), dim3(kBlock), 0, ctx->stream,
                           (unsigned long long *)ctx->herm_slots, (int64_t)(ctx->herm_slots_bytes / 8));
        MXLO_LAUNCH_CHECK();
        ctx->herm_slots_layout = (int64_t)n * 64 + C * 2 + (sizeof(T) == 8 ? 1 : 0);
        ctx->herm_slots_dirty = false;
      }
      double *Sr = ctx->herm_slots, *Sc = Sr + prow_len;
      const int64_t n_int1 = ngf > 1 ? Q * ngf * (ngf - 1) / 2 : 0, n_light1 = n_int1 + (int64_t)DT * ngf;
      constexpr int FR1 = 32;
      const int64_t grid1 = n_light1 + (n + FR1 - 1) / FR1;
      bool fits = true;
      const int32_t st1 = dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
        auto go = [&]<int C_>() -> int32_t {
          // (no co-residency requirement: the strip workgroups wait for nobody, and the finishers — the last workgroups of the
          //  1-D grid — are dispatched after all of them)
          hipLaunchKernelGGL((herm_single_kernel<T, C_, CA, CB, B0, FR1>), dim3((unsigned)grid1), dim3(kBlock), 0, ctx->stream, res, d, A,
                             lda, v, n, Sr, Sc, ng, Q, n_int1, n_light1, (CA)alpha, (CB)beta, fused_timeout_ticks(ctx), ctx->fault_dev,
                             ctx->tune.herm_order && n > 2048 ? (int)ngf : 0, ctx->tune.herm_poll_sleep);
          MXLO_LAUNCH_CHECK();
          return MXLO_OK;
        };
        return C == 8 ? go.template operator()<8>() : (C == 2 ? go.template operator()<2>() : go.template operator()<1>());
      });
      if (st1 != MXLO_OK || fits) return st1;
    }
  }
  if (ctx->scratch_bytes < need) {            // stream-ordered users only: drain before the buffer is replaced
    if (ctx->scratch) {
      MXLO_HIP(hipStreamSynchronize(ctx->stream));
      MXLO_HIP(hipFree(ctx->scratch));
    }
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    hipError_t e = hipMalloc(&ctx->scratch, need);
    MXLO_REQUIRE(e == hipSuccess, MXLO_ENOMEM, "opHermitian scratch: %s", hipGetErrorString(e));
    ctx->scratch_bytes = need;
    ++ctx->scratch_generation;     // graphs that recorded the old workspace pointer are stale now
  }
  if (ctx->capturing) ctx->scratch_used_in_capture = true;
  double *Prow = (double *)ctx->scratch, *Pcol = Prow + prow_len;
  // full row groups (aligned A): interior strips + diagonal tiles in the unmasked launch; everything else is masked
  const int64_t gi = aligned ? ngf : 0;
  const int64_t n_int = gi > 1 ? Q * gi * (gi - 1) / 2 : 0;
  const int64_t n_dsel = (int64_t)DT * gi;
  const int64_t n_all = !aligned && ng > 1 ? Q * ng * (ng - 1) / 2 : 0;       // mode 0, masked
  const int64_t n_last = aligned && ng > ngf ? Q * (ng - 1) : 0;               // mode 1
  const int64_t n_diag = (int64_t)DT * (ng - gi);                              // mode 2, row groups gi .. ng-1
  const int64_t n_light = n_int + n_dsel, n_edge = n_all + n_last + n_diag;
  MXLO_REQUIRE(n_light < (1LL << 31) && n_edge < (1LL << 31), MXLO_ESHAPE, "opHermitian: n too large");
#define HERM_LAUNCH(C_)                                                                                          \
  {                                                                                                              \
    if (n_light > 0) {                                                                                           \
      if (nt)                                                                                                    \
        hipLaunchKernelGGL((herm_pass_kernel<T, C_, true>), dim3((unsigned)n_light), dim3(kBlock), (size_t)ctx->tune.herm_lds_pad, ctx->stream, A, lda, v, \
                           n, Prow, Pcol, ng, Q, n_int, ctx->tune.herm_order ? (int)gi : 0);                     \
      else                                                                                                       \
        hipLaunchKernelGGL((herm_pass_kernel<T, C_, false>), dim3((unsigned)n_light), dim3(kBlock), (size_t)ctx->tune.herm_lds_pad, ctx->stream, A, lda, v, \
                           n, Prow, Pcol, ng, Q, n_int, ctx->tune.herm_order ? (int)gi : 0);                     \
      MXLO_LAUNCH_CHECK();                                                                                       \
    }                                                                                                            \
    if (n_edge > 0) {                                                                                            \
      hipLaunchKernelGGL((herm_edge_kernel<T, C_>), dim3((unsigned)n_edge), dim3(kBlock), 0, ctx->stream, A, lda, v,  \
                         n, Prow, Pcol, ng, Q, n_all, n_last, gi);                                               \
      MXLO_LAUNCH_CHECK();                                                                                       \
    }                                                                                                            \
  }
  if (C == 8) HERM_LAUNCH(8) else if (C == 2) HERM_LAUNCH(2) else HERM_LAUNCH(1)
#undef HERM_LAUNCH
  constexpr int FR = 32;                       // rows per finishing workgroup (8 ... 64 measured alike)
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    hipLaunchKernelGGL((herm_finish_kernel<T, CA, CB, B0, FR>), dim3((unsigned)((n + FR - 1) / FR)), dim3(kBlock), 0,
                       ctx->stream, res, d, v, Prow, Pcol, n, (int)ng, Q, (CA)alpha, (CB)beta);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

// res[:, c] = alpha * ((d .* V[:, c] + L V[:, c]) + L' V[:, c]) + beta * res[:, c], c < k — mulHermitian! (src/linalg.jl:97-103)
// applied to the columns of a matrix (src/operations.jl:34-36), the strict lower triangle read ONCE per chunk of up to 4
// columns instead of once per column. Same launch geometry, partial layout and summation order as hermitian_t per column.
template <typename T>
int32_t hermitian_block_t(mxlo_ctx *ctx, T *res, int64_t ldr, const T *d, const T *A, int64_t lda, const T *V, int64_t ldv,
                          int64_t n, int64_t k, double alpha, double beta, int32_t flags) {
  if (n == 0 || k == 0) return MXLO_OK;
  constexpr int HR = HermCfg<T>::HR, DT = HermCfg<T>::DT, RPL = HermCfg<T>::RPL;
  const int64_t ng = (n + HR - 1) / HR, ngf = n / HR;
  MXLO_REQUIRE(4 * ng * (ng + 1) < (1LL << 31), MXLO_ESHAPE, "opHermitian: n too large");
  const int64_t pairs = ng * (ng - 1) / 2;
  const int C = herm_strip_tiles<T>(ctx, n, pairs), Q = DT / C;
  const int64_t prow_len = herm_row_base<HR, DT>(ng, Q), pcol_len = herm_col_base<HR>(ng, ng);
  const int64_t pstride = prow_len + pcol_len;
  constexpr int KVMAX = 4;
  MXLO_TRY(ensure_scratch(ctx, sizeof(double) * (size_t)pstride * KVMAX, "opHermitian block"));
  double *Prow = (double *)ctx->scratch, *Pcol = Prow + prow_len;
  const bool aligned = (((uintptr_t)A & 15u) == 0) && (lda % RPL == 0);
  const int nt = herm_nt_policy<T>(ctx, n);
  const int64_t gi = aligned ? ngf : 0;
  const int64_t n_int = gi > 1 ? Q * gi * (gi - 1) / 2 : 0;
  const int64_t n_dsel = (int64_t)DT * gi;
  const int64_t n_all = !aligned && ng > 1 ? Q * ng * (ng - 1) / 2 : 0;
  const int64_t n_last = aligned && ng > ngf ? Q * (ng - 1) : 0;
  const int64_t n_diag = (int64_t)DT * (ng - gi);
  const int64_t n_light = n_int + n_dsel, n_edge = n_all + n_last + n_diag;
  MXLO_REQUIRE(n_light < (1LL << 31) && n_edge < (1LL << 31), MXLO_ESHAPE, "opHermitian: n too large");
  int64_t done = 0;
  while (done < k) {
    const int64_t left = k - done;
    const int kv = left >= 4 ? 4 : (left >= 2 ? 2 : 1);
    T *r = res + done * ldr;
    const T *v = V + done * ldv;
    if (kv == 1) {
      MXLO_TRY(hermitian_t<T>(ctx, r, d, A, lda, v, n, alpha, beta, flags));
      Prow = (double *)ctx->scratch;           // (hermitian_t may have grown the workspace — it cannot: ours is KVMAX times larger)
      Pcol = Prow + prow_len;
      done += 1;
      continue;
    }
    auto launch = [&]<int C_, int KV_>() -> int32_t {
      if (n_light > 0) {
        if (nt)
          hipLaunchKernelGGL((herm_pass_block_kernel<T, C_, KV_, true>), dim3((unsigned)n_light), dim3(kBlock), (size_t)ctx->tune.herm_lds_pad, ctx->stream, A, lda, v, ldv, n,
                             Prow, Pcol, pstride, ng, Q, n_int, ctx->tune.herm_order ? (int)gi : 0);
        else
          hipLaunchKernelGGL((herm_pass_block_kernel<T, C_, KV_, false>), dim3((unsigned)n_light), dim3(kBlock), (size_t)ctx->tune.herm_lds_pad, ctx->stream, A, lda, v, ldv, n,
                             Prow, Pcol, pstride, ng, Q, n_int, ctx->tune.herm_order ? (int)gi : 0);
        MXLO_LAUNCH_CHECK();
      }
      if (n_edge > 0) {
        hipLaunchKernelGGL((herm_edge_block_kernel<T, C_, KV_>), dim3((unsigned)n_edge), dim3(kBlock), 0, ctx->stream, A, lda, v, ldv, n,
                           Prow, Pcol, pstride, ng, Q, n_all, n_last, gi);
        MXLO_LAUNCH_CHECK();
      }
      return MXLO_OK;
    };
    auto by_c = [&]<int KV_>() -> int32_t {
      return C == 8 ? launch.template operator()<8, KV_>() : (C == 2 ? launch.template operator()<2, KV_>() : launch.template operator()<1, KV_>());
    };
    MXLO_TRY(kv == 4 ? by_c.template operator()<4>() : by_c.template operator()<2>());
    constexpr int FR = 32;
    MXLO_TRY((dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
      hipLaunchKernelGGL((herm_finish_block_kernel<T, CA, CB, B0, FR>), dim3((unsigned)((n + FR - 1) / FR), (unsigned)kv), dim3(kBlock), 0,
                         ctx->stream, r, ldr, d, v, ldv, Prow, Pcol, pstride, n, (int)ng, Q, (CA)alpha, (CB)beta);
      MXLO_LAUNCH_CHECK();
      return MXLO_OK;
    })));
    done += kv;
  }
  return MXLO_OK;
}

// The premise of the XCD-local kron fusion, probed ONCE per ctx (one 64-workgroup launch + a 256-byte copy): the workgroups
// of a 1-D grid are dealt to the 8 XCDs round-robin, i.e. workgroups whose ids agree modulo 8 share an XCD (and its L2).
// WHICH XCD workgroup 0 lands on varies — 0 in an otherwise idle process (tools/xcc_probe.hip), 5 inside bench.py, where
// other queues had been active — so the check is the PERIOD, not the phase: XCC_ID(i) == XCC_ID(i % 8) and the first eight
// are all different. (A launch that broke the period would not compute wrong results silently: producers and consumers of a
// row block would sit behind different L2s, the consumers' bounded wait would time out and raise the ctx fault word.)
__global__ void xcc_probe_kernel(unsigned *__restrict__ out) {
  if (threadIdx.x == 0) out[blockIdx.x] = gl_xcc_id();
}
bool xcd_map_ok(mxlo_ctx *ctx) {
  if (ctx->xcd_map == 0) {
    ctx->xcd_map = -1;
    unsigned *d = nullptr, h[64];
    if (hipMalloc((void **)&d, sizeof(h)) == hipSuccess) {
      hipLaunchKernelGGL(xcc_probe_kernel, dim3(64), dim3(64), 0, ctx->stream, d);
      if (hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
          hipStreamSynchronize(ctx->stream) == hipSuccess) {
        bool ok = true;
        unsigned seen = 0;
        for (int i = 0; i < 8; ++i) seen |= h[i] < 8 ? 1u << h[i] : 0u;
        ok = seen == 0xffu;                                   // eight XCDs, each once among the first eight workgroups
        for (int i = 0; i < 64; ++i) ok = ok && h[i] == h[i & 7];
        if (ok) ctx->xcd_map = 1;
        if (getenv("MXLO_DEBUG_XCC")) {
          fprintf(stderr, "[mxlo] xcc probe on stream %p:", (void *)ctx->stream);
          for (int i = 0; i < 64; ++i) fprintf(stderr, " %u", h[i]);
          fprintf(stderr, " -> %s\n", ok ? "period 8" : "NOT periodic");
        }
      }
      (void)hipFree(d);
    }
    (void)hipGetLastError();
  }
  return ctx->xcd_map == 1;
}

// kron(opA, opB) * x with opA = A or A^T, opB = B or B^T of the STORED column-major matrices (src/kron.jl:14-40):
//   opA is m x n, opB is p x q;  X = reshape(x, q, n);  Ut = opA * X^T (m x q);  R = opB * Ut^T (p x m);
//   res = alpha*vec(R) + beta*res.
// Both products have a B' operand that is N-contiguous (X as stored; Ut as just written), and an A operand that is
// M-contiguous (factor taken as stored) or K-contiguous (factor taken transposed): the two layouts of the DMA kernel,
// so prod!, tprod!, ctprod! and row-major (transposed-in-place) factors all run at the same rate, copy-free.
template <typename T>
int32_t kron_t(mxlo_ctx *ctx, T *res, const T *A, int64_t am, int64_t an, int64_t lda, bool trans_a, const T *B,
               int64_t bp, int64_t bq, int64_t ldb, bool trans_b, const T *x, T *work, double alpha, double beta,
               int32_t flags) {
  const int64_t m = trans_a ? an : am, n = trans_a ? am : an;
  const int64_t p = trans_b ? bq : bp, q = trans_b ? bp : bq;
  // ---- both products in ONE launch with an XCD-local dependency (gemm_glds.h: kron_fused_kernel; tune key kron_fuse):
  // same tile class for both (the one gemm() would pick for each), the same A layout, every tile of both products on its
  // own CU (grid <= #CU and co-resident), DMA preconditions as gemm().
  // (inside a graph capture only once a warm-up apply has probed the XCD map and allocated the counters: neither can happen there)
  if (ctx->tune.kron_fuse && trans_a == trans_b && ctx->fault_dev && (!ctx->capturing || (ctx->xcd_map == 1 && ctx->kron_cnt_n > 0)) && m > 0 && n > 0 && p > 0 && q > 0 &&
      m < (1LL << 31) && n < (1LL << 31) && p < (1LL << 31) && q < (1LL << 31) && lda < (1LL << 22) && ldb < (1LL << 22) &&
      m < (1LL << 22) && q < (1LL << 22) && gemm_glds_ok<T>(A, lda, trans_a, x, q, m, q, n) &&
      gemm_glds_ok<T>(B, ldb, trans_b, work, m, p, m, q)) {
    auto tile_of = [&](int64_t M, int64_t N) {
      auto tiles = [&](int tm, int tn) { return ((M + tm - 1) / tm) * ((N + tn - 1) / tn); };
      if (tiles(128, 128) >= ctx->num_cu) return 128;
      return tiles(64, 64) * 5 >= (int64_t)ctx->num_cu * 3 ? 64 : 32;
    };
    const int t1 = tile_of(m, q), t2 = tile_of(p, m);
    if (t1 == t2 && t1 != 128) {
      const int tile = t1;
      const int64_t nrb = (m + tile - 1) / tile, gy1 = (q + tile - 1) / tile, gx2 = (p + tile - 1) / tile;
      const int64_t per_rb = gy1 > gx2 ? gy1 : gx2, per_xcd = ((nrb + 7) / 8) * per_rb, grid = 8 * per_xcd;
      int32_t fst = MXLO_OK;
      if (grid <= ctx->num_cu && nrb <= 64 && xcd_map_ok(ctx) && (fst = fused_fault_check(ctx)) == MXLO_OK && ctx->tune.kron_fuse) {
        if (ctx->kron_cnt_n < 2 * 64 * kGlFuseStride) {
          if (ctx->kron_cnt) {
            MXLO_HIP(hipStreamSynchronize(ctx->stream));
            MXLO_HIP(hipFree(ctx->kron_cnt));
            ctx->kron_cnt = nullptr;
            ctx->kron_cnt_n = 0;
          }
          MXLO_HIP(hipMalloc((void **)&ctx->kron_cnt, sizeof(unsigned) * 2 * 64 * kGlFuseStride));
          MXLO_HIP(hipMemsetAsync(ctx->kron_cnt, 0, sizeof(unsigned) * 2 * 64 * kGlFuseStride, ctx->stream));
          ctx->kron_cnt_n = 2 * 64 * kGlFuseStride;
        }
        bool fits = true;
        const int32_t st = dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
          GlFuse F{ctx->kron_cnt, (int)nrb, (int)gy1, (int)gx2, (int)per_xcd, ctx->tune.kron_fuse == 2 ? 0ull : fused_timeout_ticks(ctx), ctx->fault_dev, kFaultKron,
                   ctx->tune.fused_debug_drop};
#define KFUSE(AK_, TM_, WM_, WN_, BK_, NST_, PAIR_, PFD_, UNR_)                                                                \
  {                                                                                                                        \
    GlShape S1{(int)m, (int)q, (int)n, (int)nrb, (int)gy1}, S2{(int)p, (int)m, (int)q, (int)gx2, (int)nrb};               \
    if ((fits = coresident<kron_fused_kernel<T, CA, CB, B0, AK_, TM_, TM_, WM_, WN_, BK_, NST_, PAIR_, PFD_, UNR_>, WM_ * WN_ * 64>(ctx, grid))) \
      hipLaunchKernelGGL((kron_fused_kernel<T, CA, CB, B0, AK_, TM_, TM_, WM_, WN_, BK_, NST_, PAIR_, PFD_, UNR_>), dim3((unsigned)grid), \
                         dim3(WM_ * WN_ * 64), 0, ctx->stream, res, p, B, ldb, work, m, A, lda, x, q, S1, S2, (CA)alpha, (CB)beta, F); \
  }
#define KFUSE_BY_A(TM_, WM_, WN_, BK_, NST_, PAIR_, PFD_, UNR_AM_)                                                         \
  if (trans_a) KFUSE(true, TM_, WM_, WN_, BK_, NST_, PAIR_, PFD_, true) else KFUSE(false, TM_, WM_, WN_, BK_, NST_, PAIR_, PFD_, UNR_AM_)
          if constexpr (sizeof(T) == 8) {
            if (tile == 64) KFUSE_BY_A(64, 4, 2, 32, 3, true, 1, true)
            else KFUSE_BY_A(32, 2, 2, 32, 4, false, 2, true)
          } else {
            if (tile == 64) KFUSE_BY_A(64, 4, 2, 64, 3, true, 1, false)
            else KFUSE_BY_A(32, 2, 2, 64, 4, false, 2, true)
          }
#undef KFUSE_BY_A
#undef KFUSE
          if (!fits) return MXLO_OK;
          MXLO_LAUNCH_CHECK();
          return MXLO_OK;
        });
        if (st != MXLO_OK || fits) return st;
      }
      if (fst != MXLO_OK) return fst;
    }
  }
  MXLO_TRY(gemm<T>(ctx, work, m, A, lda, trans_a, x, q, true, m, q, n, 1.0, 0.0, 0));
  return gemm<T>(ctx, res, p, B, ldb, trans_b, work, m, true, p, m, q, alpha, beta, flags);
}

}  // namespace

// kron on Complex{R} with the factors and x split into real / imaginary PLANES: every complex product is four (two for a
// real factor) REAL GEMMs on the MFMA kernel above — Ut = opA*X^T, R = opB*Ut^T with
//   (Pr + i Pi)(Qr + i Qi) = (Pr Qr - Pi Qi) + i (Pr Qi + Pi Qr),
// the second GEMM of each pair accumulating through beta = 1. sign_ai / sign_bi = -1 conjugates a factor (ctprod!).
// Ai / Bi may be NULL (real factor next to complex data: kron(Float64 A, ComplexF64 B), test/test_kron.jl:3-8).
namespace mxlo {
template <typename R>
int32_t kron_planes(mxlo_ctx *ctx, R *rr, R *ri, const R *Ar, const R *Ai, int64_t am, int64_t an, int64_t lda,
                    bool trans_a, double sign_ai, const R *Br, const R *Bi, int64_t bp, int64_t bq, int64_t ldb,
                    bool trans_b, double sign_bi, const R *xr, const R *xi, R *utr, R *uti) {
  const int64_t m = trans_a ? an : am, n = trans_a ? am : an;
  const int64_t p = trans_b ? bq : bp, q = trans_b ? bp : bq;
  MXLO_TRY(gemm<R>(ctx, utr, m, Ar, lda, trans_a, xr, q, true, m, q, n, 1.0, 0.0, 0));
  MXLO_TRY(gemm<R>(ctx, uti, m, Ar, lda, trans_a, xi, q, true, m, q, n, 1.0, 0.0, 0));
  if (Ai) {
    MXLO_TRY(gemm<R>(ctx, utr, m, Ai, lda, trans_a, xi, q, true, m, q, n, -sign_ai, 1.0, 0));
    MXLO_TRY(gemm<R>(ctx, uti, m, Ai, lda, trans_a, xr, q, true, m, q, n, sign_ai, 1.0, 0));
  }
  MXLO_TRY(gemm<R>(ctx, rr, p, Br, ldb, trans_b, utr, m, true, p, m, q, 1.0, 0.0, 0));
  MXLO_TRY(gemm<R>(ctx, ri, p, Br, ldb, trans_b, uti, m, true, p, m, q, 1.0, 0.0, 0));
  if (Bi) {
    MXLO_TRY(gemm<R>(ctx, rr, p, Bi, ldb, trans_b, uti, m, true, p, m, q, -sign_bi, 1.0, 0));
    MXLO_TRY(gemm<R>(ctx, ri, p, Bi, ldb, trans_b, utr, m, true, p, m, q, sign_bi, 1.0, 0));
  }
  return MXLO_OK;
}
template int32_t kron_planes<double>(mxlo_ctx *, double *, double *, const double *, const double *, int64_t, int64_t, int64_t,
                                     bool, double, const double *, const double *, int64_t, int64_t, int64_t, bool, double,
                                     const double *, const double *, double *, double *);
template int32_t kron_planes<float>(mxlo_ctx *, float *, float *, const float *, const float *, int64_t, int64_t, int64_t, bool,
                                    double, const float *, const float *, int64_t, int64_t, int64_t, bool, double,
                                    const float *, const float *, float *, float *);
}  // namespace mxlo

// ---- kron on Complex{R}, Gauss / Karatsuba form: THREE real GEMMs per complex product instead of four ------------------
//   (a + ib)(c + id):  k1 = (a + b) c,  k2 = a (d - c),  k3 = b (c + d);  real = k1 - k3,  imag = k1 + k2
// with a, b the planes of a factor and c, d the planes of the data. The factor sum a + b (sign of b flipped for a
// conjugated factor) is formed per call in the workspace (one elementwise pass over the factor: the library is
// stateless about factors), d - c and c + d come out of the kernel that produces c, d anyway (the split of x for the
// first product, the combination of k1..k3 for the second), and the three GEMMs write three separate planes with
// beta = 0 (no read-modify-write of C). A real factor keeps its two plain GEMMs.
namespace mxlo {
template <typename R>
__global__ void __launch_bounds__(kBlock)
plane_sum_kernel(R *__restrict__ out, const R *__restrict__ a, const R *__restrict__ b, int64_t rows, int64_t cols,
                 int64_t ld, R sign) {   // out (rows x cols, ld = rows) = a + sign*b, a / b with leading dimension ld
  using V = typename Vec16<R>::type;
  constexpr int VEC = Vec16<R>::N;
  const int64_t total = rows * cols;
  if (ld == rows && ((((uintptr_t)a) | ((uintptr_t)b)) & 15u) == 0) {     // contiguous factor: 16-byte accesses
    const int64_t nv = total / VEC;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kBlock) {
      const V av = reinterpret_cast<const V *>(a)[i], bv = reinterpret_cast<const V *>(b)[i];
      V o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = av[e] + sign * bv[e];
      reinterpret_cast<V *>(out)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < total - nv * VEC) {
      const int64_t i = nv * VEC + threadIdx.x;
      out[i] = a[i] + sign * b[i];
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i % rows, c = i / rows;
    out[i] = a[r + c * ld] + sign * b[r + c * ld];
  }
}
// (k1, k2, k3) -> re = k1 - k3, im = k1 + k2 and, for the next Gauss product, im - re and re + im. K3 false: the factor
// was real and (re, im) hold the two products already: only the two combinations are formed. All planes are 16-byte
// aligned workspace planes: 16-byte accesses, scalar tail.
template <typename R, bool K3, bool NEXT>
__global__ void __launch_bounds__(kBlock)
gauss_planes_kernel(R *__restrict__ re, R *__restrict__ im, R *__restrict__ dmr, R *__restrict__ spr,
                    const R *__restrict__ k1, const R *__restrict__ k2, const R *__restrict__ k3, int64_t n) {
  using V = typename Vec16<R>::type;
  constexpr int VEC = Vec16<R>::N;
  const int64_t nv = n / VEC;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kBlock) {
    V r_, i_;
    if constexpr (K3) {
      const V a = reinterpret_cast<const V *>(k1)[i], b = reinterpret_cast<const V *>(k2)[i], c = reinterpret_cast<const V *>(k3)[i];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        r_[e] = a[e] - c[e];
        i_[e] = a[e] + b[e];
      }
      reinterpret_cast<V *>(re)[i] = r_;
      reinterpret_cast<V *>(im)[i] = i_;
    } else {
      r_ = reinterpret_cast<const V *>(re)[i];
      i_ = reinterpret_cast<const V *>(im)[i];
    }
    if constexpr (NEXT) {
      V d, s;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        d[e] = i_[e] - r_[e];
        s[e] = r_[e] + i_[e];
      }
      reinterpret_cast<V *>(dmr)[i] = d;
      reinterpret_cast<V *>(spr)[i] = s;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < n - nv * VEC) {
    const int64_t i = nv * VEC + threadIdx.x;
    R r_, i_;
    if constexpr (K3) {
      r_ = k1[i] - k3[i];
      i_ = k1[i] + k2[i];
      re[i] = r_;
      im[i] = i_;
    } else {
      r_ = re[i];
      i_ = im[i];
    }
    if constexpr (NEXT) {
      dmr[i] = i_ - r_;
      spr[i] = r_ + i_;
    }
  }
}

// work layout (R scalars, every plane 16-byte aligned): see mxlo_kron_mul_c3 in include/mxlo.h
// As / Bs: optional factor-sum planes a + sign*b (contiguous, leading dimension = rows) cached by the caller per factor
// state; NULL: formed here, per call. kout: when non-NULL and B is complex, the three products of the SECOND stage are
// left in kout[0..2] (re = k1 - k3, im = k1 + k2 is then the caller's — it is folded into the join pass).
template <typename R>
int32_t kron_planes3(mxlo_ctx *ctx, R *rr, R *ri, const R *Ar, const R *Ai, const R *As, int64_t am, int64_t an, int64_t lda,
                     bool trans_a, double sign_ai, const R *Br, const R *Bi, const R *Bs, int64_t bp, int64_t bq, int64_t ldb,
                     bool trans_b, double sign_bi, const R *xr, const R *xi, const R *xd, const R *xs, R *work,
                     const R **kout) {
  const int64_t m = trans_a ? an : am, n = trans_a ? am : an;
  const int64_t p = trans_b ? bq : bp, q = trans_b ? bp : bq;
  auto pad = [](int64_t k) { return (k + 3) & ~(int64_t)3; };
  R *ur = work, *ui = ur + pad(m * q), *ud = ui + pad(m * q), *us = ud + pad(m * q);
  R *k1 = us + pad(m * q), *k2 = k1 + pad(std::max(m * q, p * m)), *k3 = k2 + pad(std::max(m * q, p * m));
  R *fs = k3 + pad(std::max(m * q, p * m));                       // factor sum plane (am*an, then bp*bq)
  auto egrid = [&](int64_t cnt) { return grid_for(ctx, (cnt + 1) / 2, kBlock, 8); };
  // ---- first product: U^T = opA(A) X^T  (m x q, K = n)
  if (Ai) {
    if (!As) {
      hipLaunchKernelGGL((plane_sum_kernel<R>), dim3(egrid(am * an)), dim3(kBlock), 0, ctx->stream, fs, Ar, Ai, am, an, lda,
                         (R)sign_ai);
      MXLO_LAUNCH_CHECK();
      As = fs;
    }
    MXLO_TRY(gemm<R>(ctx, k1, m, As, am, trans_a, xr, q, true, m, q, n, 1.0, 0.0, 0));          // (a + b) c
    MXLO_TRY(gemm<R>(ctx, k2, m, Ar, lda, trans_a, xd, q, true, m, q, n, 1.0, 0.0, 0));         // a (d - c)
    MXLO_TRY(gemm<R>(ctx, k3, m, Ai, lda, trans_a, xs, q, true, m, q, n, sign_ai, 0.0, 0));     // b (c + d)
    if (Bi)
      hipLaunchKernelGGL((gauss_planes_kernel<R, true, true>), dim3(egrid(m * q)), dim3(kBlock), 0, ctx->stream, ur, ui, ud, us,
                         (const R *)k1, (const R *)k2, (const R *)k3, m * q);
    else
      hipLaunchKernelGGL((gauss_planes_kernel<R, true, false>), dim3(egrid(m * q)), dim3(kBlock), 0, ctx->stream, ur, ui,
                         (R *)nullptr, (R *)nullptr, (const R *)k1, (const R *)k2, (const R *)k3, m * q);
    MXLO_LAUNCH_CHECK();
  } else {
    MXLO_TRY(gemm<R>(ctx, ur, m, Ar, lda, trans_a, xr, q, true, m, q, n, 1.0, 0.0, 0));
    MXLO_TRY(gemm<R>(ctx, ui, m, Ar, lda, trans_a, xi, q, true, m, q, n, 1.0, 0.0, 0));
    if (Bi) {
      hipLaunchKernelGGL((gauss_planes_kernel<R, false, true>), dim3(egrid(m * q)), dim3(kBlock), 0, ctx->stream, ur, ui, ud, us,
                         (const R *)nullptr, (const R *)nullptr, (const R *)nullptr, m * q);
      MXLO_LAUNCH_CHECK();
    }
  }
  // ---- second product: R = opB(B) U  (p x m, K = q)
  if (Bi) {
    if (!Bs) {
      hipLaunchKernelGGL((plane_sum_kernel<R>), dim3(egrid(bp * bq)), dim3(kBlock), 0, ctx->stream, fs, Br, Bi, bp, bq, ldb,
                         (R)sign_bi);
      MXLO_LAUNCH_CHECK();
      Bs = fs;
    }
    MXLO_TRY(gemm<R>(ctx, k1, p, Bs, bp, trans_b, ur, m, true, p, m, q, 1.0, 0.0, 0));
    MXLO_TRY(gemm<R>(ctx, k2, p, Br, ldb, trans_b, ud, m, true, p, m, q, 1.0, 0.0, 0));
    MXLO_TRY(gemm<R>(ctx, k3, p, Bi, ldb, trans_b, us, m, true, p, m, q, sign_bi, 0.0, 0));
    if (kout) {                       // the caller's join pass forms re = k1 - k3, im = k1 + k2 on the fly
      kout[0] = k1;
      kout[1] = k2;
      kout[2] = k3;
    } else {
      hipLaunchKernelGGL((gauss_planes_kernel<R, true, false>), dim3(egrid(p * m)), dim3(kBlock), 0, ctx->stream, rr, ri,
                         (R *)nullptr, (R *)nullptr, (const R *)k1, (const R *)k2, (const R *)k3, p * m);
      MXLO_LAUNCH_CHECK();
    }
  } else {
    if (kout) kout[0] = kout[1] = kout[2] = nullptr;
    MXLO_TRY(gemm<R>(ctx, rr, p, Br, ldb, trans_b, ur, m, true, p, m, q, 1.0, 0.0, 0));
    MXLO_TRY(gemm<R>(ctx, ri, p, Br, ldb, trans_b, ui, m, true, p, m, q, 1.0, 0.0, 0));
  }
  return MXLO_OK;
}
template int32_t kron_planes3<double>(mxlo_ctx *, double *, double *, const double *, const double *, const double *, int64_t,
                                      int64_t, int64_t, bool, double, const double *, const double *, const double *, int64_t,
                                      int64_t, int64_t, bool, double, const double *, const double *, const double *,
                                      const double *, double *, const double **);
template int32_t kron_planes3<float>(mxlo_ctx *, float *, float *, const float *, const float *, const float *, int64_t, int64_t,
                                     int64_t, bool, double, const float *, const float *, const float *, int64_t, int64_t,
                                     int64_t, bool, double, const float *, const float *, const float *, const float *, float *,
                                     const float **);
// out (rows x cols, contiguous) = a + sign*b: the factor-sum plane of the Gauss form, for callers that cache it
template <typename R>
int32_t plane_sum(mxlo_ctx *ctx, R *out, const R *a, const R *b, int64_t rows, int64_t cols, int64_t ld, double sign) {
  if (rows * cols <= 0) return MXLO_OK;
  hipLaunchKernelGGL((plane_sum_kernel<R>), dim3(grid_for(ctx, (rows * cols + 1) / 2, kBlock, 8)), dim3(kBlock), 0, ctx->stream,
                     out, a, b, rows, cols, ld, (R)sign);
  MXLO_LAUNCH_CHECK();
  return MXLO_OK;
}
template int32_t plane_sum<double>(mxlo_ctx *, double *, const double *, const double *, int64_t, int64_t, int64_t, double);
template int32_t plane_sum<float>(mxlo_ctx *, float *, const float *, const float *, int64_t, int64_t, int64_t, double);
}  // namespace mxlo

static inline void eff_ab(int32_t dtype, int32_t flags, double &alpha, double &beta) {
  eff_scalars(dtype == MXLO_F64 ? 8 : 4, flags, alpha, beta);
}

MXLO_API int32_t mxlo_gemv(mxlo_ctx *ctx, int32_t dtype, void *res, const void *M, int64_t m,
                           int64_t n, int64_t ld, const void *v, double alpha, double beta,
                           int32_t op_mode, int32_t flags) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_gemv: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(dtype == MXLO_F64 || dtype == MXLO_F32, MXLO_EINVAL, "bad dtype");
  MXLO_REQUIRE(m >= 0 && n >= 0 && ld >= (m > 1 ? m : 1), MXLO_ESHAPE, "mxlo_gemv: bad shape");
  MXLO_REQUIRE(op_mode >= MXLO_OP_N && op_mode <= MXLO_OP_C, MXLO_EINVAL, "bad op_mode");
  eff_ab(dtype, flags, alpha, beta);
  if (dtype == MXLO_F64)
    return gemv_any<double>(ctx, (double *)res, (const double *)M, m, n, ld, (const double *)v, alpha, beta, op_mode, flags);
  return gemv_any<float>(ctx, (float *)res, (const float *)M, m, n, ld, (const float *)v, alpha, beta, op_mode, flags);
}

MXLO_API int32_t mxlo_gemv_block(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t ldr, const void *M, int64_t m, int64_t n,
                                 int64_t ld, const void *V, int64_t ldv, int64_t k, double alpha, double beta,
                                 int32_t op_mode, int32_t flags) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_gemv_block: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(dtype == MXLO_F64 || dtype == MXLO_F32, MXLO_EINVAL, "bad dtype");
  MXLO_REQUIRE(op_mode >= MXLO_OP_N && op_mode <= MXLO_OP_C, MXLO_EINVAL, "bad op_mode");
  const int64_t nres = op_mode == MXLO_OP_N ? m : n, nin = op_mode == MXLO_OP_N ? n : m;
  MXLO_REQUIRE(m >= 0 && n >= 0 && k >= 0 && ld >= (m > 1 ? m : 1) && ldr >= (nres > 1 ? nres : 1) &&
                   ldv >= (nin > 1 ? nin : 1), MXLO_ESHAPE, "mxlo_gemv_block: bad shape");
  MXLO_REQUIRE(nres == 0 || k == 0 || (res && (nin == 0 || (M && V))), MXLO_EINVAL, "mxlo_gemv_block: NULL operand");
  eff_ab(dtype, flags, alpha, beta);
  if (dtype == MXLO_F64)
    return gemv_block<double>(ctx, (double *)res, ldr, (const double *)M, m, n, ld, (const double *)V, ldv, k, alpha,
                              beta, op_mode == MXLO_OP_N ? MXLO_OP_N : MXLO_OP_T, flags);
  return gemv_block<float>(ctx, (float *)res, ldr, (const float *)M, m, n, ld, (const float *)V, ldv, k, alpha, beta,
                           op_mode == MXLO_OP_N ? MXLO_OP_N : MXLO_OP_T, flags);
}

MXLO_API int32_t mxlo_hermitian_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *d,
                                    const void *A, int64_t lda, const void *v, int64_t n,
                                    double alpha, double beta, int32_t flags) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_hermitian_mul: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(dtype == MXLO_F64 || dtype == MXLO_F32, MXLO_EINVAL, "bad dtype");
  MXLO_REQUIRE(n >= 0 && lda >= (n > 1 ? n : 1), MXLO_ESHAPE, "mxlo_hermitian_mul: bad shape");
  MXLO_REQUIRE(n == 0 || (res && d && A && v), MXLO_EINVAL, "mxlo_hermitian_mul: NULL operand");
  eff_ab(dtype, flags, alpha, beta);
  if (dtype == MXLO_F64)
    return hermitian_t<double>(ctx, (double *)res, (const double *)d, (const double *)A, lda, (const double *)v, n, alpha, beta, flags);
  return hermitian_t<float>(ctx, (float *)res, (const float *)d, (const float *)A, lda, (const float *)v, n, alpha, beta, flags);
}

MXLO_API int32_t mxlo_hermitian_mul_block(mxlo_ctx *ctx, int32_t dtype, void *res, int64_t ldr, const void *d, const void *A, int64_t lda,
                                          const void *V, int64_t ldv, int64_t n, int64_t k, double alpha, double beta, int32_t flags) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_hermitian_mul_block: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(n >= 0 && k >= 0, MXLO_ESHAPE, "mxlo_hermitian_mul_block: negative size");
  if (n == 0 || k == 0) return MXLO_OK;
  MXLO_REQUIRE(res && d && A && V, MXLO_EINVAL, "mxlo_hermitian_mul_block: NULL operand");
  MXLO_REQUIRE(lda >= n && ldr >= n && ldv >= n, MXLO_ESHAPE, "mxlo_hermitian_mul_block: a leading dimension is smaller than n");
  eff_ab(dtype, flags, alpha, beta);
  if (dtype == MXLO_F64)
    return hermitian_block_t<double>(ctx, (double *)res, ldr, (const double *)d, (const double *)A, lda, (const double *)V, ldv, n, k, alpha, beta, flags);
  if (dtype == MXLO_F32)
    return hermitian_block_t<float>(ctx, (float *)res, ldr, (const float *)d, (const float *)A, lda, (const float *)V, ldv, n, k, alpha, beta, flags);
  set_error("mxlo_hermitian_mul_block: dtype %d (Float64 / Float32 only; complex blocks take the column loop)", dtype);
  return MXLO_EINVAL;
}

MXLO_API int32_t mxlo_kron_mul_ex(mxlo_ctx *ctx, int32_t dtype, void *res, const void *A, int64_t am, int64_t an,
                                  int64_t lda, int32_t trans_a, const void *B, int64_t bp, int64_t bq, int64_t ldb,
                                  int32_t trans_b, const void *x, void *work, double alpha, double beta,
                                  int32_t flags) {
  MXLO_REQUIRE(ctx, MXLO_EINVAL, "mxlo_kron_mul_ex: ctx is NULL");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(dtype == MXLO_F64 || dtype == MXLO_F32, MXLO_EINVAL, "bad dtype");
  MXLO_REQUIRE(am >= 0 && an >= 0 && bp >= 0 && bq >= 0, MXLO_ESHAPE, "mxlo_kron_mul_ex: negative size");
  MXLO_REQUIRE(lda >= (am > 1 ? am : 1) && ldb >= (bp > 1 ? bp : 1), MXLO_ESHAPE, "mxlo_kron_mul_ex: bad leading dimension");
  const int64_t m = trans_a ? an : am, p = trans_b ? bq : bp;
  if (m * p == 0) return MXLO_OK;
  MXLO_REQUIRE(res && A && B && x && work, MXLO_EINVAL, "mxlo_kron_mul_ex: NULL operand");
  eff_ab(dtype, flags, alpha, beta);
  if (dtype == MXLO_F64)
    return kron_t<double>(ctx, (double *)res, (const double *)A, am, an, lda, trans_a != 0, (const double *)B, bp, bq,
                          ldb, trans_b != 0, (const double *)x, (double *)work, alpha, beta, flags);
  return kron_t<float>(ctx, (float *)res, (const float *)A, am, an, lda, trans_a != 0, (const float *)B, bp, bq, ldb,
                       trans_b != 0, (const float *)x, (float *)work, alpha, beta, flags);
}

MXLO_API int32_t mxlo_kron_mul(mxlo_ctx *ctx, int32_t dtype, void *res, const void *A, int64_t m,
                               int64_t n, int64_t lda, const void *B, int64_t p, int64_t q,
                               int64_t ldb, const void *x, void *work, double alpha, double beta,
                               int32_t op_mode, int32_t flags) {
  MXLO_REQUIRE(op_mode >= MXLO_OP_N && op_mode <= MXLO_OP_C, MXLO_EINVAL, "bad op_mode");
  const int32_t tr = op_mode == MXLO_OP_N ? 0 : 1;   // tprod!/ctprod!: both factors transposed (real dtypes)
  return mxlo_kron_mul_ex(ctx, dtype, res, A, m, n, lda, tr, B, p, q, ldb, tr, x, work, alpha, beta, flags);
}
