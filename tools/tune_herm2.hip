// tune_herm2.hip — ablation of the opHermitian 256x32 panel kernel (product kernel = variant 0).
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=off tune_herm2.hip -o tune_herm2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int HR = 256, HC = 32, kBlock = 256;
typedef double V2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void panel_of(int64_t t, int64_t &G, int64_t &J) {
  G = (int64_t)((sqrt(1.0 + (double)t) - 1.0) * 0.5);
  while (4 * G * (G + 1) > t) --G;
  while (4 * (G + 1) * (G + 2) <= t) ++G;
  J = t - 4 * G * (G + 1);
}

// column butterfly + stores shared by the variants
template <bool HAVE_W8>
__device__ __forceinline__ void reduce_store(double (&pcol)[16], double prow0, double prow1, int tid, int cg, int rp,
                                             int64_t G, int64_t J, int64_t i0, int64_t j0, int64_t n,
                                             double *Prow, double *Pcol) {
  const int lane = tid & 63, wave = tid >> 6;
  double w8[8], w4[4], w2[2], w1;
  if constexpr (HAVE_W8) {
#pragma unroll
    for (int q = 0; q < 8; ++q) w8[q] = pcol[q];
  } else { const bool hi = (lane & 32) != 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const double send = hi ? pcol[q] : pcol[8 + q]; w8[q] = (hi ? pcol[8 + q] : pcol[q]) + __shfl_xor(send, 32, 64); } }
  { const bool hi = (lane & 16) != 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const double send = hi ? w8[q] : w8[4 + q]; w4[q] = (hi ? w8[4 + q] : w8[q]) + __shfl_xor(send, 16, 64); } }
  { const bool hi = (lane & 8) != 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) { const double send = hi ? w4[q] : w4[2 + q]; w2[q] = (hi ? w4[2 + q] : w4[q]) + __shfl_xor(send, 8, 64); } }
  { const bool hi = (lane & 4) != 0; const double send = hi ? w2[0] : w2[1]; w1 = (hi ? w2[1] : w2[0]) + __shfl_xor(send, 4, 64); }
  w1 += __shfl_xor(w1, 2, 64);
  w1 += __shfl_xor(w1, 1, 64);
  __shared__ double colred[4][16];
  __shared__ double rowred[2][HR];
  if ((lane & 3) == 0) {
    const int k = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    colred[wave][k] = w1;
  }
  rowred[cg][2 * rp] = prow0;
  rowred[cg][2 * rp + 1] = prow1;
  __syncthreads();
  if (tid < HC) {
    const int c_cg = tid & 1, k = tid >> 1;
    const int64_t gc = j0 + tid;
    if (gc < n) Pcol[G * n + gc] = colred[2 * c_cg][k] + colred[2 * c_cg + 1][k];
  }
  { const int64_t row = i0 + tid; if (row < n) Prow[J * n + row] = rowred[0][tid] + rowred[1][tid]; }
}

// VARIANT 0: the product kernel (masked loads everywhere). LEVEL 0 = loads only.
template <int VARIANT, int LEVEL>
__global__ void __launch_bounds__(kBlock)
herm_panel(const double *__restrict__ A, int64_t lda, const double *__restrict__ v, int64_t n,
           double *__restrict__ Prow, double *__restrict__ Pcol, double *sink) {
  int64_t G, J;
  panel_of(blockIdx.x, G, J);
  const int64_t i0 = G * HR, j0 = J * HC;
  if (j0 >= n) return;
  const int tid = threadIdx.x;
  const int rp = tid & 127;
  const int cg = VARIANT >= 1 ? __builtin_amdgcn_readfirstlane(tid >> 7) : (tid >> 7);
  const int64_t gr = i0 + 2 * rp;
  double e0[16], e1[16];
  const bool interior = VARIANT >= 1 && (j0 + HC <= i0) && (i0 + HR <= n);
  if (VARIANT >= 2 && !interior) return;                 // edge panels go to a second (masked) launch
  if (VARIANT >= 2 || interior) {
    const double *base = A + (j0 + cg) * lda;           // wave-uniform
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const V2 x = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(base + (int64_t)(2 * k) * lda + gr));
      e0[k] = x[0];
      e1[k] = x[1];
    }
  } else if constexpr (VARIANT < 2) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int64_t gc = j0 + cg + 2 * k;
      e0[k] = 0; e1[k] = 0;
      if (gc < n && gr + 1 > gc) {
        const double *p = A + gr + gc * lda;
        if (gr + 1 < n) { const V2 x = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(p)); e0[k] = x[0]; e1[k] = x[1]; }
        else if (gr < n) e0[k] = p[0];
        if (!(gr > gc)) e0[k] = 0;
      }
    }
  }
  if constexpr (LEVEL == 0) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += e0[k] + e1[k];
    if (s == 1.2345e300) sink[0] = s;
    return;
  }
  const double vr0 = gr < n ? v[gr] : 0.0, vr1 = gr + 1 < n ? v[gr + 1] : 0.0;
  double prow0 = 0.0, prow1 = 0.0, pcol[16];
  if constexpr (VARIANT >= 3) {
    const bool hi = (tid & 32) != 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double vca = v[j0 + cg + 2 * q], vcb = v[j0 + cg + 2 * (q + 8)];
      prow0 = fma(e0[q], vca, prow0);
      prow1 = fma(e1[q], vca, prow1);
      prow0 = fma(e0[q + 8], vcb, prow0);
      prow1 = fma(e1[q + 8], vcb, prow1);
      const double pa = fma(e1[q], vr1, e0[q] * vr0), pb = fma(e1[q + 8], vr1, e0[q + 8] * vr0);
      pcol[q] = (hi ? pb : pa) + __shfl_xor(hi ? pa : pb, 32, 64);
      if constexpr (VARIANT == 4) __builtin_amdgcn_sched_barrier(0);
    }
    reduce_store<true>(pcol, prow0, prow1, tid, cg, rp, G, J, i0, j0, n, Prow, Pcol);
    return;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t gc = j0 + cg + 2 * k;
    const double vc = gc < n ? v[gc] : 0.0;
    prow0 = fma(e0[k], vc, prow0);
    prow1 = fma(e1[k], vc, prow1);
    pcol[k] = fma(e1[k], vr1, e0[k] * vr0);
  }
  if constexpr (LEVEL == 1) {
    double s = prow0 + prow1;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += pcol[k];
    if (s == 1.2345e300) sink[0] = s;
    return;
  }
  reduce_store<false>(pcol, prow0, prow1, tid, cg, rp, G, J, i0, j0, n, Prow, Pcol);
}

template <int VARIANT, int LEVEL>
void run(const double *A, const double *v, int64_t n, double *Prow, double *Pcol, double *sink) {
  const int64_t ng = (n + HR - 1) / HR, grid = 4 * ng * (ng + 1);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((herm_panel<VARIANT, LEVEL>), dim3(grid), dim3(kBlock), 0, 0, A, n, v, n, Prow, Pcol, sink);
  CK(hipEventRecord(e0, 0));
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((herm_panel<VARIANT, LEVEL>), dim3(grid), dim3(kBlock), 0, 0, A, n, v, n, Prow, Pcol, sink);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("variant %d level %d: %.1f us  %.0f GB/s (of 4n^2 = %.2f GB)\n", VARIANT, LEVEL, ms * 1e3, 4.0 * n * n / ms / 1e6, 4.0 * n * n / 1e9);
}

// ---- strip kernel: one workgroup owns row group G and C consecutive column tiles; prow accumulates across
// the tiles in registers (ONE LDS/barrier per strip), column partials are stored per wave-half (no barrier).
template <int C, bool PREFETCH>
__global__ void __launch_bounds__(kBlock)
herm_strip(const double *__restrict__ A, int64_t lda, const double *__restrict__ v, int64_t n,
           double *__restrict__ Prow, double *__restrict__ Pcol2) {
  constexpr int Q = 8 / C;
  const int64_t t = blockIdx.x, u = t / Q;
  int64_t G = (int64_t)((sqrt(8.0 * (double)u + 1.0) - 1.0) * 0.5);
  while (G * (G + 1) / 2 > u) --G;
  while ((G + 1) * (G + 2) / 2 <= u) ++G;
  const int64_t sidx = (u - G * (G + 1) / 2) * Q + t % Q;
  const int64_t i0 = G * HR;
  const int tid = threadIdx.x, lane = tid & 63, rp = tid & 127;
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int half = __builtin_amdgcn_readfirstlane((tid >> 6) & 1);
  const int64_t gr = i0 + 2 * rp;
  const double vr0 = v[gr], vr1 = v[gr + 1];
  double prow0 = 0.0, prow1 = 0.0;
  V2 cur[16], nxt[16];
  auto load = [&](V2 (&e)[16], int64_t j0) {
    const double *base = A + (j0 + cg) * lda + gr;
#pragma unroll
    for (int k = 0; k < 16; ++k) e[k] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(base + (int64_t)(2 * k) * lda));
  };
  auto process = [&](V2 (&e)[16], int64_t j0) {
    const bool hi = (lane & 32) != 0;
    double w8[8], w4[4], w2[2], w1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double vca = v[j0 + cg + 2 * q], vcb = v[j0 + cg + 2 * (q + 8)];
      prow0 = fma(e[q][0], vca, prow0);
      prow1 = fma(e[q][1], vca, prow1);
      prow0 = fma(e[q + 8][0], vcb, prow0);
      prow1 = fma(e[q + 8][1], vcb, prow1);
      const double pa = fma(e[q][1], vr1, e[q][0] * vr0), pb = fma(e[q + 8][1], vr1, e[q + 8][0] * vr0);
      w8[q] = (hi ? pb : pa) + __shfl_xor(hi ? pa : pb, 32, 64);
      __builtin_amdgcn_sched_barrier(0);
    }
    { const bool h2 = (lane & 16) != 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const double send = h2 ? w8[q] : w8[4 + q]; w4[q] = (h2 ? w8[4 + q] : w8[q]) + __shfl_xor(send, 16, 64); } }
    { const bool h2 = (lane & 8) != 0;
#pragma unroll
      for (int q = 0; q < 2; ++q) { const double send = h2 ? w4[q] : w4[2 + q]; w2[q] = (h2 ? w4[2 + q] : w4[q]) + __shfl_xor(send, 8, 64); } }
    { const bool h2 = (lane & 4) != 0; const double send = h2 ? w2[0] : w2[1]; w1 = (h2 ? w2[1] : w2[0]) + __shfl_xor(send, 4, 64); }
    w1 += __shfl_xor(w1, 2, 64);
    w1 += __shfl_xor(w1, 1, 64);
    if ((lane & 3) == 0) {
      const int k = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
      Pcol2[(2 * G + half) * n + j0 + cg + 2 * k] = w1;
    }
  };
  // harness: interior tiles only (j0 + 32 <= i0)
  int64_t jt = 0;
  auto j0_of = [&](int64_t x) { return (sidx * C + x) * HC; };
  if constexpr (PREFETCH) {
    if (j0_of(0) + HC <= i0) load(cur, j0_of(0));
    for (; jt < C; jt += 2) {
      if (j0_of(jt) + HC > i0) break;
      const bool has1 = jt + 1 < C && j0_of(jt + 1) + HC <= i0;
      if (has1) load(nxt, j0_of(jt + 1));
      process(cur, j0_of(jt));
      if (!has1) break;
      const bool has2 = jt + 2 < C && j0_of(jt + 2) + HC <= i0;
      if (has2) load(cur, j0_of(jt + 2));
      process(nxt, j0_of(jt + 1));
    }
  } else {
    for (; jt < C; ++jt) {
      if (j0_of(jt) + HC > i0) break;
      load(cur, j0_of(jt));
      process(cur, j0_of(jt));
    }
  }
  __shared__ double rowred[2][HR];
  rowred[cg][2 * rp] = prow0;
  rowred[cg][2 * rp + 1] = prow1;
  __syncthreads();
  Prow[sidx * n + i0 + tid] = rowred[0][tid] + rowred[1][tid];
}

template <int C, bool PREFETCH>
void run_strip(const double *A, const double *v, int64_t n, double *Prow, double *Pcol) {
  const int64_t ng = (n + HR - 1) / HR, grid = (8 / C) * ng * (ng + 1) / 2;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((herm_strip<C, PREFETCH>), dim3(grid), dim3(kBlock), 0, 0, A, n, v, n, Prow, Pcol);
  CK(hipEventRecord(e0, 0));
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((herm_strip<C, PREFETCH>), dim3(grid), dim3(kBlock), 0, 0, A, n, v, n, Prow, Pcol);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("strip C=%d prefetch=%d: %.1f us  %.0f GB/s\n", C, (int)PREFETCH, ms * 1e3, 4.0 * n * n / ms / 1e6);
}

int main() {
  const int64_t n = 16384, ng = n / HR, nbc = n / HC;
  double *A, *v, *Prow, *Pcol, *sink;
  CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&Prow, nbc * n * 8)); CK(hipMalloc(&Pcol, 2 * ng * n * 8)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(A, 1, n * n * 8)); CK(hipMemset(v, 1, n * 8));
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>(A, v, n, Prow, Pcol, sink); run<0, 1>(A, v, n, Prow, Pcol, sink); run<0, 2>(A, v, n, Prow, Pcol, sink);
    run<1, 0>(A, v, n, Prow, Pcol, sink); run<1, 1>(A, v, n, Prow, Pcol, sink); run<1, 2>(A, v, n, Prow, Pcol, sink);
    run<2, 0>(A, v, n, Prow, Pcol, sink); run<2, 1>(A, v, n, Prow, Pcol, sink); run<2, 2>(A, v, n, Prow, Pcol, sink);
    run<3, 2>(A, v, n, Prow, Pcol, sink); run<4, 2>(A, v, n, Prow, Pcol, sink);
    run_strip<8, false>(A, v, n, Prow, Pcol); run_strip<4, false>(A, v, n, Prow, Pcol); run_strip<2, false>(A, v, n, Prow, Pcol); run_strip<1, false>(A, v, n, Prow, Pcol);
    run_strip<8, true>(A, v, n, Prow, Pcol); run_strip<4, true>(A, v, n, Prow, Pcol); run_strip<2, true>(A, v, n, Prow, Pcol);
  }
  return 0;
}
