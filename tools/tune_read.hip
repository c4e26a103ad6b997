// tools/tune_read.hip — experiment harness (NOT the library): what a READ-ONLY pass over an HBM-sized buffer can reach on one
// MI355X, by access shape. Motivation: the row-band GEMV (dense.hip: gemv_n_rows_kernel) streams a 2 GB matrix at 7.0 TB/s
// (0.88 of peak), above the 6.3 TB/s the 1-D read sweep of tools/tune_persist.hip (A1) measured. Which ingredient is it?
//   1-D : grid-stride over 16-byte vectors; block 256 / 512, UNR loads in flight per lane, nontemporal or default loads,
//         persistent grids of 1 .. 8 workgroups per CU or one workgroup per chunk;
//   2-D : the row-band shape — a workgroup owns RB rows of a column-major m x n matrix, LPR lanes cover the band's piece of a
//         column, the other lanes take other columns.
//   K streams: the same bytes as K separate arrays read in lockstep (the shape of a dots pass over K panel columns).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/tune_read tools/tune_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BLOCK, int UNR, bool NT>
__global__ void __launch_bounds__(BLOCK) read1d(const f64x2 *__restrict__ p, int64_t nvec, double *out) {
  double s = 0;
  const int64_t stride = (int64_t)gridDim.x * BLOCK * UNR;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK * UNR + threadIdx.x; i + (int64_t)(UNR - 1) * BLOCK < nvec; i += stride) {
    f64x2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * BLOCK) : p[i + u * BLOCK];
#pragma unroll
    for (int u = 0; u < UNR; ++u) s += v[u][0] + v[u][1];
  }
  if (s == 1.2345e300) out[blockIdx.x] = s;
}

template <int RB, int UNR, bool NT>
__global__ void __launch_bounds__(512) read2d(const double *__restrict__ M, int64_t m, int64_t n, double *out) {
  constexpr int LPR = RB / 2, NCL = 512 / LPR;
  const int seg = threadIdx.x % LPR, cl = threadIdx.x / LPR;
  const double *base = M + (int64_t)blockIdx.x * RB + seg * 2;
  double s = 0;
  for (int64_t j = cl; j + (int64_t)(UNR - 1) * NCL < n; j += (int64_t)UNR * NCL) {
    f64x2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const f64x2 *q = reinterpret_cast<const f64x2 *>(base + (j + (int64_t)u * NCL) * m);
      v[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) s += v[u][0] + v[u][1];
  }
  if (s == 1.2345e300) out[blockIdx.x] = s;
}

struct Ptrs { const f64x2 *p[40]; };
template <int K, int UNR, bool NT>
__global__ void __launch_bounds__(256) readk(Ptrs P, int64_t nvec_each, double *out) {
  double s = 0;
  constexpr int PER = UNR / K > 0 ? UNR / K : 1;       // loads per stream per step
  const int64_t stride = (int64_t)gridDim.x * 256 * PER;
  for (int64_t i = (int64_t)blockIdx.x * 256 * PER + threadIdx.x; i + (int64_t)(PER - 1) * 256 < nvec_each; i += stride) {
    f64x2 v[K][PER];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int u = 0; u < PER; ++u) v[k][u] = NT ? __builtin_nontemporal_load(P.p[k] + i + u * 256) : P.p[k][i + u * 256];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int u = 0; u < PER; ++u) s += v[k][u][0] + v[k][u][1];
  }
  if (s == 1.2345e300) out[blockIdx.x] = s;
}


// K read streams + ONE write stream (the shape of a combine pass: res = f(x, K panel columns)); one vector per thread
template <int K, bool NT, int BATCH>
__global__ void __launch_bounds__(256) readk_w(Ptrs P, f64x2 *__restrict__ res, int64_t nvec_each) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec_each) return;
  f64x2 q = {0, 0};
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += BATCH) {
    f64x2 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) v[u] = NT ? __builtin_nontemporal_load(P.p[k0 + u] + i) : P.p[k0 + u][i];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) { q[0] += 1.5 * v[u][0]; q[1] += 1.5 * v[u][1]; }
  }
  if (NT) __builtin_nontemporal_store(q, res + i); else res[i] = q;
}

// K read streams + TWO write streams (round 6: the shape of a push! pass — the panel columns are read for the Gram rows while the new
// pair is stored into its two slots): one vector per thread; the two stored vectors are two of the K read ones (s, y)
template <int K, bool NT, int BATCH>
__global__ void __launch_bounds__(256) readk_w2(Ptrs P, f64x2 *__restrict__ w1, f64x2 *__restrict__ w2, int64_t nvec_each, double *out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec_each) return;
  f64x2 q = {0, 0}, s0, s1;
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += BATCH) {
    f64x2 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) v[u] = NT ? __builtin_nontemporal_load(P.p[k0 + u] + i) : P.p[k0 + u][i];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) { q[0] += 1.5 * v[u][0]; q[1] += 1.5 * v[u][1]; }
    if (k0 == 0) { s0 = v[0]; s1 = v[1]; }
  }
  __builtin_nontemporal_store(s0, w1 + i);
  __builtin_nontemporal_store(s1, w2 + i);
  if (q[0] == 1.2345e300) out[0] = q[1];
}

// the library's streaming-map shape (stream_kernels.h: map_kernel): 2 reads + 1 write, UNROLL vectors per thread spaced by
// the block size, one chunk per workgroup (blockIdx -> chunk), nontemporal loads and stores
template <int UNROLL, bool NTL, bool NTS, int BLOCK>
__global__ void __launch_bounds__(BLOCK) map2r1w(f64x2 *__restrict__ r, const f64x2 *__restrict__ a, const f64x2 *__restrict__ b, int64_t nvec, double c) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * UNROLL + threadIdx.x;
  f64x2 x[UNROLL], y[UNROLL];
  if (base + (int64_t)(UNROLL - 1) * BLOCK < nvec) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      x[u] = NTL ? __builtin_nontemporal_load(a + base + u * BLOCK) : a[base + u * BLOCK];
      y[u] = NTL ? __builtin_nontemporal_load(b + base + u * BLOCK) : b[base + u * BLOCK];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      f64x2 o = {x[u][0] - c * y[u][0], x[u][1] - c * y[u][1]};
      if (NTS) __builtin_nontemporal_store(o, r + base + u * BLOCK); else r[base + u * BLOCK] = o;
    }
  }
}

template <typename F>
double time_us(F &&launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, (double)ms * 1e3 / reps);
  }
  return best;
}

int main() {
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  const int64_t m = 16384, n = 16384, nd = m * n, nvec = nd / 2;   // 2 GiB
  const double bytes = 8.0 * nd;
  double *buf, *out;
  CK(hipMalloc(&buf, sizeof(double) * nd));
  CK(hipMalloc(&out, sizeof(double) * 1 << 20));
  CK(hipMemset(buf, 0, sizeof(double) * nd));
  // clocks up
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((read1d<256, 8, true>), dim3(cus * 4), dim3(256), 0, 0, (const f64x2 *)buf, nvec, out);
  CK(hipDeviceSynchronize());
  printf("# %d CUs, 2 GiB read-only, us per pass and TB/s\n", cus);
#define R1(BLOCK, UNR, NT, PERCU)                                                                                         \
  {                                                                                                                        \
    const int g = PERCU > 0 ? cus * PERCU : (int)((nvec + (int64_t)BLOCK * UNR - 1) / ((int64_t)BLOCK * UNR));             \
    const double us = time_us([&] { hipLaunchKernelGGL((read1d<BLOCK, UNR, NT>), dim3(g), dim3(BLOCK), 0, 0, (const f64x2 *)buf, nvec, out); }, 10); \
    printf("1-D block %3d unr %2d nt %d grid %7d (%d/CU): %8.1f us  %.2f TB/s (%.3f)\n", BLOCK, UNR, (int)NT, g, PERCU, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0); \
  }
  R1(256, 4, false, 4) R1(256, 4, true, 4) R1(256, 8, false, 4) R1(256, 8, true, 4) R1(256, 8, true, 8) R1(256, 8, true, 2) R1(256, 8, true, 0)
  R1(256, 16, true, 2) R1(256, 16, true, 4) R1(512, 8, true, 1) R1(512, 8, false, 1) R1(512, 8, true, 2) R1(512, 8, true, 4) R1(512, 16, true, 1) R1(512, 16, true, 2) R1(512, 4, true, 4)
#define R2(RB, UNR, NT)                                                                                                    \
  {                                                                                                                        \
    const int g = (int)(m / RB);                                                                                           \
    const double us = time_us([&] { hipLaunchKernelGGL((read2d<RB, UNR, NT>), dim3(g), dim3(512), 0, 0, buf, m, n, out); }, 10); \
    printf("2-D row bands RB %3d unr %2d nt %d grid %5d: %8.1f us  %.2f TB/s (%.3f)\n", RB, UNR, (int)NT, g, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0); \
  }
  R2(64, 8, true) R2(64, 8, false) R2(32, 8, true) R2(16, 8, true) R2(64, 16, true) R2(64, 4, true)
#define RK(K, UNR, NT, PERCU)                                                                                              \
  {                                                                                                                        \
    Ptrs P;                                                                                                                \
    const int64_t each = (nvec / K) & ~(int64_t)255;                                                                       \
    for (int k = 0; k < K; ++k) P.p[k] = (const f64x2 *)buf + (int64_t)k * each;                                           \
    const int g = cus * PERCU;                                                                                             \
    const double us = time_us([&] { hipLaunchKernelGGL((readk<K, UNR, NT>), dim3(g), dim3(256), 0, 0, P, each, out); }, 10); \
    printf("%2d streams (lockstep) unr %2d nt %d grid %d/CU: %8.1f us  %.2f TB/s (%.3f)\n", K, UNR, (int)NT, PERCU, us, 16.0 * each * K / us / 1e6, 16.0 * each * K / us / 1e6 / 8.0); \
  }
  RK(1, 8, true, 4) RK(2, 8, true, 4) RK(4, 8, true, 4) RK(8, 8, true, 4) RK(10, 10, true, 4) RK(20, 20, true, 4) RK(20, 20, true, 2) RK(10, 20, true, 4) RK(2, 16, true, 4)

#define RKW(K, NT, BATCH)                                                                                                  \
  {                                                                                                                        \
    Ptrs P;                                                                                                                \
    const int64_t each = (nvec / (K + 1)) & ~(int64_t)255;                                                                 \
    for (int k = 0; k < K; ++k) P.p[k] = (const f64x2 *)buf + (int64_t)(k + 1) * each;                                     \
    const int g = (int)(each / 256);                                                                                       \
    const double us = time_us([&] { hipLaunchKernelGGL((readk_w<K, NT, BATCH>), dim3(g), dim3(256), 0, 0, P, (f64x2 *)buf, each); }, 10); \
    printf("%2d read streams + 1 write stream, one vector per thread, batches of %2d, nt %d: %8.1f us  %.2f TB/s (%.3f)\n", K, BATCH, (int)NT, us, 16.0 * each * (K + 1) / us / 1e6, 16.0 * each * (K + 1) / us / 1e6 / 8.0); \
  }
  RKW(10, true, 10) RKW(20, true, 10) RKW(20, true, 20) RKW(20, true, 5) RKW(40, true, 8) RKW(40, true, 20) RKW(20, false, 10) RKW(2, true, 2) RKW(1, true, 1)

#define RKW2(K, NT, BATCH)                                                                                                 \
  {                                                                                                                        \
    Ptrs P;                                                                                                                \
    const int64_t each = (nvec / (K + 2)) & ~(int64_t)255;                                                                 \
    for (int k = 0; k < K; ++k) P.p[k] = (const f64x2 *)buf + (int64_t)(k + 2) * each;                                     \
    const int g = (int)(each / 256);                                                                                       \
    const double us = time_us([&] { hipLaunchKernelGGL((readk_w2<K, NT, BATCH>), dim3(g), dim3(256), 0, 0, P, (f64x2 *)buf, (f64x2 *)buf + each, each, out); }, 10); \
    printf("%2d read streams + 2 write streams (push! shape), one vector per thread, batches of %2d, nt %d: %8.1f us  %.2f TB/s (%.3f)\n", K, BATCH, (int)NT, us, 16.0 * each * (K + 2) / us / 1e6, 16.0 * each * (K + 2) / us / 1e6 / 8.0); \
  }
  RKW2(11, true, 11) RKW2(22, true, 11) RKW2(22, true, 22) RKW2(20, true, 10) RKW2(10, true, 10) RKW2(42, true, 14) RKW2(2, true, 2)

  {
    const int64_t ne = 100000000, nv = ne / 2;     // the headline's n = 1e8 doubles per operand: 3 x 800 MB inside the 2 GiB buffer? no: own buffers
    double *a, *b, *r;
    CK(hipMalloc(&a, 8 * ne)); CK(hipMalloc(&b, 8 * ne)); CK(hipMalloc(&r, 8 * ne));
    CK(hipMemset(a, 0, 8 * ne)); CK(hipMemset(b, 0, 8 * ne));
#define MAPV(UNROLL, NTL, NTS, BLOCK)                                                                                      \
    {                                                                                                                      \
      const int g = (int)((nv + (int64_t)BLOCK * UNROLL - 1) / ((int64_t)BLOCK * UNROLL));                                 \
      const double us = time_us([&] { hipLaunchKernelGGL((map2r1w<UNROLL, NTL, NTS, BLOCK>), dim3(g), dim3(BLOCK), 0, 0, (f64x2 *)r, (const f64x2 *)a, (const f64x2 *)b, nv, 0.5); }, 20); \
      printf("map 2R+1W n=1e8 f64 block %3d unroll %d ntl %d nts %d: %8.1f us  %.2f TB/s (%.3f)\n", BLOCK, UNROLL, (int)NTL, (int)NTS, us, 24.0 * ne / us / 1e6, 24.0 * ne / us / 1e6 / 8.0); \
    }
    MAPV(4, true, true, 256) MAPV(2, true, true, 256) MAPV(1, true, true, 256) MAPV(1, true, false, 256) MAPV(2, true, false, 256) MAPV(4, true, false, 256)
    MAPV(1, true, true, 512) MAPV(2, true, true, 512) MAPV(1, true, true, 128) MAPV(2, true, true, 128) MAPV(8, true, true, 256) MAPV(1, false, true, 256) MAPV(4, true, true, 256)
    MAPV(1, true, true, 64) MAPV(2, true, true, 64) MAPV(4, true, true, 64) MAPV(1, true, true, 192) MAPV(1, true, true, 128) MAPV(4, true, true, 128) MAPV(1, true, true, 256) MAPV(4, true, true, 256) MAPV(1, true, true, 64) MAPV(1, true, true, 128)
  }
  return 0;
}
