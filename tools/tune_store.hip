// Store FLAVOURS for the streaming kernels (round 6, second half): the K-reads + 1-write passes sit at 0.75-0.80 of HBM peak while
// the same reads alone reach 0.87-0.90 (profiles/r05_tune_read.txt). MI355X_MICROARCH.md: plain / sc0 / nt stores KEEP the
// line in the XCD's L2 (written back whenever it is evicted), sc1 / sc0 sc1 stores are write-through and DROP it. Does the
// flavour of the one write stream move the pass?
//   hipcc -O3 --offload-arch=gfx950 tools/tune_store.hip -o tools/tune_store && tools/tune_store
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// SF: 0 plain, 1 nt (builtin), 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0, 6 sc0 nt, 7 sc0 sc1 nt
template <int SF>
__device__ __forceinline__ void put(f64x2 *p, f64x2 v) {
  if constexpr (SF == 0) *p = v;
  else if constexpr (SF == 1) __builtin_nontemporal_store(v, p);
  else if constexpr (SF == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SF == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SF == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SF == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (SF == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
static const char *sfname[] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0", "sc0 nt", "sc0 sc1 nt"};

// the library's streaming-map shape (stream_kernels.h: map_kernel): 2 reads + 1 write, one chunk per workgroup
template <int UNROLL, int SF, int BLOCK>
__global__ void __launch_bounds__(BLOCK) map2r1w(f64x2 *__restrict__ r, const f64x2 *__restrict__ a, const f64x2 *__restrict__ b, int64_t nvec, double c) {
  const int64_t base = (int64_t)blockIdx.x * BLOCK * UNROLL + threadIdx.x;
  f64x2 x[UNROLL], y[UNROLL];
  if (base + (int64_t)(UNROLL - 1) * BLOCK < nvec) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      x[u] = __builtin_nontemporal_load(a + base + u * BLOCK);
      y[u] = __builtin_nontemporal_load(b + base + u * BLOCK);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      f64x2 o = {x[u][0] - c * y[u][0], x[u][1] - c * y[u][1]};
      put<SF>(r + base + u * BLOCK, o);
    }
  }
}

// 1 read + 1 write (opDiagonal with a scalar, copy)
template <int UNROLL, int SF>
__global__ void __launch_bounds__(256) map1r1w(f64x2 *__restrict__ r, const f64x2 *__restrict__ a, int64_t nvec, double c) {
  const int64_t base = (int64_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  f64x2 x[UNROLL];
  if (base + (int64_t)(UNROLL - 1) * 256 < nvec) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) x[u] = __builtin_nontemporal_load(a + base + u * 256);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      f64x2 o = {c * x[u][0], c * x[u][1]};
      put<SF>(r + base + u * 256, o);
    }
  }
}

// the combine shape of the quasi-Newton applies: K read streams + 1 write stream, one vector per thread, batches of BATCH
struct Ptrs { const f64x2 *p[48]; };
template <int K, int BATCH, int SF>
__global__ void __launch_bounds__(256) readk_w(Ptrs P, f64x2 *__restrict__ res, int64_t nvec_each) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec_each) return;
  f64x2 q = {0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += BATCH) {
    f64x2 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) v[u] = __builtin_nontemporal_load(P.p[k0 + u] + i);
#pragma unroll
    for (int u = 0; u < BATCH; ++u) q += v[u];
  }
  put<SF>(res + i, q);
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
  const int64_t ne = 100000000, nv = ne / 2;     // the headline's n = 1e8 doubles per operand
  double *a, *b, *r, *big;
  CK(hipMalloc(&a, 8 * ne)); CK(hipMalloc(&b, 8 * ne)); CK(hipMalloc(&r, 8 * ne));
  CK(hipMemset(a, 0, 8 * ne)); CK(hipMemset(b, 0, 8 * ne));
  const int64_t nd = 16384ll * 16384, nvec = nd / 2;   // 2 GiB for the K-stream shape
  CK(hipMalloc(&big, 8 * nd));
  CK(hipMemset(big, 0, 8 * nd));
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((map2r1w<1, 1, 256>), dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, 0, (f64x2 *)r, (const f64x2 *)a, (const f64x2 *)b, nv, 0.5);
  CK(hipDeviceSynchronize());
  {  // placement of the three streams: b and r shifted against a by 0 ... 1 MiB (does the relative channel / bank phase matter?)
    char *pool;
    const int64_t span = 8 * ne + (8ll << 20);
    CK(hipMalloc(&pool, 3 * span));
    CK(hipMemset(pool, 0, 3 * span));
    const int64_t offs[] = {0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1048576};
    for (int rep = 0; rep < 2; ++rep)
      for (int64_t o1 : offs) {
        const f64x2 *pa = (const f64x2 *)pool, *pb = (const f64x2 *)(pool + span + o1);
        f64x2 *pr = (f64x2 *)(pool + 2 * span + 2 * o1);
        const int g = (int)((nv + 255) / 256);
        const double us = time_us([&] { hipLaunchKernelGGL((map2r1w<1, 1, 256>), dim3(g), dim3(256), 0, 0, pr, pa, pb, nv, 0.5); }, 20);
        printf("map 2R+1W n=1e8 nt, b shifted by %8lld B, r by %8lld B against a: %8.1f us (%.3f)\n", (long long)o1, (long long)(2 * o1), us, 24.0 * ne / us / 1e6 / 8.0);
      }
    CK(hipFree(pool));
  }
  for (int round = 0; round < 1; ++round) {
#define MAPV(UNROLL, SF, BLOCK)                                                                                            \
    {                                                                                                                      \
      const int g = (int)((nv + (int64_t)BLOCK * UNROLL - 1) / ((int64_t)BLOCK * UNROLL));                                 \
      const double us = time_us([&] { hipLaunchKernelGGL((map2r1w<UNROLL, SF, BLOCK>), dim3(g), dim3(BLOCK), 0, 0, (f64x2 *)r, (const f64x2 *)a, (const f64x2 *)b, nv, 0.5); }, 20); \
      printf("map 2R+1W n=1e8 f64 block %3d unroll %d store %-10s: %8.1f us  %.2f TB/s (%.3f)\n", BLOCK, UNROLL, sfname[SF], us, 24.0 * ne / us / 1e6, 24.0 * ne / us / 1e6 / 8.0); \
    }
    MAPV(1, 0, 256) MAPV(1, 1, 256) MAPV(1, 2, 256) MAPV(1, 3, 256) MAPV(1, 4, 256) MAPV(1, 5, 256) MAPV(1, 6, 256) MAPV(1, 7, 256)
    MAPV(4, 0, 256) MAPV(4, 1, 256) MAPV(4, 2, 256) MAPV(4, 3, 256) MAPV(4, 4, 256) MAPV(4, 5, 256) MAPV(4, 6, 256) MAPV(4, 7, 256)
#define MAP1(UNROLL, SF)                                                                                                   \
    {                                                                                                                      \
      const int g = (int)((nv + 256ll * UNROLL - 1) / (256ll * UNROLL));                                                   \
      const double us = time_us([&] { hipLaunchKernelGGL((map1r1w<UNROLL, SF>), dim3(g), dim3(256), 0, 0, (f64x2 *)r, (const f64x2 *)a, nv, 0.5); }, 20); \
      printf("map 1R+1W n=1e8 f64 unroll %d store %-10s: %8.1f us  %.2f TB/s (%.3f)\n", UNROLL, sfname[SF], us, 16.0 * ne / us / 1e6, 16.0 * ne / us / 1e6 / 8.0); \
    }
    MAP1(1, 0) MAP1(1, 1) MAP1(1, 2) MAP1(1, 3) MAP1(1, 4) MAP1(4, 1) MAP1(4, 2) MAP1(4, 3)
#define RKW(K, BATCH, SF)                                                                                                  \
    {                                                                                                                      \
      Ptrs P;                                                                                                              \
      const int64_t each = (nvec / (K + 1)) & ~(int64_t)255;                                                               \
      for (int k = 0; k < K; ++k) P.p[k] = (const f64x2 *)big + (int64_t)(k + 1) * each;                                   \
      const int g = (int)(each / 256);                                                                                     \
      const double us = time_us([&] { hipLaunchKernelGGL((readk_w<K, BATCH, SF>), dim3(g), dim3(256), 0, 0, P, (f64x2 *)big, each); }, 10); \
      printf("%2d read streams + 1 write stream, batches of %2d, store %-10s: %8.1f us  %.2f TB/s (%.3f)\n", K, BATCH, sfname[SF], us, 16.0 * each * (K + 1) / us / 1e6, 16.0 * each * (K + 1) / us / 1e6 / 8.0); \
    }
    RKW(20, 10, 0) RKW(20, 10, 1) RKW(20, 10, 2) RKW(20, 10, 3) RKW(20, 10, 4) RKW(20, 10, 7)
    RKW(10, 10, 1) RKW(10, 10, 2) RKW(10, 10, 3) RKW(40, 8, 1) RKW(40, 8, 2) RKW(40, 8, 3)
  }
  return 0;
}
