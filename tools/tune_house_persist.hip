// Would the HEADLINE (opHouseholder mul!, n = 1e8 fp64: a dots pass + an update pass, 40 B/elt) gain from ONE persistent launch
// that PARKS part of h and v on the chip between its two phases — registers (~0.4 MB per CU idle in a streaming kernel), LDS
// (144 KiB per CU) and whatever the L2 keeps across the phases — the way the cache-resident quasi-Newton apply does (round 6)?
// On-chip capacity ~ 256 x (0.4 + 0.14) MB + 32 MB L2 ~ 170 MB of the 1.6 GB the update pass re-reads: at best -7 % of its bytes.
//   hipcc -O3 --offload-arch=gfx950 tools/tune_house_persist.hip -o tools/tune_house_persist && tools/tune_house_persist
// Rows: two launches in the library's shapes (dots: one 512-thread workgroup per CU; update: 256 threads x 4 vectors, one chunk
// per workgroup, back to front) against the persistent launch with PR chunks per workgroup parked in registers and PL in LDS.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int PB = 512;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---- two launches
__global__ void __launch_bounds__(PB) dots_kernel(const f64x2 *__restrict__ h, const f64x2 *__restrict__ v, int64_t nvec, double *__restrict__ part) {
  double acc = 0.0;
  const int64_t per = (nvec + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < nvec ? lo + per : nvec;
  for (int64_t i = lo + threadIdx.x; i < hi; i += PB * 4) {
    f64x2 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * PB < hi) { a[u] = __builtin_nontemporal_load(h + i + u * PB); b[u] = __builtin_nontemporal_load(v + i + u * PB); }
      else { a[u] = f64x2{0, 0}; b[u] = f64x2{0, 0}; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fma(a[u][0], b[u][0], fma(a[u][1], b[u][1], acc));
  }
  __shared__ double red[PB / 64];
  const double w = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int k = 0; k < PB / 64; ++k) s += red[k];
    part[blockIdx.x] = s;
  }
}
__global__ void __launch_bounds__(256) update_kernel(f64x2 *__restrict__ r, const f64x2 *__restrict__ h, const f64x2 *__restrict__ v, int64_t nvec,
                                                     const double *__restrict__ part, int nparts) {
  double dot = 0;
  for (int k = 0; k < nparts; ++k) dot += part[k];      // (the library: one finalize launch; here every workgroup re-adds 256 values)
  const double c = 2.0 * dot;
  const int64_t nch = (nvec + 1023) / 1024, ch = nch - 1 - blockIdx.x, base = ch * 1024 + threadIdx.x;
  f64x2 a[4], b[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (base + u * 256 < nvec) { a[u] = __builtin_nontemporal_load(h + base + u * 256); b[u] = __builtin_nontemporal_load(v + base + u * 256); }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (base + u * 256 < nvec) {
      f64x2 o = {b[u][0] - c * a[u][0], b[u][1] - c * a[u][1]};
      __builtin_nontemporal_store(o, r + base + u * 256);
    }
}

// ---- one persistent launch: PR chunks (of PB vectors) per workgroup parked in registers, PL in LDS
template <int PR, int PL, bool NT2>
__global__ void __launch_bounds__(PB) persist_kernel(f64x2 *__restrict__ r, const f64x2 *__restrict__ h, const f64x2 *__restrict__ v, int64_t nvec,
                                                     int cpw, unsigned long long *__restrict__ slots, unsigned epoch) {
  extern __shared__ __align__(16) unsigned char dyn[];
  f64x2 *Lh = reinterpret_cast<f64x2 *>(dyn), *Lv = Lh + (int64_t)PL * PB;
  const int tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
  const int64_t nch = (nvec + PB - 1) / PB;
  const int64_t c0 = (int64_t)b * cpw < nch ? (int64_t)b * cpw : nch, c1 = c0 + cpw < nch ? c0 + cpw : nch;
  auto full = [&](int64_t ch) { return (ch + 1) * PB <= nvec; };
  f64x2 Rh[PR > 0 ? PR : 1], Rv[PR > 0 ? PR : 1];
  double acc = 0.0;
  // phase A: the first PR chunks stay in registers, the next PL in LDS, the rest streams (4 chunks per step)
#pragma unroll
  for (int j = 0; j < PR; ++j) {
    const int64_t ch = c0 + j;
    if (ch < c1 && full(ch)) {
      Rh[j] = __builtin_nontemporal_load(h + ch * PB + tid);
      Rv[j] = __builtin_nontemporal_load(v + ch * PB + tid);
    } else {
      Rh[j] = f64x2{0, 0};
      Rv[j] = f64x2{0, 0};
    }
  }
#pragma unroll
  for (int j = 0; j < PR; ++j) acc = fma(Rh[j][0], Rv[j][0], fma(Rh[j][1], Rv[j][1], acc));
  for (int j = 0; j < PL; ++j) {
    const int64_t ch = c0 + PR + j;
    f64x2 a = {0, 0}, bb = {0, 0};
    if (ch < c1 && full(ch)) { a = __builtin_nontemporal_load(h + ch * PB + tid); bb = __builtin_nontemporal_load(v + ch * PB + tid); }
    Lh[j * PB + tid] = a;
    Lv[j * PB + tid] = bb;
    acc = fma(a[0], bb[0], fma(a[1], bb[1], acc));
  }
  for (int64_t ch = c0 + PR + PL; ch < c1; ch += 4) {
    f64x2 a[4], bb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ch + u < c1 && full(ch + u)) {
        a[u] = NT2 ? __builtin_nontemporal_load(h + (ch + u) * PB + tid) : h[(ch + u) * PB + tid];
        bb[u] = NT2 ? __builtin_nontemporal_load(v + (ch + u) * PB + tid) : v[(ch + u) * PB + tid];
      } else { a[u] = f64x2{0, 0}; bb[u] = f64x2{0, 0}; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fma(a[u][0], bb[u][0], fma(a[u][1], bb[u][1], acc));
  }
  // exchange: one slot per workgroup (value tagged with the launch's epoch), everybody gathers in a fixed order
  __shared__ double red[PB / 64], sdot;
  const double w = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = w;
  __syncthreads();
  if (tid == 0) {
    double s = 0;
    for (int k = 0; k < PB / 64; ++k) s += red[k];
    __hip_atomic_store(slots + 2 * b, (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(slots + 2 * b + 1, (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  double p = 0.0;
  if (tid < G) {
    while (__hip_atomic_load(slots + 2 * tid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)epoch) __builtin_amdgcn_s_sleep(2);
    p = __longlong_as_double((long long)__hip_atomic_load(slots + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  const double w2 = wave_sum(p);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = w2;
  __syncthreads();
  if (tid == 0) {
    double s = 0;
    for (int k = 0; k < PB / 64; ++k) s += red[k];
    sdot = s;
  }
  __syncthreads();
  const double c = 2.0 * sdot;
  // phase B: back to front over the streamed chunks, then LDS, then registers
  for (int64_t ch = c1 - 1; ch >= c0 + PR + PL; ch -= 4) {
    f64x2 a[4], bb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ch - u >= c0 + PR + PL && full(ch - u)) {
        a[u] = NT2 ? __builtin_nontemporal_load(h + (ch - u) * PB + tid) : h[(ch - u) * PB + tid];
        bb[u] = NT2 ? __builtin_nontemporal_load(v + (ch - u) * PB + tid) : v[(ch - u) * PB + tid];
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ch - u >= c0 + PR + PL && full(ch - u)) {
        f64x2 o = {bb[u][0] - c * a[u][0], bb[u][1] - c * a[u][1]};
        __builtin_nontemporal_store(o, r + (ch - u) * PB + tid);
      }
  }
  for (int j = PL - 1; j >= 0; --j) {
    const int64_t ch = c0 + PR + j;
    if (ch < c1 && full(ch)) {
      const f64x2 a = Lh[j * PB + tid], bb = Lv[j * PB + tid];
      f64x2 o = {bb[0] - c * a[0], bb[1] - c * a[1]};
      __builtin_nontemporal_store(o, r + ch * PB + tid);
    }
  }
#pragma unroll
  for (int j = PR - 1; j >= 0; --j) {
    const int64_t ch = c0 + j;
    if (ch < c1 && full(ch)) {
      f64x2 o = {Rv[j][0] - c * Rh[j][0], Rv[j][1] - c * Rh[j][1]};
      __builtin_nontemporal_store(o, r + ch * PB + tid);
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
  const int64_t ne = 100000000, nv = ne / 2;
  double *h, *v, *r, *part;
  unsigned long long *slots;
  CK(hipMalloc(&h, 8 * ne)); CK(hipMalloc(&v, 8 * ne)); CK(hipMalloc(&r, 8 * ne)); CK(hipMalloc(&part, 8 * 1024)); CK(hipMalloc(&slots, 16 * 1024));
  CK(hipMemset(h, 0, 8 * ne)); CK(hipMemset(v, 0, 8 * ne)); CK(hipMemset(slots, 0, 16 * 1024));
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(dots_kernel, dim3(cus), dim3(PB), 0, 0, (const f64x2 *)h, (const f64x2 *)v, nv, part);
  CK(hipDeviceSynchronize());
  const int gu = (int)((nv + 1023) / 1024);
  unsigned epoch = 0;
  for (int round = 0; round < 2; ++round) {
    {
      const double us = time_us([&] {
        hipLaunchKernelGGL(dots_kernel, dim3(cus), dim3(PB), 0, 0, (const f64x2 *)h, (const f64x2 *)v, nv, part);
        hipLaunchKernelGGL(update_kernel, dim3(gu), dim3(256), 0, 0, (f64x2 *)r, (const f64x2 *)h, (const f64x2 *)v, nv, part, cus);
      }, 20);
      printf("two launches (dots 1 wg/CU x 512, update 256 x 4 back to front): %8.1f us  %.3f of peak on 40 B/elt\n", us, 40.0 * ne / us / 1e6 / 8.0);
    }
    const int64_t nch = (nv + PB - 1) / PB;
    const int cpw = (int)((nch + cus - 1) / cus);
#define PERSIST(PR_, PL_, NT2_)                                                                                               \
    {                                                                                                                          \
      const size_t dyn = (size_t)PL_ * PB * 32;                                                                                \
      CK(hipFuncSetAttribute((const void *)persist_kernel<PR_, PL_, NT2_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
      const double us = time_us([&] {                                                                                          \
        ++epoch;                                                                                                               \
        hipLaunchKernelGGL((persist_kernel<PR_, PL_, NT2_>), dim3(cus), dim3(PB), dyn, 0, (f64x2 *)r, (const f64x2 *)h, (const f64x2 *)v, nv, cpw, slots, epoch); \
      }, 20);                                                                                                                  \
      printf("persistent, %2d chunks in registers + %d in LDS per workgroup (of %d; %.1f %% parked), phase-B loads %s: %8.1f us  %.3f\n", PR_, PL_, cpw, \
             100.0 * (PR_ + PL_) / cpw, NT2_ ? "nt" : "default", us, 40.0 * ne / us / 1e6 / 8.0);                              \
    }
    PERSIST(0, 0, true) PERSIST(0, 0, false) PERSIST(0, 9, true) PERSIST(8, 9, true) PERSIST(16, 9, true) PERSIST(22, 9, true) PERSIST(22, 9, false) PERSIST(22, 0, true)
  }
  return 0;
}
