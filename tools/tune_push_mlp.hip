// Experiment harness (not part of the library): the HBM-bound push pass — two caller vectors and NC = 10 panel columns,
// dual accumulators per column — with its loads (0) left to the compiler's schedule, (1) all pinned ahead of the FMAs,
// (2) in two batches of five columns, (3) in three batches (4 + 3 + 3). profiles/r04_bench_push.txt: the library's
// pass runs at 0.70-0.75 of peak at n = 5e7; its fully branch-free form (FAST) lost 6 % there.
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=off tools/tune_push_mlp.hip -o tools/tune_push_mlp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>

constexpr int NC = 10, kBlock = 256;
typedef double V __attribute__((ext_vector_type(2)));
struct Cols { const double *p[NC]; };

template <int VARIANT>
__global__ void __launch_bounds__(kBlock)
push10(Cols cols, const double *__restrict__ x1, const double *__restrict__ x2, int64_t nvec, double *__restrict__ partials) {
  const int tid = threadIdx.x;
  double a1[NC], a2[NC], e12 = 0.0, e22 = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) a1[c] = a2[c] = 0.0;
  const int64_t nchunks = (nvec + kBlock - 1) / kBlock;
  auto fmas = [&](const V &cv, const V &u, const V &v, int c) {
    a1[c] = fma(cv[0], u[0], a1[c]); a1[c] = fma(cv[1], u[1], a1[c]);
    a2[c] = fma(cv[0], v[0], a2[c]); a2[c] = fma(cv[1], v[1], a2[c]);
  };
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t i = ch * kBlock + tid;
    if (i >= nvec) continue;
    const V u = __builtin_nontemporal_load(reinterpret_cast<const V *>(x1) + i);
    const V v = __builtin_nontemporal_load(reinterpret_cast<const V *>(x2) + i);
    constexpr int NB = VARIANT <= 1 ? 1 : VARIANT;            // batches
    constexpr int PER = (NC + NB - 1) / NB;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      V cv[PER];
#pragma unroll
      for (int c = 0; c < PER; ++c)
        if (b * PER + c < NC) cv[c] = __builtin_nontemporal_load(reinterpret_cast<const V *>(cols.p[b * PER + c]) + i);
      if constexpr (VARIANT >= 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < PER; ++c)
        if (b * PER + c < NC) fmas(cv[c], u, v, b * PER + c);
      if constexpr (VARIANT >= 1) __builtin_amdgcn_sched_barrier(0);
    }
    e12 = fma(u[0], v[0], e12); e12 = fma(u[1], v[1], e12);
    e22 = fma(v[0], v[0], e22); e22 = fma(v[1], v[1], e22);
  }
  double s = e12 + e22;
#pragma unroll
  for (int c = 0; c < NC; ++c) s += a1[c] + a2[c];
  partials[(int64_t)blockIdx.x * kBlock + tid] = s;
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  for (int64_t n : {int64_t(1) << 22, int64_t(1) << 24, int64_t(50000000)}) {
    double *panel, *x, *partials;
    (void)hipMalloc(&panel, sizeof(double) * n * NC);
    (void)hipMalloc(&x, sizeof(double) * n * 2);
    (void)hipMalloc(&partials, sizeof(double) * 8192 * kBlock);
    (void)hipMemset(panel, 0x3c, sizeof(double) * n * NC);
    (void)hipMemset(x, 0x3c, sizeof(double) * n * 2);
    Cols cols;
    for (int c = 0; c < NC; ++c) cols.p[c] = panel + (int64_t)c * n;
    const int64_t nvec = n / 2, need = (nvec + kBlock - 1) / kBlock;
    for (int per_cu : {4, 8}) {
      const int grid = (int)std::min<int64_t>(need, (int64_t)ncu * per_cu);
      double us[4];
      for (int var = 0; var < 4; ++var) {
        auto launch = [&]() {
          if (var == 0) hipLaunchKernelGGL(push10<0>, dim3(grid), dim3(kBlock), 0, 0, cols, x, x + n, nvec, partials);
          else if (var == 1) hipLaunchKernelGGL(push10<1>, dim3(grid), dim3(kBlock), 0, 0, cols, x, x + n, nvec, partials);
          else if (var == 2) hipLaunchKernelGGL(push10<2>, dim3(grid), dim3(kBlock), 0, 0, cols, x, x + n, nvec, partials);
          else hipLaunchKernelGGL(push10<3>, dim3(grid), dim3(kBlock), 0, 0, cols, x, x + n, nvec, partials);
        };
        for (int w = 0; w < 3; ++w) launch();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int reps = n > (1 << 22) ? 20 : 100;
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        us[var] = ms * 1e3 / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      }
      const double gb = 8.0 * n * (NC + 2) / 1e9;
      printf("n=%9lld grid=%5d (%d/CU): compiler %7.1f us (%5.2f TB/s) | pinned %7.1f (%5.2f) | 2 batches %7.1f (%5.2f) | 3 batches %7.1f (%5.2f)\n",
             (long long)n, grid, per_cu, us[0], gb / us[0] * 1e3, us[1], gb / us[1] * 1e3, us[2], gb / us[2] * 1e3, us[3], gb / us[3] * 1e3);
      fflush(stdout);
    }
    (void)hipFree(panel); (void)hipFree(x); (void)hipFree(partials);
  }
  return 0;
}
