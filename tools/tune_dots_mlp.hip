// Experiment harness (not part of the library): does pinning the column loads of a 20-column dots pass AHEAD of the FMAs
// pay? The library's panel_dots_kernel<VEC=2, NC=20, UNROLL=1> compiles to "load one column, wait, FMA, load the next"
// (tools/isa_load_scan.py: longest run of loads = 2); at HBM-bound sizes 8-10 resident waves per SIMD hide that, at
// mid sizes they cannot. Variants: 0 = as the compiler schedules it, 1 = __builtin_amdgcn_sched_barrier(0) between the
// load loop and the FMA loop (all NC + 1 loads of a lane in flight), 2 = two batches of NC/2 columns.
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=off tools/tune_dots_mlp.hip -o tools/tune_dots_mlp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NC = 20, kBlock = 256;
typedef double V __attribute__((ext_vector_type(2)));
struct Cols { const double *p[NC]; };

template <int VARIANT>
__global__ void __launch_bounds__(kBlock)
dots20(Cols cols, const double *__restrict__ x, int64_t nvec, double *__restrict__ partials) {
  const int tid = threadIdx.x;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  const int64_t nchunks = (nvec + kBlock - 1) / kBlock;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t i = ch * kBlock + tid;
    if (i >= nvec) continue;
    const V xv = __builtin_nontemporal_load(reinterpret_cast<const V *>(x) + i);
    if constexpr (VARIANT == 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        V cv[NC / 2];
#pragma unroll
        for (int c = 0; c < NC / 2; ++c) cv[c] = __builtin_nontemporal_load(reinterpret_cast<const V *>(cols.p[h * (NC / 2) + c]) + i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC / 2; ++c) {
          acc[h * (NC / 2) + c] = fma(cv[c][0], xv[0], acc[h * (NC / 2) + c]);
          acc[h * (NC / 2) + c] = fma(cv[c][1], xv[1], acc[h * (NC / 2) + c]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      V cv[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) cv[c] = __builtin_nontemporal_load(reinterpret_cast<const V *>(cols.p[c]) + i);
      if constexpr (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        acc[c] = fma(cv[c][0], xv[0], acc[c]);
        acc[c] = fma(cv[c][1], xv[1], acc[c]);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) s += acc[c];
  partials[(int64_t)blockIdx.x * kBlock + tid] = s;
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  for (int64_t n : {int64_t(1) << 19, int64_t(1) << 20, int64_t(1) << 21, int64_t(1) << 22, int64_t(1) << 24, int64_t(50000000)}) {
    double *panel, *x, *partials;
    hipMalloc(&panel, sizeof(double) * n * NC);
    hipMalloc(&x, sizeof(double) * n);
    hipMalloc(&partials, sizeof(double) * 8192 * kBlock);
    hipMemset(panel, 0, sizeof(double) * n * NC);
    hipMemset(x, 0, sizeof(double) * n);
    Cols cols;
    for (int c = 0; c < NC; ++c) cols.p[c] = panel + (int64_t)c * n;
    const int64_t nvec = n / 2, need = (nvec + kBlock - 1) / kBlock;
    for (int per_cu : {4, 8}) {
      const int grid = (int)std::min<int64_t>(need, (int64_t)ncu * per_cu);
      double us[3];
      for (int v = 0; v < 3; ++v) {
        auto launch = [&]() {
          if (v == 0) hipLaunchKernelGGL(dots20<0>, dim3(grid), dim3(kBlock), 0, 0, cols, x, nvec, partials);
          else if (v == 1) hipLaunchKernelGGL(dots20<1>, dim3(grid), dim3(kBlock), 0, 0, cols, x, nvec, partials);
          else hipLaunchKernelGGL(dots20<2>, dim3(grid), dim3(kBlock), 0, 0, cols, x, nvec, partials);
        };
        for (int w = 0; w < 5; ++w) launch();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = n > (1 << 22) ? 20 : 200;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        us[v] = ms * 1e3 / reps;
        hipEventDestroy(e0); hipEventDestroy(e1);
      }
      const double gb = 8.0 * n * (NC + 1) / 1e9;
      printf("n=%10lld grid=%5d (%d/CU): as scheduled %8.1f us (%5.2f TB/s) | loads pinned %8.1f us (%5.2f) | two batches %8.1f us (%5.2f)\n",
             (long long)n, grid, per_cu, us[0], gb / us[0] * 1e3, us[1], gb / us[1] * 1e3, us[2], gb / us[2] * 1e3);
      fflush(stdout);
    }
    hipFree(panel); hipFree(x); hipFree(partials);
  }
  return 0;
}
