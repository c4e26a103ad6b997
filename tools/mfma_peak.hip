// tools/mfma_peak.hip — experiment (NOT product): sustained v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 rate on
// MI355X with nothing else in the loop, for 1/2/4 waves per SIMD and 1/2/4 independent accumulators per wave.
// Gives the practical ceiling (clock under matrix-pipe load) the GEMM numbers should be read against.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC>
__global__ void k64(double *out, int iters, double a0) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float *out, int iters, float a0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  float a = a0 + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double *out;
  CK(hipMalloc(&out, 8 * 1024 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int cus = 256;
  for (int rep = 0; rep < 2; ++rep)
  for (int threads : {256, 512, 1024}) {
    auto run = [&](auto kern, int nacc, const char *nm, double flop_per) -> int {
      const int iters = 4096 / nacc * 4 * 256 / threads;   // same MFMA count per SIMD in every configuration
      hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, (decltype(out))out, 16, 0.5);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, (decltype(out))out, iters, 0.5);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double nm_ = 5.0 * cus * (threads / 64) * (double)iters * 8 * nacc;
      printf("%s waves/SIMD %d acc/wave %d : %8.3f ms  %7.1f TF  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", nm, threads / 256, nacc, ms,
             nm_ * flop_per / ms / 1e9, ms * 1e-3 * 2.4e9 / (nm_ / (cus * 4)));
      return 0;
    };
    run(k64<1>, 1, "f64", 2048.0);
    run(k64<2>, 2, "f64", 2048.0);
    run(k64<4>, 4, "f64", 2048.0);
  }
  for (int threads : {256, 1024}) {
    auto run = [&](auto kern, int nacc) -> int {
      const int iters = 4096 / nacc * 8 * 256 / threads;
      hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, (float *)out, 16, 0.5f);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, (float *)out, iters, 0.5f);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double nm_ = 5.0 * cus * (threads / 64) * (double)iters * 8 * nacc;
      printf("f32 waves/SIMD %d acc/wave %d : %8.3f ms  %7.1f TF\n", threads / 256, nacc, ms, nm_ * 2048.0 / ms / 1e9);
      return 0;
    };
    run(k32<1>, 1);
    run(k32<4>, 4);
  }
  return 0;
}
