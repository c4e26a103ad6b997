// tools/mfma_ingredients.hip — experiment (NOT product): starts from the bare f64 MFMA loop (16 waves per CU, one
// accumulator each — the w4x4 layout of gemm_glds.h) and adds the GEMM loop's other ingredients one at a time:
//   R: two conflict-free ds_read_b64 fragment reads per MFMA      B: one s_barrier per 8 MFMAs
//   D: two global_load_lds_dwordx4 (1 KiB each) per wave per 8 MFMAs (= 32 KiB per slab per CU, L2-resident source)
// Prints cycles per MFMA per SIMD at 2.4 GHz, so the step that costs the ~18 cycles shows up directly.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool R, bool B, bool D, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k(double *out, const double *src, int slabs) {
  __shared__ __attribute__((aligned(1024))) double lds[3 * 4096];   // 3 stages x 32 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 3 * 4096; i += WAVES * 64) lds[i] = 1e-3 * (i & 7);
  __syncthreads();
  f64x4 acc = {0, 0, 0, 0};
  double a = 0.5 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int offA = l4 * 64 + (((wave & 3) * 16 + l15) ^ ((l4 & 1) << 4));
  const int offB = 2048 + l4 * 64 + (((wave >> 2) * 16 + l15) ^ ((l4 & 1) << 4));
  const double *g = src + (size_t)blockIdx.x * 4096 + wave * 128 + lane * 2;
  int stage = 0;
  for (int s = 0; s < slabs; ++s) {
    const double *st = lds + stage * 4096;
    double av[2], bv[2];
    if (R) { av[0] = st[offA]; bv[0] = st[offB]; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (R && ks + 1 < 8) { av[(ks + 1) & 1] = st[offA + (ks + 1) * 256]; bv[(ks + 1) & 1] = st[offB + (ks + 1) * 256]; }
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(R ? av[ks & 1] : a, R ? bv[ks & 1] : b, acc, 0, 0, 0);
      if (B && ks == 3) {
        if (D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (D && (ks == 4 || ks == 6)) {
        const int p = (ks - 4) >> 1;
        int ns = stage == 0 ? 2 : stage - 1;
        double *l = lds + ns * 4096 + (wave + p * WAVES) * 128;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + p * 2048 + (s & 7) * 8),
                                         (__attribute__((address_space(3))) void *)l, 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[(size_t)blockIdx.x * WAVES * 64 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  double *out, *src;
  CK(hipMalloc(&out, 8 << 20));
  CK(hipMalloc(&src, 64 << 20));
  CK(hipMemset(src, 0, 64 << 20));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int cus = 256, slabs = 2048;
  auto run = [&](auto kern, const char *nm, int waves) -> int {
    hipLaunchKernelGGL(kern, dim3(cus), dim3(waves * 64), 0, 0, out, (const double *)src, 8);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(cus), dim3(waves * 64), 0, 0, out, (const double *)src, slabs);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)slabs * 8 * waves / 4;
    printf("%-28s waves/CU %2d : %8.3f ms  %6.1f cycles/MFMA/SIMD @2.4GHz  (%5.1f TF equivalent)\n", nm, waves, ms,
           ms * 1e-3 * 2.4e9 / mfma_per_simd, cus * 4 * mfma_per_simd * 2048.0 / ms / 1e9);
    return 0;
  };
  for (int rep = 0; rep < 2; ++rep) {
    run(k<false, false, false, 16>, "mfma only", 16);
    run(k<true, false, false, 16>, "+reads", 16);
    run(k<false, true, false, 16>, "+barrier", 16);
    run(k<true, true, false, 16>, "+reads+barrier", 16);
    run(k<false, false, true, 16>, "+dma", 16);
    run(k<false, true, true, 16>, "+dma+barrier", 16);
    run(k<true, true, true, 16>, "+reads+barrier+dma", 16);
    run(k<true, false, true, 16>, "+reads+dma", 16);
  }
  return 0;
}
