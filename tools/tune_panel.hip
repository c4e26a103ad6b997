// tune_panel.hip — standalone: achievable HBM rate for reading a column-major n x n matrix in
// R x C panels (one workgroup per panel), loads only. Guides the opHermitian panel shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// panel R rows x C cols; 256 threads; thread t: row-pair rp = t % (R/2), col group cg = t / (R/2); cols cg + CG*k
template <int R, int C, bool NT>
__global__ void __launch_bounds__(256) panel_read(const double* __restrict__ A, int64_t lda, int64_t n, double* out) {
  constexpr int RP = R / 2, CG = 256 / RP, K = C / CG;
  const int64_t nbr = n / R;
  const int64_t pr = blockIdx.x % nbr, pc = blockIdx.x / nbr;
  const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
  const double* base = A + pr * R + 2 * rp + (pc * C + cg) * lda;
  f64x2 e[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const f64x2* p = reinterpret_cast<const f64x2*>(base + (int64_t)k * CG * lda);
    e[k] = NT ? __builtin_nontemporal_load(p) : *p;
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += e[k][0] + e[k][1];
  if (s == 1.2345e300) out[0] = s;
}

template <int R, int C, bool NT>
void run(const double* A, int64_t n, double* out) {
  const int64_t grid = (n / R) * (n / C);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((panel_read<R, C, NT>), dim3(grid), dim3(256), 0, 0, A, n, n, out);
  CK(hipEventRecord(e0, 0));
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((panel_read<R, C, NT>), dim3(grid), dim3(256), 0, 0, A, n, n, out);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("panel %4d x %3d nt=%d: %.1f us  %.0f GB/s\n", R, C, (int)NT, ms * 1e3, 8.0 * n * n / ms / 1e6);
}

int main() {
  const int64_t n = 16384;
  double *A, *out;
  CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&out, 8));
  CK(hipMemset(A, 1, n * n * 8));
  run<64, 64, true>(A, n, out);  run<64, 64, false>(A, n, out);
  run<128, 32, true>(A, n, out); run<128, 64, true>(A, n, out);
  run<256, 32, true>(A, n, out); run<256, 32, false>(A, n, out);
  run<256, 16, true>(A, n, out); run<512, 16, true>(A, n, out); run<512, 8, true>(A, n, out);
  run<256, 64, true>(A, n, out);
  return 0;
}
