// tune_herm.hip — standalone ablation of the opHermitian 64x64 tile kernel: which phase costs what.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int HT = 64;

// LEVEL 0: triangular tile map + loads; 1: + v loads + FMAs; 2: + column butterfly + Pcol; 3: + LDS row reduce + Prow
template <int LEVEL>
__global__ void __launch_bounds__(256) herm_tile(const double* __restrict__ A, int64_t lda, const double* __restrict__ v,
                                                  int64_t n, double* __restrict__ Prow, double* __restrict__ Pcol, double* sink) {
  const int64_t t = blockIdx.x;
  int64_t I = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int64_t J = t - I * (I + 1) / 2;
  const int64_t i0 = I * HT, j0 = J * HT;
  const int tid = threadIdx.x, rp = tid & 31, cg = tid >> 5;
  const int64_t gr = i0 + 2 * rp;
  double e0[8], e1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t gc = j0 + cg + 8 * k;
    const f64x2 x = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(A + gr + gc * lda));
    e0[k] = gr > gc ? x[0] : 0.0;
    e1[k] = gr + 1 > gc ? x[1] : 0.0;
  }
  if constexpr (LEVEL == 0) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += e0[k] + e1[k];
    if (s == 1.2345e300) sink[0] = s;
    return;
  }
  const double vr0 = v[gr], vr1 = v[gr + 1];
  double prow0 = 0, prow1 = 0, pcol[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double vc = v[j0 + cg + 8 * k];
    prow0 = fma(e0[k], vc, prow0);
    prow1 = fma(e1[k], vc, prow1);
    pcol[k] = fma(e1[k], vr1, e0[k] * vr0);
  }
  if constexpr (LEVEL == 1) {
    double s = prow0 + prow1;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += pcol[k];
    if (s == 1.2345e300) sink[0] = s;
    return;
  }
  const int l5 = tid & 31;
  double w4[4], w2[2], w1;
  { const bool hi = (l5 & 16) != 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const double send = hi ? pcol[q] : pcol[4 + q]; w4[q] = (hi ? pcol[4 + q] : pcol[q]) + __shfl_xor(send, 16, 64); } }
  { const bool hi = (l5 & 8) != 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) { const double send = hi ? w4[q] : w4[2 + q]; w2[q] = (hi ? w4[2 + q] : w4[q]) + __shfl_xor(send, 8, 64); } }
  { const bool hi = (l5 & 4) != 0; const double send = hi ? w2[0] : w2[1]; w1 = (hi ? w2[1] : w2[0]) + __shfl_xor(send, 4, 64); }
  w1 += __shfl_xor(w1, 2, 64);
  w1 += __shfl_xor(w1, 1, 64);
  if ((l5 & 3) == 0) {
    const int k = ((l5 >> 4) & 1) * 4 + ((l5 >> 3) & 1) * 2 + ((l5 >> 2) & 1);
    Pcol[I * n + j0 + cg + 8 * k] = w1;
  }
  if constexpr (LEVEL == 2) {
    if (prow0 + prow1 == 1.2345e300) sink[0] = prow0;
    return;
  }
  __shared__ double red[8][HT];
  red[cg][2 * rp] = prow0;
  red[cg][2 * rp + 1] = prow1;
  __syncthreads();
  if (tid < HT) {
    double s = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[g][tid];
    Prow[J * n + i0 + tid] = s;
  }
}

template <int LEVEL>
void run(const double* A, const double* v, int64_t n, double* Prow, double* Pcol, double* sink) {
  const int64_t nb = n / HT, grid = nb * (nb + 1) / 2;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((herm_tile<LEVEL>), dim3(grid), dim3(256), 0, 0, A, n, v, n, Prow, Pcol, sink);
  CK(hipEventRecord(e0, 0));
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((herm_tile<LEVEL>), dim3(grid), dim3(256), 0, 0, A, n, v, n, Prow, Pcol, sink);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("level %d: %.1f us  %.0f GB/s (of 4n^2 = %.2f GB)\n", LEVEL, ms * 1e3, 4.0 * n * n / ms / 1e6, 4.0 * n * n / 1e9);
}

int main() {
  const int64_t n = 16384, nb = n / HT;
  double *A, *v, *Prow, *Pcol, *sink;
  CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&Prow, nb * n * 8)); CK(hipMalloc(&Pcol, nb * n * 8)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(A, 1, n * n * 8)); CK(hipMemset(v, 1, n * 8));
  run<0>(A, v, n, Prow, Pcol, sink); run<1>(A, v, n, Prow, Pcol, sink); run<2>(A, v, n, Prow, Pcol, sink); run<3>(A, v, n, Prow, Pcol, sink);
  run<0>(A, v, n, Prow, Pcol, sink); run<3>(A, v, n, Prow, Pcol, sink);
  return 0;
}
