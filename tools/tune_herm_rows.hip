// What would a ROW-BAND sweep give opHermitian at large n? (round 6, last experiment; profiles/r06_herm_policy.txt: the 256 x 32 tile
// pattern's loads alone reach 0.81 of peak, the row-band GEMV streams at 0.88.) Here a 512-thread workgroup owns a band of RB = 64
// rows and walks its columns left to right like gemv_n_rows_kernel — 32 lanes cover the band's rows of one column with 16-byte loads,
// the 16 half-waves take 16 columns at a time, 8 loads in flight per lane —, which gives the band's part of L*v complete inside the
// workgroup (no row partials) and one column partial per (band, column) for L'*v (halving butterfly over the 32 lanes: 9 exchanges
// per 8 columns). Bands are paired (b, nb-1-b) so every workgroup reads the same number of columns.
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=off tools/tune_herm_rows.hip -o tools/tune_herm_rows && tools/tune_herm_rows
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double V2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int PB = 512, RB = 64, LPR = RB / 2, NCL = PB / LPR, U = 8;

__device__ __forceinline__ double sx(double v, int m) { return __shfl_xor(v, m, 64); }

template <bool NT, bool COMPUTE>
__global__ void __launch_bounds__(PB) herm_rows(const double *__restrict__ A, int64_t lda, const double *__restrict__ v, int64_t n,
                                                double *__restrict__ Rsum, double *__restrict__ Pcol, int nb, double *sink) {
  __shared__ double sred[NCL][RB];
  const int tid = threadIdx.x, lane = tid & 63, seg = tid % LPR, cl = tid / LPR;
  const int kcol = ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);   // the column a lane group ends up with
  double keep = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    const int band = pass == 0 ? (int)blockIdx.x : nb - 1 - (int)blockIdx.x;
    if (pass == 1 && band == (int)blockIdx.x) break;
    const int64_t r0 = (int64_t)band * RB, row = r0 + 2 * seg;
    const double vr0 = v[row], vr1 = v[row + 1];
    double acc0 = 0.0, acc1 = 0.0;
    const double *base = A + row;
    double *pc = Pcol + (int64_t)band * n;
    int64_t j = cl;
    for (; j + (int64_t)(U - 1) * NCL < r0; j += (int64_t)U * NCL) {          // columns strictly left of the band's diagonal block
      V2 a[U];
      double x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const V2 *q = reinterpret_cast<const V2 *>(base + (j + (int64_t)u * NCL) * lda);
        a[u] = NT ? __builtin_nontemporal_load(q) : *q;
        x[u] = v[j + (int64_t)u * NCL];
      }
      if constexpr (!COMPUTE) {
#pragma unroll
        for (int u = 0; u < U; ++u) keep += a[u][0] + a[u][1] + x[u];
        continue;
      }
      double c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc0 = fma(a[u][0], x[u], acc0);
        acc1 = fma(a[u][1], x[u], acc1);
        c[u] = fma(a[u][1], vr1, a[u][0] * vr0);
      }
      double w4[4], w2[2], w1;
      {
        const bool hi = (lane & 16) != 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) w4[q] = (hi ? c[4 + q] : c[q]) + sx(hi ? c[q] : c[4 + q], 16);
      }
      {
        const bool hi = (lane & 8) != 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) w2[q] = (hi ? w4[2 + q] : w4[q]) + sx(hi ? w4[q] : w4[2 + q], 8);
      }
      {
        const bool hi = (lane & 4) != 0;
        w1 = (hi ? w2[1] : w2[0]) + sx(hi ? w2[0] : w2[1], 4);
      }
      w1 += sx(w1, 2);
      w1 += sx(w1, 1);
      if ((lane & 3) == 0) pc[j + (int64_t)kcol * NCL] = w1;
    }
    for (; j < r0 + RB; j += NCL) {                                             // leftover interior columns and the diagonal block (masked)
      if (j >= n) break;
      V2 a = *reinterpret_cast<const V2 *>(base + j * lda);
      if (j >= r0) {                                                              // strict lower triangle only
        a[0] = row > j ? a[0] : 0.0;
        a[1] = row + 1 > j ? a[1] : 0.0;
      }
      const double x = v[j];
      acc0 = fma(a[0], x, acc0);
      acc1 = fma(a[1], x, acc1);
      double c = fma(a[1], vr1, a[0] * vr0);
      c += sx(c, 16); c += sx(c, 8); c += sx(c, 4); c += sx(c, 2); c += sx(c, 1);
      if ((lane & 31) == 0) pc[j] = c;
    }
    sred[cl][2 * seg] = acc0;
    sred[cl][2 * seg + 1] = acc1;
    __syncthreads();
    if (tid < RB) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NCL; ++k) s += sred[k][tid];
      Rsum[r0 + tid] = s;
    }
    __syncthreads();
  }
  if (!COMPUTE && keep == 1.2345e300) sink[0] = keep;
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
  {  // correctness at n = 1024 against a host triangle product
    const int64_t n = 1024, nb = n / RB;
    std::vector<double> hA(n * n), hv(n);
    for (int64_t i = 0; i < n * n; ++i) hA[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int64_t i = 0; i < n; ++i) hv[i] = (double)((i * 40503u) % 997) / 997.0 - 0.5;
    double *A, *v, *R, *P, *sink;
    CK(hipMalloc(&A, 8 * n * n)); CK(hipMalloc(&v, 8 * n)); CK(hipMalloc(&R, 8 * n)); CK(hipMalloc(&P, 8 * nb * n)); CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(A, hA.data(), 8 * n * n, hipMemcpyHostToDevice)); CK(hipMemcpy(v, hv.data(), 8 * n, hipMemcpyHostToDevice));
    CK(hipMemset(P, 0, 8 * nb * n));
    hipLaunchKernelGGL((herm_rows<true, true>), dim3((nb + 1) / 2), dim3(PB), 0, 0, A, n, v, n, R, P, (int)nb, sink);
    CK(hipDeviceSynchronize());
    std::vector<double> hR(n), hP(nb * n);
    CK(hipMemcpy(hR.data(), R, 8 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hP.data(), P, 8 * nb * n, hipMemcpyDeviceToHost));
    double err = 0, nrm = 0;
    for (int64_t i = 0; i < n; ++i) {
      double want = 0;
      for (int64_t j = 0; j < i; ++j) want += hA[i + j * n] * hv[j];          // L v
      for (int64_t r = i + 1; r < n; ++r) want += hA[r + i * n] * hv[r];      // L' v
      double got = hR[i];
      for (int64_t b = i / RB; b < nb; ++b) got += hP[b * n + i];
      err += (got - want) * (got - want);
      nrm += want * want;
    }
    printf("# n = 1024: rel error of L v + L' v against the host %.2e\n", std::sqrt(err / nrm));
    CK(hipFree(A)); CK(hipFree(v)); CK(hipFree(R)); CK(hipFree(P)); CK(hipFree(sink));
  }
  for (int64_t n : {8192, 16384, 24576}) {
    const int64_t nb = n / RB;
    double *A, *v, *R, *P, *sink;
    CK(hipMalloc(&A, 8 * n * n)); CK(hipMalloc(&v, 8 * n)); CK(hipMalloc(&R, 8 * n)); CK(hipMalloc(&P, 8 * nb * n)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(A, 0, 8 * n * n)); CK(hipMemset(v, 0, 8 * n));
    const double tri = 4.0 * n * n;
    for (int rep = 0; rep < 2; ++rep) {
      double us = time_us([&] { hipLaunchKernelGGL((herm_rows<true, false>), dim3((nb + 1) / 2), dim3(PB), 0, 0, A, n, v, n, R, P, (int)nb, sink); }, 10);
      printf("n = %6lld row bands, loads only          : %8.1f us  %.3f of peak on 4n^2 B\n", (long long)n, us, tri / us / 1e6 / 8.0);
      us = time_us([&] { hipLaunchKernelGGL((herm_rows<true, true>), dim3((nb + 1) / 2), dim3(PB), 0, 0, A, n, v, n, R, P, (int)nb, sink); }, 10);
      printf("n = %6lld row bands, full pass (nt loads) : %8.1f us  %.3f\n", (long long)n, us, tri / us / 1e6 / 8.0);
      us = time_us([&] { hipLaunchKernelGGL((herm_rows<false, true>), dim3((nb + 1) / 2), dim3(PB), 0, 0, A, n, v, n, R, P, (int)nb, sink); }, 10);
      printf("n = %6lld row bands, full pass (default)  : %8.1f us  %.3f\n", (long long)n, us, tri / us / 1e6 / 8.0);
    }
    CK(hipFree(A)); CK(hipFree(v)); CK(hipFree(R)); CK(hipFree(P)); CK(hipFree(sink));
  }
  return 0;
}
