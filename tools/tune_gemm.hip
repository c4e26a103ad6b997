// tools/tune_gemm.hip — experiment harness (NOT product): sweeps the wave layout / K-slab / ring depth of
// csrc/gemm_glds.h on the two chained GEMMs of kron(A,B) (src/kron.jl:14-22), next to the library's current
// kron path and the vendor GEMM on the same box.
//   build: hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=off tools/tune_gemm.hip -o tools/tune_gemm
//                -Ilinearoperators.jl_amd/csrc -Llinearoperators.jl_amd/csrc -lmxlo -L/opt/rocm/lib -lrocblas
//                -Wl,-rpath,$PWD/linearoperators.jl_amd/csrc
//   run  : tools/tune_gemm [n ...]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "gemm_glds.h"

using namespace mxlo;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

template <typename T>
__global__ void ref_gemm_kernel(double *C, const T *A, int64_t lda, bool ak, const T *B, int64_t ldb, int M, int N,
                                int K) {
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= M || j >= N) return;
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)(ak ? A[k + (int64_t)i * lda] : A[i + (int64_t)k * lda]) * (double)B[j + (int64_t)k * ldb];
  C[i + (int64_t)j * M] = s;
}

template <typename T>
__global__ void diff_kernel(const T *C, const double *R, int64_t n, double *out) {
  double m = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) m = fmax(m, fabs((double)C[i] - R[i]));
  __shared__ double sm[256];
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

template <typename T, bool AK, int TM, int TN, int WM, int WN, int BK, int NST, bool SP = true, bool PIN = true, bool PAIR = false, int PFD = 1, bool SWAPC = false, bool NTC = false, bool XNOBAR = false, bool UNR = false>
void launch(hipStream_t st, T *C, int64_t ldc, const T *A, int64_t lda, const T *B, int64_t ldb, int M, int N, int K) {
  GlShape S{M, N, K, (M + TM - 1) / TM, (N + TN - 1) / TN};
  hipLaunchKernelGGL((gemm_glds_kernel<T, T, T, true, AK, TM, TN, WM, WN, BK, NST, SP, PIN, PAIR, PFD, SWAPC, NTC, XNOBAR, UNR>), dim3(S.gx * S.gy), dim3(WM * WN * 64),
                     0, st, C, ldc, A, lda, B, ldb, S, (T)1, (T)0);
}

static std::vector<std::string> g_only;   // --only <substring>: run matching variants only (profiling)
static int g_reps = 0;                    // --reps N

template <typename T>
struct Bench {
  int n;
  T *A, *B, *X, *W, *R;
  double *ref, *refT, *dmax;
  hipStream_t st;
  hipEvent_t e0, e1;
  Bench(int n_) : n(n_) {
    const size_t bytes = sizeof(T) * (size_t)n * n;
    CK(hipMalloc(&A, bytes));
    CK(hipMalloc(&B, bytes));
    CK(hipMalloc(&X, bytes));
    CK(hipMalloc(&W, bytes));
    CK(hipMalloc(&R, bytes));
    CK(hipMalloc(&ref, sizeof(double) * (size_t)n * n));
    CK(hipMalloc(&refT, sizeof(double) * (size_t)n * n));
    CK(hipMalloc(&dmax, 8));
    std::vector<T> h((size_t)n * n);
    for (T **p : {&A, &B, &X}) {
      for (auto &v : h) v = (T)((rand() / (double)RAND_MAX * 2 - 1) / 8);
      CK(hipMemcpy(*p, h.data(), bytes, hipMemcpyHostToDevice));
    }
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
  }
  ~Bench() {
    for (void *p : {(void *)A, (void *)B, (void *)X, (void *)W, (void *)R, (void *)ref, (void *)refT, (void *)dmax}) (void)hipFree(p);
  }
  // times fn (which enqueues the two chained GEMMs W = A*X^T, R = B*W^T on st); checks R's first GEMM against ref
  template <typename F>
  void run(const char *name, F &&fn, int check = 1) {   // check: 0 none, 1 vs A*X', 2 vs A'*X'
    if (!g_only.empty()) {
      bool hit = false;
      for (auto &o : g_only) hit = hit || std::string(name).find(o) != std::string::npos;
      if (!hit) return;
    }
    CK(hipMemsetAsync(W, 0xff, sizeof(T) * (size_t)n * n, st));
    fn();
    CK(hipStreamSynchronize(st));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
      printf("%-44s n=%5d  LAUNCH ERROR %s\n", name, n, hipGetErrorString(le));
      return;
    }
    double err = -1;
    if (check) {
      hipLaunchKernelGGL(diff_kernel<T>, dim3(1), dim3(256), 0, st, (const T *)W, (const double *)(check == 2 ? refT : ref), (int64_t)n * n, dmax);
      CK(hipMemcpy(&err, dmax, 8, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < 5; ++i) fn();
    const int reps = g_reps > 0 ? g_reps : (n <= 1024 ? 200 : 50);
    float best = 1e30f, tot = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) fn();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      tot += ms;
    }
    const double us = best * 1e3 / reps, tf = 4.0 * n * (double)n * n / us / 1e6;
    printf("%-44s n=%5d %9.1f us/kron %7.1f us/GEMM %7.1f TF  maxerr %.2e\n", name, n, us, us / 2, tf, err);
    fflush(stdout);
  }
};

extern "C" {
#include "../include/mxlo.h"
}

template <typename T>
void sweep(int n, rocblas_handle rb) {
  Bench<T> b(n);
  constexpr bool F64 = sizeof(T) == 8;
  const char *tn = F64 ? "f64" : "f32";
  hipLaunchKernelGGL(ref_gemm_kernel<T>, dim3((n + 15) / 16, (n + 15) / 16), dim3(256), 0, b.st, b.ref, b.A, (int64_t)n, false,
                     b.X, (int64_t)n, n, n, n);
  hipLaunchKernelGGL(ref_gemm_kernel<T>, dim3((n + 15) / 16, (n + 15) / 16), dim3(256), 0, b.st, b.refT, b.A, (int64_t)n, true,
                     b.X, (int64_t)n, n, n, n);
  CK(hipStreamSynchronize(b.st));
  char name[128];
  // vendor
  {
    CK(hipStreamSynchronize(b.st));
    rocblas_set_stream(rb, b.st);
    auto f = [&] {
      if constexpr (F64) {
        const double one = 1, zero = 0;
        rocblas_dgemm(rb, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &one, b.A, n, b.X, n, &zero, b.W, n);
        rocblas_dgemm(rb, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &one, b.B, n, b.W, n, &zero, b.R, n);
      } else {
        const float one = 1, zero = 0;
        rocblas_sgemm(rb, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &one, (const float *)b.A, n,
                      (const float *)b.X, n, &zero, (float *)b.W, n);
        rocblas_sgemm(rb, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &one, (const float *)b.B, n,
                      (const float *)b.W, n, &zero, (float *)b.R, n);
      }
    };
    snprintf(name, sizeof name, "%s vendor rocBLAS gemm x2", tn);
    b.run(name, f);
  }
  // library (current kron path)
  {
    mxlo_ctx *ctx = nullptr;
    if (mxlo_ctx_create(0, b.st, &ctx) == 0) {
      auto f = [&] {
        mxlo_kron_mul(ctx, F64 ? MXLO_F64 : MXLO_F32, b.R, b.A, n, n, n, b.B, n, n, n, b.X, b.W, 1.0, 0.0, MXLO_OP_N, 0);
      };
      snprintf(name, sizeof name, "%s libmxlo mxlo_kron_mul (N)", tn);
      b.run(name, f);
      auto ft = [&] {
        mxlo_kron_mul(ctx, F64 ? MXLO_F64 : MXLO_F32, b.R, b.A, n, n, n, b.B, n, n, n, b.X, b.W, 1.0, 0.0, MXLO_OP_T, 0);
      };
      snprintf(name, sizeof name, "%s libmxlo mxlo_kron_mul (T)", tn);
      b.run(name, ft, 2);
      mxlo_ctx_destroy(ctx);
    }
  }
#define V(AK_, TM_, TN_, WM_, WN_, BK_, NST_) VS(AK_, TM_, TN_, WM_, WN_, BK_, NST_, true, true)
#define VS(AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_) VP(AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, false)
#define VQ(AK_, TM_, TN_, WM_, WN_, BK_, NST_) VP(AK_, TM_, TN_, WM_, WN_, BK_, NST_, true, true, true)
#define VP(AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, PAIR_) VF(AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, PAIR_, 1)
#define VF(AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, PAIR_, PFD_) VW(AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, PAIR_, PFD_, false)
#define VW(AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, PAIR_, PFD_, SW_) VN(AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, PAIR_, PFD_, SW_, false)
#define VN(AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, PAIR_, PFD_, SW_, NT_)                                     \
  {                                                                                                                 \
    auto f = [&] {                                                                                                  \
      launch<T, AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, PAIR_, PFD_, SW_, NT_>(b.st, b.W, n, b.A, n, b.X, n, n, n, n); \
      launch<T, AK_, TM_, TN_, WM_, WN_, BK_, NST_, SP_, PIN_, PAIR_, PFD_, SW_, NT_>(b.st, b.R, n, b.B, n, b.W, n, n, n, n); \
    };                                                                                                              \
    snprintf(name, sizeof name, "%s glds %s t%dx%d w%dx%d bk%d st%d%s%s%s%s%s%s", tn, AK_ ? "AK" : "AM", TM_, TN_, WM_, WN_, BK_, \
             NST_, SP_ ? "" : " nospread", PIN_ ? "" : " nopin", PAIR_ ? " PAIR" : "", PFD_ == 2 ? " PFD2" : "", SW_ ? " SWAPC" : "", NT_ ? " NTC" : ""); \
    b.run(name, f, AK_ ? 2 : 1);                                                                                    \
  }
  if constexpr (F64) {
#define VU(AK_, TM_, TN_, WM_, WN_, BK_, NST_, PAIR_, PFD_, TAG_)                                                    \
  {                                                                                                                 \
    auto f = [&] {                                                                                                  \
      launch<T, AK_, TM_, TN_, WM_, WN_, BK_, NST_, true, true, PAIR_, PFD_, true, false, false, true>(b.st, b.W, n, b.A, n, b.X, n, n, n, n); \
      launch<T, AK_, TM_, TN_, WM_, WN_, BK_, NST_, true, true, PAIR_, PFD_, true, false, false, true>(b.st, b.R, n, b.B, n, b.W, n, n, n, n); \
    };                                                                                                              \
    snprintf(name, sizeof name, "%s glds %s t%dx%d w%dx%d bk%d st%d %s SWAPC UNR", tn, AK_ ? "AK" : "AM", TM_, TN_, WM_, WN_, BK_, NST_, TAG_); \
    b.run(name, f, AK_ ? 2 : 1);                                                                                    \
  }
    VU(false, 64, 64, 4, 2, 32, 3, true, 1, "PAIR")
    VU(true, 64, 64, 4, 2, 32, 3, true, 1, "PAIR")
    VU(false, 128, 128, 4, 4, 16, 3, true, 1, "PAIR")
    VU(false, 32, 32, 2, 2, 32, 4, false, 2, "PFD2")
#undef VU
    {  // timing experiment: the per-slab barrier removed (results wrong): what the barrier costs
      auto f = [&] {
        launch<T, false, 64, 64, 4, 2, 32, 3, true, true, true, 1, true, false, true>(b.st, b.W, n, b.A, n, b.X, n, n, n, n);
        launch<T, false, 64, 64, 4, 2, 32, 3, true, true, true, 1, true, false, true>(b.st, b.R, n, b.B, n, b.W, n, n, n, n);
      };
      snprintf(name, sizeof name, "%s glds AM t64x64 w4x2 bk32 st3 PAIR SWAPC XNOBAR(wrong results)", tn);
      b.run(name, f, 0);
      auto f2 = [&] {
        launch<T, false, 32, 32, 2, 2, 32, 4, true, true, false, 2, true, false, true>(b.st, b.W, n, b.A, n, b.X, n, n, n, n);
        launch<T, false, 32, 32, 2, 2, 32, 4, true, true, false, 2, true, false, true>(b.st, b.R, n, b.B, n, b.W, n, n, n, n);
      };
      snprintf(name, sizeof name, "%s glds AM t32x32 w2x2 bk32 st4 PFD2 SWAPC XNOBAR(wrong results)", tn);
      b.run(name, f2, 0);
    }
    VN(false, 64, 64, 4, 2, 32, 3, true, true, true, 1, true, true)
    VN(true, 64, 64, 4, 2, 32, 3, true, true, true, 1, true, true)
    VN(false, 128, 128, 4, 4, 16, 3, true, true, true, 1, true, true)
    VN(false, 32, 32, 2, 2, 32, 4, true, true, false, 2, true, true)
    VW(false, 64, 64, 4, 2, 32, 3, true, true, true, 1, true)
    VW(true, 64, 64, 4, 2, 32, 3, true, true, true, 1, true)
    VW(false, 128, 128, 4, 4, 16, 3, true, true, true, 1, true)
    VW(false, 32, 32, 2, 2, 32, 4, true, true, false, 2, true)
    VW(true, 32, 32, 2, 2, 32, 4, true, true, false, 2, true)
    VF(false, 32, 32, 2, 2, 32, 4, true, true, false, 2)
    VF(true, 32, 32, 2, 2, 32, 4, true, true, false, 2)
    VF(false, 64, 64, 4, 2, 32, 3, true, true, true, 2)
    VF(true, 64, 64, 4, 2, 32, 3, true, true, true, 2)
    VF(false, 128, 128, 4, 4, 16, 3, true, true, true, 2)
    VF(false, 64, 64, 2, 2, 32, 3, true, true, true, 2)
    VQ(false, 64, 64, 4, 2, 32, 3)
    VQ(false, 64, 64, 2, 4, 32, 3)
    VQ(false, 64, 64, 2, 2, 32, 3)
    VQ(false, 64, 64, 2, 2, 32, 4)
    VQ(false, 128, 128, 4, 4, 16, 3)
    VQ(false, 128, 64, 4, 2, 32, 3)
    VQ(true, 64, 64, 4, 2, 32, 3)
    VQ(false, 32, 32, 1, 1, 32, 4)
    VQ(false, 64, 64, 4, 2, 48, 3)
    VQ(false, 64, 64, 2, 4, 48, 3)
    VQ(false, 64, 64, 4, 2, 16, 4)
    VQ(false, 64, 32, 2, 2, 32, 3)
    VQ(false, 64, 32, 2, 2, 64, 3)
    VQ(false, 32, 64, 2, 2, 32, 3)
    VQ(false, 32, 64, 2, 2, 64, 3)
    VQ(true, 64, 64, 4, 2, 48, 3)
    V(false, 64, 64, 2, 2, 32, 3)
    V(false, 64, 64, 2, 2, 32, 4)
    V(false, 64, 64, 2, 2, 16, 4)
    V(false, 64, 64, 4, 2, 32, 3)
    V(false, 64, 64, 4, 2, 32, 4)
    V(false, 64, 64, 2, 4, 32, 3)
    V(false, 64, 64, 4, 4, 32, 3)
    V(false, 64, 64, 4, 4, 32, 4)
    V(false, 64, 64, 4, 4, 48, 3)
    V(false, 64, 64, 4, 4, 16, 4)
    VS(false, 64, 64, 4, 4, 32, 3, false, true)
    VS(false, 64, 64, 4, 4, 32, 3, true, false)
    VS(false, 64, 64, 2, 2, 32, 3, true, false)
    V(false, 64, 32, 2, 2, 32, 3)
    V(false, 64, 32, 4, 2, 32, 3)
    V(false, 32, 64, 2, 4, 32, 3)
    V(false, 32, 32, 2, 2, 32, 4)
    V(false, 128, 64, 4, 2, 32, 3)
    V(false, 128, 64, 4, 4, 32, 3)
    V(false, 128, 128, 4, 4, 16, 3)
    V(false, 128, 128, 4, 2, 16, 3)
    V(false, 128, 128, 4, 4, 16, 4)
    V(true, 64, 64, 2, 2, 32, 3)
    V(true, 64, 64, 4, 4, 32, 3)
    V(true, 64, 64, 4, 2, 32, 3)
    V(true, 32, 32, 2, 2, 32, 4)
  } else {
    VN(false, 64, 64, 4, 2, 64, 3, true, true, true, 1, true, true)
    VN(false, 128, 128, 4, 4, 32, 3, true, true, false, 1, true, true)
    VN(false, 32, 32, 2, 2, 64, 4, true, true, false, 2, true, true)
    VW(false, 64, 64, 4, 2, 64, 3, true, true, true, 1, true)
    VW(true, 64, 64, 4, 2, 64, 3, true, true, true, 1, true)
    VW(false, 128, 128, 4, 4, 32, 3, true, true, false, 1, true)
    VW(false, 32, 32, 2, 2, 64, 4, true, true, false, 2, true)
    VF(false, 32, 32, 2, 2, 64, 4, true, true, false, 2)
    VF(true, 32, 32, 2, 2, 64, 4, true, true, false, 2)
    VF(false, 64, 64, 4, 2, 64, 3, true, true, true, 2)
    VF(false, 128, 128, 4, 4, 32, 3, true, true, false, 2)
    VQ(false, 64, 64, 4, 2, 64, 3)
    VQ(false, 128, 128, 4, 4, 32, 3)
    VQ(false, 32, 32, 2, 2, 64, 4)
    VQ(true, 64, 64, 4, 2, 64, 3)
    VQ(false, 64, 64, 4, 2, 96, 3)
    VQ(false, 64, 64, 2, 4, 64, 3)
    VQ(false, 64, 64, 2, 4, 96, 3)
    VQ(true, 64, 64, 4, 2, 96, 3)
    V(false, 64, 64, 2, 2, 32, 3)
    V(false, 64, 64, 2, 2, 64, 3)
    V(false, 64, 64, 4, 2, 64, 3)
    V(false, 64, 64, 4, 2, 32, 4)
    V(false, 64, 64, 4, 4, 64, 3)
    V(false, 128, 64, 4, 2, 32, 3)
    V(false, 128, 128, 4, 2, 32, 3)
    V(false, 128, 128, 4, 4, 32, 3)
    V(false, 64, 32, 2, 2, 64, 3)
    V(false, 32, 32, 2, 2, 64, 4)
    V(true, 64, 64, 4, 2, 32, 3)
    V(true, 64, 64, 4, 2, 64, 3)
    V(true, 64, 64, 2, 2, 64, 3)
  }
#undef V
#undef VS
#undef VQ
#undef VP
#undef VF
#undef VW
#undef VN
}

// time-vs-K at fixed M = N: slope = per-slab cost, intercept = launch + prologue + epilogue
template <typename T, int TM, int TN, int WM, int WN, int BK, int NST>
void ksweep(int n) {
  const int kmax = 4096;
  T *A, *B, *C;
  CK(hipMalloc(&A, sizeof(T) * (size_t)n * kmax));
  CK(hipMalloc(&B, sizeof(T) * (size_t)n * kmax));
  CK(hipMalloc(&C, sizeof(T) * (size_t)n * n));
  CK(hipMemset(A, 0, sizeof(T) * (size_t)n * kmax));
  CK(hipMemset(B, 0, sizeof(T) * (size_t)n * kmax));
  hipStream_t st;
  hipEvent_t e0, e1;
  CK(hipStreamCreate(&st));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int K : {32, 64, 128, 256, 512, 1024, 2048, 4096}) {
    auto f = [&] { launch<T, false, TM, TN, WM, WN, BK, NST, true, true>(st, C, n, A, n, B, n, n, n, K); };
    for (int i = 0; i < 10; ++i) f();
    const int reps = 200;
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) f();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("ksweep t%dx%d w%dx%d bk%d st%d  M=N=%d K=%5d  %8.2f us/GEMM  %6.1f TF (zero-filled operands)\n", TM, TN, WM, WN, BK,
           NST, n, K, best * 1e3 / reps, 2.0 * n * n * (double)K / (best * 1e3 / reps) / 1e6);
  }
  (void)hipFree(A); (void)hipFree(B); (void)hipFree(C);
}

int main(int argc, char **argv) {
  std::vector<int> ns;
  bool f64 = true, f32 = true;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--only" && i + 1 < argc) g_only.push_back(argv[++i]);
    else if (a == "--reps" && i + 1 < argc) g_reps = atoi(argv[++i]);
    else if (a == "--ksweep") {
      ksweep<double, 64, 64, 4, 4, 32, 3>(1024);
      ksweep<double, 64, 64, 2, 2, 32, 3>(1024);
      ksweep<double, 128, 128, 4, 4, 16, 3>(2048);
      return 0;
    }
    else if (a == "--f64") f32 = false;
    else if (a == "--f32") f64 = false;
    else ns.push_back(atoi(argv[i]));
  }
  if (ns.empty()) ns = {1024, 1000, 512, 2048};
  rocblas_handle rb;
  rocblas_create_handle(&rb);
  for (int n : ns) {
    if (f64) sweep<double>(n, rb);
    if (f32) sweep<float>(n, rb);
  }
  rocblas_destroy_handle(rb);
  return 0;
}
