// tools/tune_gemvb.hip — experiment harness (NOT the library): the row-band block apply M*V of dense.hip
// (gemvb_n_rows_kernel, round 6) taken apart. A 512-thread workgroup owns RB = 64 rows of a column-major f64 matrix across all
// columns, LPR = 32 lanes per column piece (512 bytes), KB accumulators per row and lane. Variants (MODE):
//   0  V from LDS (chunks of 512 columns, double-buffered, one barrier per chunk), TWO register sets of U loads (prefetch)
//   1  V from LDS as 0, ONE register set (load U, consume U)
//   2  V from global memory per lane (no LDS, no barrier), one register set — gemv_n_rows_kernel with KB accumulators
//   3  as 0 but the staging is done once (chunk 0 only) and no barrier in the loop: WRONG results, isolates barrier + staging cost
//   4  as 2 but V read through scalar-uniform... (not applicable: two columns per wave) -> V[j] broadcast from lane reads (readlane)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/tune_gemvb tools/tune_gemvb.hip ; run: tools/tune_gemvb [m n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double VV __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KB, int U, int MODE>
__global__ void __launch_bounds__(512)
gemvb(double *__restrict__ res, int64_t ldr, const double *__restrict__ M, int64_t m, int64_t n, int64_t ld,
      const double *__restrict__ V, int64_t ldv) {
  constexpr int VR = 2, RB = 64, LPR = RB / VR, NCL = 512 / LPR, STEP = NCL * U, ITS = 512 / STEP, CH = 512;
  constexpr int SL = KB * CH / 512;
  __shared__ double vlds[2 * KB * CH > NCL * RB ? 2 * KB * CH : NCL * RB];
  const int tid = threadIdx.x, seg = tid % LPR, cl = tid / LPR;
  const int64_t row = (int64_t)blockIdx.x * RB + (int64_t)seg * VR;
  const double *base = M + (row < m ? row : 0);
  double acc[KB][VR];
#pragma unroll
  for (int c = 0; c < KB; ++c) acc[c][0] = acc[c][1] = 0.0;
  const int64_t nmain = n / STEP, nfull = nmain / ITS;
  double sreg[SL];
  auto stage_load = [&](int64_t ch) {
#pragma unroll
    for (int k = 0; k < SL; ++k) {
      const int idx = tid + k * 512, c = idx / CH, jl = idx % CH;
      const int64_t j = ch * CH + jl;
      const double x = V[(j < n ? j : n - 1) + (int64_t)c * ldv];
      sreg[k] = j < n ? x : 0.0;
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int k = 0; k < SL; ++k) vlds[(size_t)buf * KB * CH + tid + k * 512] = sreg[k];
  };
  auto load_step = [&](VV (&a)[U], int64_t it) {
    const int64_t j = it * STEP + cl;
#pragma unroll
    for (int u = 0; u < U; ++u)
      a[u] = __builtin_nontemporal_load(reinterpret_cast<const VV *>(base + (j + (int64_t)u * NCL) * ld));
  };
  auto consume = [&](const VV (&a)[U], int64_t it) {
    if constexpr (MODE == 2) {
      const int64_t j = it * STEP + cl;
      double xv[U][KB];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < KB; ++c) xv[u][c] = V[j + (int64_t)u * NCL + (int64_t)c * ldv];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < KB; ++c) {
          acc[c][0] = fma(a[u][0], xv[u][c], acc[c][0]);
          acc[c][1] = fma(a[u][1], xv[u][c], acc[c][1]);
        }
    } else {
      const double *x = vlds + (size_t)((MODE == 3 ? 0 : (it / ITS)) & 1) * KB * CH + (it % ITS) * STEP + cl;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < KB; ++c) {
          const double xv = x[c * CH + u * NCL];
          acc[c][0] = fma(a[u][0], xv, acc[c][0]);
          acc[c][1] = fma(a[u][1], xv, acc[c][1]);
        }
    }
  };
  if constexpr (MODE != 2) {
    stage_load(0);
    stage_store(0);
    __syncthreads();
  }
  if (nmain > 0) {
    if constexpr (MODE == 0 || MODE == 3) {
      VV a0[U], a1[U];
      load_step(a0, 0);
      __builtin_amdgcn_sched_barrier(0);
      for (int64_t ch = 0; ch < nfull; ++ch) {
        if constexpr (MODE == 0) stage_load(ch + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sidx = 0; sidx < ITS; sidx += 2) {
          const int64_t it = ch * ITS + sidx;
          load_step(a1, it + 1);
          __builtin_amdgcn_sched_barrier(0);
          consume(a0, it);
          __builtin_amdgcn_sched_barrier(0);
          load_step(a0, it + 2 < nmain ? it + 2 : nmain - 1);
          __builtin_amdgcn_sched_barrier(0);
          consume(a1, it + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 0) {
          stage_store((int)((ch + 1) & 1));
          __syncthreads();
        }
      }
    } else {
      VV a0[U];
      for (int64_t ch = 0; ch < nfull; ++ch) {
        if constexpr (MODE == 1) stage_load(ch + 1);
#pragma unroll
        for (int sidx = 0; sidx < ITS; ++sidx) {
          const int64_t it = ch * ITS + sidx;
          load_step(a0, it);
          consume(a0, it);
        }
        if constexpr (MODE == 1) {
          stage_store((int)((ch + 1) & 1));
          __syncthreads();
        }
      }
    }
  }
  double(*sred)[RB] = reinterpret_cast<double(*)[RB]>(vlds);
  const int r = tid % RB, q = tid / RB;
#pragma unroll
  for (int c = 0; c < KB; ++c) {
    __syncthreads();
    sred[cl][seg * VR] = acc[c][0];
    sred[cl][seg * VR + 1] = acc[c][1];
    __syncthreads();
    double part = 0.0;
    for (int cc = q; cc < NCL; cc += 8) part += sred[cc][r];
    __syncthreads();
    sred[q][r] = part;
    __syncthreads();
    if (tid < RB) {
      const int64_t i = (int64_t)blockIdx.x * RB + tid;
      if (i < m) {
        double s = 0.0;
        for (int qq = 0; qq < 8; ++qq) s += sred[qq][tid];
        res[i + (int64_t)c * ldr] = s;
      }
    }
  }
}

template <int KB, int U, int MODE>
void run(const char *name, double *res, const double *M, int64_t m, int64_t n, const double *V, const std::vector<double> &ref) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  dim3 grid((unsigned)((m + 63) / 64));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gemvb<KB, U, MODE>), grid, dim3(512), 0, 0, res, m, M, m, n, m, V, n);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((gemvb<KB, U, MODE>), grid, dim3(512), 0, 0, res, m, M, m, n, m, V, n);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<double> h((size_t)m);
  CK(hipMemcpy(h.data(), res, sizeof(double) * m, hipMemcpyDeviceToHost));
  double err = 0, nr = 0;
  for (int64_t i = 0; i < m; ++i) { err += (h[i] - ref[i]) * (h[i] - ref[i]); nr += ref[i] * ref[i]; }
  const double us = ms / reps * 1e3;
  printf("KB=%d U=%d mode %d %-42s %8.1f us  %6.0f GB/s of M  (%.3f)  rel err col0 %.1e\n", KB, U, MODE, name, us, 8.0 * m * n / us / 1e3,
         8.0 * m * n / us / 1e3 / 8000, nr > 0 ? sqrt(err / nr) : 0.0);
}

int main(int argc, char **argv) {
  const int64_t m = argc > 2 ? atoll(argv[1]) : 16384, n = argc > 2 ? atoll(argv[2]) : 16384;
  std::vector<double> hM((size_t)m * n), hV((size_t)n * 8), ref((size_t)m, 0.0);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
  for (auto &x : hM) x = rnd();
  for (auto &x : hV) x = rnd();
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i) ref[i] += hM[i + j * m] * hV[j];
  double *M, *V, *res;
  CK(hipMalloc(&M, sizeof(double) * m * n));
  CK(hipMalloc(&V, sizeof(double) * n * 8));
  CK(hipMalloc(&res, sizeof(double) * m * 8));
  CK(hipMemcpy(M, hM.data(), sizeof(double) * m * n, hipMemcpyHostToDevice));
  CK(hipMemcpy(V, hV.data(), sizeof(double) * n * 8, hipMemcpyHostToDevice));
  printf("# m = %lld, n = %lld (f64, %.1f MB)\n", (long long)m, (long long)n, 8.0 * m * n / 1e6);
  run<2, 8, 0>("LDS + two sets + barrier (library)", res, M, m, n, V, ref);
  run<2, 8, 1>("LDS + one set + barrier", res, M, m, n, V, ref);
  run<2, 8, 2>("V from global, one set, no barrier", res, M, m, n, V, ref);
  run<2, 8, 3>("two sets, no restaging, no barrier (wrong)", res, M, m, n, V, ref);
  run<2, 4, 0>("LDS + two sets + barrier", res, M, m, n, V, ref);
  run<4, 8, 0>("LDS + two sets + barrier (library)", res, M, m, n, V, ref);
  run<4, 8, 1>("LDS + one set + barrier", res, M, m, n, V, ref);
  run<4, 8, 2>("V from global, one set, no barrier", res, M, m, n, V, ref);
  run<8, 4, 0>("LDS + two sets + barrier (library)", res, M, m, n, V, ref);
  run<8, 4, 1>("LDS + one set + barrier", res, M, m, n, V, ref);
  run<8, 8, 1>("LDS + one set + barrier", res, M, m, n, V, ref);
  run<8, 4, 2>("V from global, one set, no barrier", res, M, m, n, V, ref);
  run<8, 4, 3>("two sets, no restaging, no barrier (wrong)", res, M, m, n, V, ref);
  return 0;
}
