// tune_stream.hip — standalone sweep of streaming-kernel launch geometry on MI355X.
// Not part of libmxlo.so: run on the GPU box to pick UNROLL / blocks-per-CU / nontemporal policy
// for the HBM-bound leaves (diag: 2 reads + 1 write; dot: 2 reads; householder update: 2r+1w).
//   usage: tune_stream [n=100000000] [iters=10]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double f64x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e = (x);                                                                \
    if (e != hipSuccess) {                                                             \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);     \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

template <bool NT>
__device__ __forceinline__ f64x2 ld(const f64x2 *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(f64x2 *p, f64x2 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// res = (a*d)*v : the opDiagonal beta==0 body
template <int BLOCK, int UNROLL, bool NTL, bool NTS>
__global__ void __launch_bounds__(BLOCK)
diag_kernel(f64x2 *__restrict__ res, const f64x2 *__restrict__ d, const f64x2 *__restrict__ v,
            int64_t nvec, double a) {
  constexpr int64_t CHUNK = (int64_t)BLOCK * UNROLL;
  const int64_t nchunks = (nvec + CHUNK - 1) / CHUNK;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t base = ch * CHUNK + threadIdx.x;
    f64x2 x[UNROLL], y[UNROLL];
    if (base + (int64_t)(UNROLL - 1) * BLOCK < nvec) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        x[u] = ld<NTL>(d + base + (int64_t)u * BLOCK);
        y[u] = ld<NTL>(v + base + (int64_t)u * BLOCK);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        f64x2 o;
        o[0] = (a * x[u][0]) * y[u][0];
        o[1] = (a * x[u][1]) * y[u][1];
        st<NTS>(res + base + (int64_t)u * BLOCK, o);
      }
    } else {
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * BLOCK;
        if (i < nvec) {
          f64x2 xx = d[i], yy = v[i], o;
          o[0] = (a * xx[0]) * yy[0];
          o[1] = (a * xx[1]) * yy[1];
          res[i] = o;
        }
      }
    }
  }
}

template <int BLOCK, int UNROLL, bool NTL>
__global__ void __launch_bounds__(BLOCK)
dot_kernel(const f64x2 *__restrict__ a, const f64x2 *__restrict__ b, int64_t nvec,
           double *__restrict__ partials) {
  constexpr int64_t CHUNK = (int64_t)BLOCK * UNROLL;
  const int64_t nchunks = (nvec + CHUNK - 1) / CHUNK;
  double acc0 = 0, acc1 = 0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t base = ch * CHUNK + threadIdx.x;
    f64x2 x[UNROLL], y[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = base + (int64_t)u * BLOCK;
      if (i < nvec) {
        x[u] = ld<NTL>(a + i);
        y[u] = ld<NTL>(b + i);
      } else {
        x[u] = f64x2{0, 0};
        y[u] = f64x2{0, 0};
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      acc0 = fma(x[u][0], y[u][0], acc0);
      acc1 = fma(x[u][1], y[u][1], acc1);
    }
  }
  double s = acc0 + acc1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ double lds[BLOCK / 64];
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < BLOCK / 64; ++w) t += lds[w];
    partials[blockIdx.x] = t;
  }
}

struct Bufs {
  f64x2 *res, *d, *v;
  double *partials;
  int64_t nvec;
  int iters;
  int ncu;
};

template <typename L>
static double time_ms(const Bufs &b, L &&launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < b.iters; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms / b.iters;
}

template <int BLOCK, int UNROLL, bool NTL, bool NTS>
static void run_diag(const Bufs &b, int per_cu) {
  const int64_t CHUNK = (int64_t)BLOCK * UNROLL;
  int64_t need = (b.nvec + CHUNK - 1) / CHUNK;
  int64_t grid = per_cu > 0 ? std::min<int64_t>(need, (int64_t)b.ncu * per_cu) : need;
  double ms = time_ms(b, [&] {
    hipLaunchKernelGGL((diag_kernel<BLOCK, UNROLL, NTL, NTS>), dim3((unsigned)grid), dim3(BLOCK), 0,
                       0, b.res, b.d, b.v, b.nvec, 1.25);
  });
  double gb = 24.0 * 2 * b.nvec / 1e9;
  printf("diag  block=%4d unroll=%d ntl=%d nts=%d per_cu=%3d grid=%8lld  %.4f ms  %8.1f GB/s\n",
         BLOCK, UNROLL, (int)NTL, (int)NTS, per_cu, (long long)grid, ms, gb / (ms * 1e-3));
  fflush(stdout);
}

template <int BLOCK, int UNROLL, bool NTL>
static void run_dot(const Bufs &b, int per_cu) {
  const int64_t CHUNK = (int64_t)BLOCK * UNROLL;
  int64_t need = (b.nvec + CHUNK - 1) / CHUNK;
  int64_t grid = std::min<int64_t>(need, (int64_t)b.ncu * per_cu);
  double ms = time_ms(b, [&] {
    hipLaunchKernelGGL((dot_kernel<BLOCK, UNROLL, NTL>), dim3((unsigned)grid), dim3(BLOCK), 0, 0,
                       b.d, b.v, b.nvec, b.partials);
  });
  double gb = 16.0 * 2 * b.nvec / 1e9;
  printf("dot   block=%4d unroll=%d ntl=%d       per_cu=%3d grid=%8lld  %.4f ms  %8.1f GB/s\n",
         BLOCK, UNROLL, (int)NTL, per_cu, (long long)grid, ms, gb / (ms * 1e-3));
  fflush(stdout);
}

int main(int argc, char **argv) {
  int64_t n = argc > 1 ? atoll(argv[1]) : 100000000LL;
  int iters = argc > 2 ? atoi(argv[2]) : 10;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  n=%lld\n", prop.name, prop.multiProcessorCount, (long long)n);
  Bufs b;
  b.nvec = n / 2;
  b.iters = iters;
  b.ncu = prop.multiProcessorCount;
  CK(hipMalloc((void **)&b.res, n * 8));
  CK(hipMalloc((void **)&b.d, n * 8));
  CK(hipMalloc((void **)&b.v, n * 8));
  CK(hipMalloc((void **)&b.partials, 8 * 1 << 20));
  {  // non-trivial data (DVFS: never bench on zeros)
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5 + (double)((i * 2654435761u) % 1000) / 1000.0;
    for (int64_t off = 0; off < n; off += (int64_t)h.size()) {
      int64_t cnt = std::min<int64_t>(h.size(), n - off);
      CK(hipMemcpy((double *)b.d + off, h.data(), cnt * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy((double *)b.v + off, h.data(), cnt * 8, hipMemcpyHostToDevice));
    }
  }
  // reference: hipMemcpy D2D (1 read + 1 write)
  {
    double ms = time_ms(b, [&] { CK(hipMemcpyAsync(b.res, b.d, n * 8, hipMemcpyDeviceToDevice, 0)); });
    printf("memcpyD2D  %.4f ms  %8.1f GB/s (16 B/elt)\n", ms, 16.0 * n / 1e9 / (ms * 1e-3));
  }
  const int pcs[] = {2, 4, 8, 16, 32, 0};
  for (int pc : pcs) run_diag<256, 1, false, false>(b, pc);
  for (int pc : pcs) run_diag<256, 2, false, false>(b, pc);
  for (int pc : pcs) run_diag<256, 4, false, false>(b, pc);
  for (int pc : pcs) run_diag<256, 8, false, false>(b, pc);
  for (int pc : {4, 8, 16, 0}) run_diag<256, 4, true, false>(b, pc);
  for (int pc : {4, 8, 16, 0}) run_diag<256, 4, false, true>(b, pc);
  for (int pc : {4, 8, 16, 0}) run_diag<256, 4, true, true>(b, pc);
  for (int pc : {2, 4, 8, 0}) run_diag<512, 2, false, false>(b, pc);
  for (int pc : {2, 4, 8, 0}) run_diag<512, 4, true, true>(b, pc);
  for (int pc : {1, 2, 4, 0}) run_diag<1024, 2, false, false>(b, pc);
  for (int pc : {1, 2, 4, 0}) run_diag<1024, 2, true, true>(b, pc);
  const int pcd[] = {2, 4, 8, 16};
  for (int pc : pcd) run_dot<256, 2, false>(b, pc);
  for (int pc : pcd) run_dot<256, 4, false>(b, pc);
  for (int pc : pcd) run_dot<256, 8, false>(b, pc);
  for (int pc : pcd) run_dot<256, 4, true>(b, pc);
  for (int pc : pcd) run_dot<256, 8, true>(b, pc);
  for (int pc : {1, 2, 4, 8}) run_dot<512, 4, false>(b, pc);
  for (int pc : {1, 2, 4}) run_dot<1024, 4, false>(b, pc);
  for (int pc : {1, 2, 4}) run_dot<1024, 4, true>(b, pc);
  return 0;
}
