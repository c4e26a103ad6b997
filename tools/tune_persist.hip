// tools/tune_persist.hip — experiment harness (NOT the library): the cache-resident regime of the quasi-Newton applies
// (n = 2^19 .. 2^22 fp64, 10 .. 40 panel columns: the panel is 40 .. 340 MB, at or below the 256 MiB Infinity Cache).
//
//   A. ceilings:   (1) a read-only sweep over footprints from L2-sized to HBM-sized, steady state (the buffer is re-read
//                      back to back, so whatever fits a cache level is served from it);
//                  (2) the COMBINE-shaped pass (reads ncol columns + x, writes res) timed alone right after a dots-shaped
//                      pass over the same panel — the best the second half of a two-pass apply can do on this chip,
//                      forward and in the reverse of the dots order.
//   B. structure:  the whole apply (dots -> exchange -> coefficients -> combine) as ONE persistent launch, G workgroups
//                  each owning a contiguous run of chunks, for several (block size, workgroups per CU, columns per batch,
//                  vectors per batch) shapes, against the same arithmetic as two grid-wide launches + a finalize launch.
// The coefficient step is a stand-in (coef_c = 1e-3 * dot_c): what is measured is the data movement and the exchange.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/tune_persist tools/tune_persist.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kMaxCols = 40;
constexpr unsigned long long kEmpty = 0x7FF8DEADBEEF0001ull;
struct Cols { const double *p[kMaxCols]; };

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp0(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_allsum(double v) {
  v += dpp0<0xb1, 0xf>(v); v += dpp0<0x4e, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
  v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// ---- A1: read-only sweep -------------------------------------------------------------------------------------------
template <int UNR>
__global__ void __launch_bounds__(256) read_kernel(const f64x2 *__restrict__ p, int64_t nvec, double *out) {
  double s = 0;
  const int64_t stride = (int64_t)gridDim.x * 256 * UNR;
  for (int64_t i = (int64_t)blockIdx.x * 256 * UNR + threadIdx.x; i < nvec; i += stride) {
    f64x2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = i + u * 256 < nvec ? p[i + u * 256] : f64x2{0, 0};
#pragma unroll
    for (int u = 0; u < UNR; ++u) s += v[u][0] + v[u][1];
  }
  if (s == 1.2345e300) out[blockIdx.x] = s;
}

// ---- two-launch reference shape: grid-wide dots (NB columns per pass over x), grid-wide combine ----------------------
template <int NC>
__global__ void __launch_bounds__(256) dots_kernel(Cols cols, int c0, const double *__restrict__ x, int64_t nvec,
                                                   double *__restrict__ partials, int pstride) {
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const f64x2 xv = reinterpret_cast<const f64x2 *>(x)[i];
    f64x2 cv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cv[c] = reinterpret_cast<const f64x2 *>(cols.p[c0 + c])[i];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = fma(cv[c][1], xv[1], fma(cv[c][0], xv[0], acc[c]));
  }
  __shared__ double lds[4][NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const double s = wave_allsum(acc[c]);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6][c] = s;
  }
  __syncthreads();
  if (threadIdx.x < NC)
    partials[(int64_t)(c0 + threadIdx.x) * pstride + blockIdx.x] =
        (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
}
__global__ void __launch_bounds__(256) finalize_kernel(const double *__restrict__ partials, int pstride, int nblocks,
                                                       double *__restrict__ coef) {
  const int c = blockIdx.x;
  double s = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partials[(int64_t)c * pstride + i];
  s = wave_allsum(s);
  __shared__ double l[4];
  if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) coef[c] = 1e-3 * ((l[0] + l[1]) + (l[2] + l[3]));
}
template <bool REV>
__global__ void __launch_bounds__(256) combine_kernel(double *__restrict__ res, Cols cols, int ncol,
                                                      const double *__restrict__ x, int64_t nvec,
                                                      const double *__restrict__ coef) {
  __shared__ double sc[kMaxCols];
  if (threadIdx.x < ncol) sc[threadIdx.x] = coef[threadIdx.x];
  __syncthreads();
  const int64_t nb = (nvec + 255) / 256;
  for (int64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) {
    const int64_t i = (REV ? nb - 1 - blk : blk) * 256 + threadIdx.x;
    if (i >= nvec) continue;
    f64x2 q = reinterpret_cast<const f64x2 *>(x)[i];
    int c0 = 0;
    for (; c0 + 8 <= ncol; c0 += 8) {
      f64x2 cv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) cv[t] = reinterpret_cast<const f64x2 *>(cols.p[c0 + t])[i];
#pragma unroll
      for (int t = 0; t < 8; ++t) { const double cc = sc[c0 + t]; q[0] = q[0] + cc * cv[t][0]; q[1] = q[1] + cc * cv[t][1]; }
    }
    for (; c0 < ncol; c0 += 2) {
      f64x2 cv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) cv[t] = reinterpret_cast<const f64x2 *>(cols.p[c0 + t])[i];
#pragma unroll
      for (int t = 0; t < 2; ++t) { const double cc = sc[c0 + t]; q[0] = q[0] + cc * cv[t][0]; q[1] = q[1] + cc * cv[t][1]; }
    }
    reinterpret_cast<f64x2 *>(res)[i] = q;
  }
}

// ---- B: the persistent single-launch apply -------------------------------------------------------------------------
// Workgroup b owns chunks [b * cpw, (b + 1) * cpw) of CH = BLOCK * UB vectors. Dots: batches of NB columns, the
// workgroup's chunks inside (x re-read per batch: it sits in L2); partial sums published as self-validating 64-bit
// agent-scope stores (no fence), gathered by every workgroup (wave w sums columns w, w + nwaves, ...; lane l adds
// workgroups l, l + 64, ... in order, then one fixed DPP tree). Combine: the workgroup's chunks again (REV: last chunk
// first — what the dots phase touched last is the likeliest to sit in this XCD's L2), all columns per chunk.
// ILV (round 6): interleaved ownership — workgroup b takes chunks b, b + G, b + 2G, ...: at any moment the grid reads ONE contiguous
// region per column instead of G separate ones.
template <int BLOCK, int NB, int UB, bool REV, bool ILV = false>
__global__ void __launch_bounds__(BLOCK) persist_kernel(double *__restrict__ res, Cols cols, int ncol,
                                                        const double *__restrict__ x, int64_t nvec,
                                                        unsigned long long *__restrict__ slots, int parity, int cpw,
                                                        unsigned *__restrict__ fault) {
  constexpr int NW = BLOCK / 64;
  constexpr int64_t CH = (int64_t)BLOCK * UB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = (int)gridDim.x, b = (int)blockIdx.x;
  __shared__ double red[NW][kMaxCols];
  __shared__ double scoef[kMaxCols];
  unsigned long long *mine = slots + (int64_t)parity * kMaxCols * 1024, *other = slots + (int64_t)(1 - parity) * kMaxCols * 1024;
  for (int i = b * BLOCK + tid; i < kMaxCols * 1024; i += G * BLOCK)
    __hip_atomic_store(other + i, kEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t nchunks = (nvec + CH - 1) / CH;
  const int64_t ch0 = ILV ? 0 : (int64_t)b * cpw, ch1 = ILV ? (nchunks - b + G - 1) / G : (ch0 + cpw < nchunks ? ch0 + cpw : nchunks);
  auto chunk_of = [&](int64_t k) { return ILV ? k * G + b : k; };
  const f64x2 *xv = reinterpret_cast<const f64x2 *>(x);
  // ---- dots
  for (int c0 = 0; c0 < ncol; c0 += NB) {
    double acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = 0;
    const f64x2 *cp[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) cp[t] = reinterpret_cast<const f64x2 *>(cols.p[c0 + t < ncol ? c0 + t : ncol - 1]);
    for (int64_t kk = ch0; kk < ch1; ++kk) {
      const int64_t ch = chunk_of(kk);
      const int64_t base = ch * CH + tid;
      if (base + (int64_t)(UB - 1) * BLOCK < nvec) {
        f64x2 xe[UB], cv[NB][UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) xe[u] = xv[base + (int64_t)u * BLOCK];
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
          for (int u = 0; u < UB; ++u) cv[t][u] = cp[t][base + (int64_t)u * BLOCK];
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
          for (int u = 0; u < UB; ++u) acc[t] = fma(cv[t][u][1], xe[u][1], fma(cv[t][u][0], xe[u][0], acc[t]));
      } else {
        for (int u = 0; u < UB; ++u) {
          const int64_t i = base + (int64_t)u * BLOCK;
          if (i < nvec) {
            const f64x2 xe = xv[i];
#pragma unroll
            for (int t = 0; t < NB; ++t) { const f64x2 cv = cp[t][i]; acc[t] = fma(cv[1], xe[1], fma(cv[0], xe[0], acc[t])); }
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const double s = wave_allsum(acc[t]);
      if (lane == 0 && c0 + t < ncol) red[wave][c0 + t] = s;
    }
  }
  __syncthreads();
  if (tid < ncol) {
    double sv = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) sv += red[w][tid];
    unsigned long long bits = (unsigned long long)__double_as_longlong(sv);
    if (sv != sv) bits = 0x7FF8000000000000ull;
    __hip_atomic_store(mine + (int64_t)tid * 1024 + b, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- gather (bounded: ~2 s of polling ends with NaN + the fault word instead of a hang)
  for (int c = wave; c < ncol; c += NW) {
    double pv = 0;
    for (int w = lane; w < G; w += 64) {
      unsigned long long bits;
      unsigned it = 0;
      while ((bits = __hip_atomic_load(mine + (int64_t)c * 1024 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kEmpty) {
        __builtin_amdgcn_s_sleep(1);
        if (++it > (1u << 22)) { *fault = 1; bits = 0x7FF8000000000000ull; break; }
      }
      pv += __longlong_as_double((long long)bits);
    }
    const double v = wave_allsum(pv);
    if (lane == 0) scoef[c] = 1e-3 * v;
  }
  __syncthreads();
  // ---- combine
  for (int64_t k = 0; k < ch1 - ch0; ++k) {
    const int64_t ch = chunk_of(REV ? ch1 - 1 - k : ch0 + k);
    const int64_t base = ch * CH + tid;
    if (base + (int64_t)(UB - 1) * BLOCK < nvec) {
      f64x2 q[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) q[u] = xv[base + (int64_t)u * BLOCK];
      for (int c0 = 0; c0 < ncol; c0 += NB) {
        f64x2 cv[NB][UB];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const f64x2 *p = reinterpret_cast<const f64x2 *>(cols.p[c0 + t < ncol ? c0 + t : ncol - 1]);
#pragma unroll
          for (int u = 0; u < UB; ++u) cv[t][u] = p[base + (int64_t)u * BLOCK];
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const double cc = c0 + t < ncol ? scoef[c0 + t] : 0.0;
#pragma unroll
          for (int u = 0; u < UB; ++u) { q[u][0] = q[u][0] + cc * cv[t][u][0]; q[u][1] = q[u][1] + cc * cv[t][u][1]; }
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) reinterpret_cast<f64x2 *>(res)[base + (int64_t)u * BLOCK] = q[u];
    } else {
      for (int u = 0; u < UB; ++u) {
        const int64_t i = base + (int64_t)u * BLOCK;
        if (i < nvec) {
          f64x2 q = xv[i];
          for (int c = 0; c < ncol; ++c) {
            const f64x2 cv = reinterpret_cast<const f64x2 *>(cols.p[c])[i];
            const double cc = scoef[c];
            q[0] = q[0] + cc * cv[0]; q[1] = q[1] + cc * cv[1];
          }
          reinterpret_cast<f64x2 *>(res)[i] = q;
        }
      }
    }
  }
}

static hipEvent_t e0, e1;
template <typename F>
static double time_us(F &&f, int reps, int warm = 5) {
  for (int i = 0; i < warm; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / reps;
}

int main(int argc, char **argv) {
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# tools/tune_persist: %s, %d CUs\n", prop.name, cus);
  const int64_t nmax = 1ll << 22;
  const int maxcols = 40;
  double *panel, *x, *res, *partials, *coef, *sink;
  unsigned long long *slots;
  unsigned *fault;
  CK(hipMalloc(&panel, sizeof(double) * nmax * maxcols));
  CK(hipMalloc(&x, sizeof(double) * nmax));
  CK(hipMalloc(&res, sizeof(double) * nmax));
  CK(hipMalloc(&partials, sizeof(double) * maxcols * 4096));
  CK(hipMalloc(&coef, sizeof(double) * 64));
  CK(hipMalloc(&sink, sizeof(double) * 8192));
  CK(hipMalloc(&slots, sizeof(unsigned long long) * 2 * kMaxCols * 1024));
  CK(hipHostMalloc(&fault, sizeof(unsigned), hipHostMallocMapped));
  *fault = 0;
  {
    std::vector<double> h(nmax);
    for (int64_t i = 0; i < nmax; ++i) h[i] = ((i * 2654435761u) % 2001) * 1e-3 - 1.0;
    CK(hipMemcpy(x, h.data(), sizeof(double) * nmax, hipMemcpyHostToDevice));
    for (int c = 0; c < maxcols; ++c) {
      for (int64_t i = 0; i < nmax; i += 997) h[i] += 1e-3 * (c + 1);
      CK(hipMemcpy(panel + (int64_t)c * nmax, h.data(), sizeof(double) * nmax, hipMemcpyHostToDevice));
    }
    std::vector<unsigned long long> hs(2 * kMaxCols * 1024, kEmpty);
    CK(hipMemcpy(slots, hs.data(), sizeof(unsigned long long) * hs.size(), hipMemcpyHostToDevice));
  }
  const bool only_b = argc > 1 && argv[1][0] == 'B';

  if (!only_b) {
    printf("\n## A1. read-only sweep, steady state (buffer re-read back to back; 16-B loads, UNR per lane in flight)\n");
    for (double mb : {16.0, 28.0, 48.0, 84.0, 126.0, 168.0, 210.0, 250.0, 336.0, 672.0, 1342.0}) {
      const int64_t nvec = (int64_t)(mb * 1e6 / 16);
      double best = 1e30;
      int bg = 0, bu = 0;
      for (int per : {4, 8, 16}) {
        const double t4 = time_us([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(cus * per), dim3(256), 0, 0, (const f64x2 *)panel, nvec, sink); }, 30);
        const double t8 = time_us([&] { hipLaunchKernelGGL(read_kernel<8>, dim3(cus * per), dim3(256), 0, 0, (const f64x2 *)panel, nvec, sink); }, 30);
        if (t4 < best) { best = t4; bg = per; bu = 4; }
        if (t8 < best) { best = t8; bg = per; bu = 8; }
      }
      printf("footprint %7.1f MB: %8.1f us  %6.2f TB/s   (best: %d workgroups/CU, %d loads in flight)\n", mb, best, mb * 1e6 / best * 1e-6, bg, bu);
    }
  }

  struct Case { int e, ncol; };
  const Case cases[] = {{19, 10}, {19, 20}, {19, 40}, {20, 10}, {20, 20}, {20, 40}, {21, 10}, {21, 20}, {21, 40}, {22, 10}, {22, 20}, {22, 40}};
  for (const Case &cs : cases) {
    const int64_t n = 1ll << cs.e, nvec = n / 2;
    const int ncol = cs.ncol;
    Cols cols;
    for (int c = 0; c < kMaxCols; ++c) cols.p[c] = panel + (int64_t)(c < ncol ? c : 0) * nmax;
    const double bytes2 = (2.0 * ncol + 3) * 8 * n;   // two-pass bytes: (ncol + 1) read twice, res written
    printf("\n## n = 2^%d, %d columns: panel %.0f MB, two-pass bytes %.0f MB\n", cs.e, ncol, ncol * 8.0 * n / 1e6, bytes2 / 1e6);
    // reference: grid-wide launches
    const int pstride = 4096;
    auto dots = [&](int grid) {
      int c0 = 0;
      for (; c0 + 20 <= ncol; c0 += 20) hipLaunchKernelGGL(dots_kernel<20>, dim3(grid), dim3(256), 0, 0, cols, c0, x, nvec, partials, pstride);
      for (; c0 + 10 <= ncol; c0 += 10) hipLaunchKernelGGL(dots_kernel<10>, dim3(grid), dim3(256), 0, 0, cols, c0, x, nvec, partials, pstride);
    };
    const int gdots = (int)std::min<int64_t>((nvec + 255) / 256, cus * 8);
    const int gcomb = (int)((nvec + 255) / 256);
    auto fin = [&] { hipLaunchKernelGGL(finalize_kernel, dim3(ncol), dim3(256), 0, 0, partials, pstride, gdots, coef); };
    auto comb = [&](bool rev) {
      if (rev) hipLaunchKernelGGL(combine_kernel<true>, dim3(gcomb), dim3(256), 0, 0, res, cols, ncol, x, nvec, coef);
      else hipLaunchKernelGGL(combine_kernel<false>, dim3(gcomb), dim3(256), 0, 0, res, cols, ncol, x, nvec, coef);
    };
    const double t_d = time_us([&] { dots(gdots); }, 50);
    const double t3 = time_us([&] { dots(gdots); fin(); comb(false); }, 50);
    const double t3r = time_us([&] { dots(gdots); fin(); comb(true); }, 50);
    printf("launches  dots alone %7.1f us (%5.2f TB/s)   dots+finalize+combine %7.1f us (%5.2f TB/s of two-pass bytes, %.2f of 8)   reverse combine %7.1f us (%5.2f)\n",
           t_d, (ncol + 1) * 8.0 * n / t_d * 1e-6, t3, bytes2 / t3 * 1e-6, bytes2 / t3 * 1e-6 / 8, t3r, bytes2 / t3r * 1e-6);
    if (!only_b) {
      // A2: the combine pass alone, timed right after a dots pass (events around the combine only)
      for (int rev = 0; rev < 2; ++rev) {
        double tot = 0;
        const int reps = 30;
        for (int r = 0; r < reps + 3; ++r) {
          dots(gdots);
          fin();
          CK(hipEventRecord(e0));
          comb(rev);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (r >= 3) tot += ms * 1e3;
        }
        printf("ceiling   combine pass alone after a dots pass (%s): %7.1f us = %5.2f TB/s of its %.0f MB\n", rev ? "reverse" : "forward",
               tot / reps, (ncol + 2) * 8.0 * n / (tot / reps) * 1e-6, (ncol + 2) * 8.0 * n / 1e6);
      }
    }
    // persistent shapes
    int parity = 0;
    auto run = [&]<int BLOCK, int NB, int UB>(int per_cu) {
      const int G = cus * per_cu;
      if (G > 1024) return;
      constexpr int64_t CH = (int64_t)BLOCK * UB;
      const int64_t nchunks = (nvec + CH - 1) / CH;
      const int cpw = (int)((nchunks + G - 1) / G);
      int nb = 0;
      CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, persist_kernel<BLOCK, NB, UB, false>, BLOCK, 0));
      if (nb < per_cu) { printf("persist   block %4d x %d/CU NB=%2d UB=%d: not co-resident (occupancy %d)\n", BLOCK, per_cu, NB, UB, nb); return; }
      double t[2];
      for (int rev = 0; rev < 2; ++rev)
        t[rev] = time_us([&] {
          if (rev) hipLaunchKernelGGL((persist_kernel<BLOCK, NB, UB, true>), dim3(G), dim3(BLOCK), 0, 0, res, cols, ncol, x, nvec, slots, parity, cpw, fault);
          else hipLaunchKernelGGL((persist_kernel<BLOCK, NB, UB, false>), dim3(G), dim3(BLOCK), 0, 0, res, cols, ncol, x, nvec, slots, parity, cpw, fault);
          parity ^= 1;
        }, 50);
      if (*fault) { printf("FAULT: a gather timed out\n"); exit(2); }
      printf("persist   block %4d x %d/CU NB=%2d UB=%d (%3d chunks/wg): fwd %7.1f us (%5.2f TB/s, %.2f of 8)   rev %7.1f us (%5.2f TB/s, %.2f)\n", BLOCK,
             per_cu, NB, UB, cpw, t[0], bytes2 / t[0] * 1e-6, bytes2 / t[0] * 1e-6 / 8, t[1], bytes2 / t[1] * 1e-6, bytes2 / t[1] * 1e-6 / 8);
    };
    auto run_ilv = [&]<int BLOCK, int NB, int UB>(int per_cu) {
      const int G = cus * per_cu;
      double t[2];
      for (int rev = 0; rev < 2; ++rev)
        t[rev] = time_us([&] {
          if (rev) hipLaunchKernelGGL((persist_kernel<BLOCK, NB, UB, true, true>), dim3(G), dim3(BLOCK), 0, 0, res, cols, ncol, x, nvec, slots, parity, 0, fault);
          else hipLaunchKernelGGL((persist_kernel<BLOCK, NB, UB, false, true>), dim3(G), dim3(BLOCK), 0, 0, res, cols, ncol, x, nvec, slots, parity, 0, fault);
          parity ^= 1;
        }, 50);
      if (*fault) { printf("FAULT: a gather timed out\n"); exit(2); }
      printf("persist   block %4d x %d/CU NB=%2d UB=%d INTERLEAVED chunks  : fwd %7.1f us (%5.2f TB/s, %.2f of 8)   rev %7.1f us (%5.2f TB/s, %.2f)\n", BLOCK,
             per_cu, NB, UB, t[0], bytes2 / t[0] * 1e-6, bytes2 / t[0] * 1e-6 / 8, t[1], bytes2 / t[1] * 1e-6, bytes2 / t[1] * 1e-6 / 8);
    };
    if (getenv("TUNE_PERSIST_ILV")) {
      run.template operator()<512, 10, 1>(1);
      run_ilv.template operator()<512, 10, 1>(1);
      run_ilv.template operator()<512, 10, 1>(2);
      run_ilv.template operator()<256, 10, 1>(4);
      continue;
    }
    run.template operator()<256, 10, 1>(2);
    run.template operator()<256, 10, 1>(4);
    run.template operator()<256, 10, 2>(2);
    run.template operator()<256, 5, 4>(2);
    run.template operator()<256, 10, 2>(1);
    run.template operator()<256, 10, 4>(1);
    run.template operator()<512, 10, 1>(1);
    run.template operator()<512, 10, 1>(2);
    run.template operator()<512, 10, 2>(1);
    run.template operator()<512, 5, 2>(2);
    run.template operator()<1024, 10, 1>(1);
    run.template operator()<1024, 5, 2>(1);
    run.template operator()<1024, 5, 1>(1);
    // check the persistent result against the launches once
    {
      dots(gdots); fin(); comb(false);
      std::vector<double> a(n), bb(n);
      CK(hipMemcpy(a.data(), res, sizeof(double) * n, hipMemcpyDeviceToHost));
      const int G = cus;
      const int64_t nchunks = (nvec + 1023) / 1024;
      hipLaunchKernelGGL((persist_kernel<1024, 10, 1, true>), dim3(G), dim3(1024), 0, 0, res, cols, ncol, x, nvec, slots, parity, (int)((nchunks + G - 1) / G), fault);
      parity ^= 1;
      CK(hipMemcpy(bb.data(), res, sizeof(double) * n, hipMemcpyDeviceToHost));
      double md = 0, mx = 0;
      for (int64_t i = 0; i < n; ++i) { md = std::max(md, std::abs(a[i] - bb[i])); mx = std::max(mx, std::abs(a[i])); }
      printf("check     max |persist - launches| = %.3e (max |res| %.3e)\n", md, mx);
    }
  }
  return 0;
}
