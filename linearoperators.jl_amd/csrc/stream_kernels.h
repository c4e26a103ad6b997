// stream_kernels.h — the one elementwise streaming kernel every HBM-bound leaf uses.
//
// Design (MI355X): HBM-bound, zero reuse -> the only things that matter are
//  * 16-byte accesses per lane (global_load_dwordx4: 1 KiB per wave-instruction),
//  * many independent loads in flight per lane (UNROLL chunks x operands) to cover
//    the ~900-cycle HBM latency at 10 B/clk/CU,
//  * a persistent grid of num_cu*blocks_per_cu workgroups walking 256*UNROLL-vector
//    chunks (no reuse -> no XCD-aware remap needed; consecutive workgroups touching
//    consecutive chunks keeps every HBM channel busy).
// Arithmetic is a functor evaluated per element in the reference's association
// order; the TU is compiled with -ffp-contract=off so nothing is fused.
#pragma once
#include "common.h"

namespace mxlo {

// NT = nontemporal (streaming) access: `global_load/store ... nt`. Measured on MI355X at
// n = 1e8 fp64 (profiles/r01_tune_stream.txt): 2-read+1-write body 5.39 -> 6.34 TB/s, 2-read
// reduction 6.30 -> 7.03 TB/s. Only used when the operands exceed the caches anyway
// (Tune::nt_min_bytes); small vectors keep default loads so L2/MALL reuse between the leaves
// of a combinator survives.
template <bool NT, typename V>
__device__ __forceinline__ V ldg(const V *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename V>
__device__ __forceinline__ void stg(V *p, V v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// (Round 6, negative: write-through `sc1` stores for the one write stream of the many-column combine passes — +3 ... 6 % in a
//  bare harness, tools/tune_store.hip — change nothing in combine_kernel: profiles/r06_tune_store.txt.)
template <typename T, int VEC>
struct VecOf {
  using type = T;
};
template <>
struct VecOf<double, 2> {
  using type = f64x2;
};
template <>
struct VecOf<float, 4> {
  using type = f32x4;
};

// complex elements travel as plain float/double vectors (nontemporal builtins take scalar/vector types only)
template <>
struct VecOf<cx<double>, 1> {
  using type = f64x2;
};
template <>
struct VecOf<cx<float>, 1> {
  using type = f32x2;
};
template <>
struct VecOf<cx<float>, 2> {
  using type = f32x4;
};

template <typename T, int VEC>
__device__ __forceinline__ T vget(const typename VecOf<T, VEC>::type &v, int i) {
  if constexpr (is_cx<T>::value) return T(v[2 * i], v[2 * i + 1]);
  else if constexpr (VEC == 1) return v;
  else return v[i];
}
template <typename T, int VEC>
__device__ __forceinline__ void vset(typename VecOf<T, VEC>::type &v, int i, T x) {
  if constexpr (is_cx<T>::value) {
    v[2 * i] = x.re;
    v[2 * i + 1] = x.im;
  } else if constexpr (VEC == 1) v = x;
  else v[i] = x;
}

// Op concept:
//   __device__ void init();                       // once per thread (load device scalars)
//   __device__ T operator()(T in0, T in1, T res_old) const;
// NIN = number of input streams (0,1,2); READ_RES = res is also an input (beta != 0).
template <typename T, int VEC, int UNROLL, int NIN, bool READ_RES, bool REVERSE, bool NT, typename Op>
__global__ void __launch_bounds__(kBlock)
map_kernel(T *__restrict__ res, const T *__restrict__ in0, const T *__restrict__ in1, int64_t head,
           int64_t nvec, int64_t n, Op op) {
  using V = typename VecOf<T, VEC>::type;
  op.init();
  const int tid = threadIdx.x;
  V *rv = reinterpret_cast<V *>(res + head);
  const V *av = reinterpret_cast<const V *>(in0 + head);
  const V *bv = reinterpret_cast<const V *>(in1 + head);
  constexpr int64_t CHUNK = (int64_t)kBlock * UNROLL;
  const int64_t nchunks = (nvec + CHUNK - 1) / CHUNK;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t c = REVERSE ? (nchunks - 1 - ch) : ch;
    const int64_t base = c * CHUNK + tid;
    V a[UNROLL], b[UNROLL], r[UNROLL];
    if (base + (int64_t)(UNROLL - 1) * kBlock < nvec) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * kBlock;
        if constexpr (NIN >= 1) a[u] = ldg<NT>(av + i);
        if constexpr (NIN >= 2) b[u] = ldg<NT>(bv + i);
        if constexpr (READ_RES) r[u] = ldg<NT>(rv + i);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        V o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const T x0 = NIN >= 1 ? vget<T, VEC>(a[u], e) : T(0);
          const T x1 = NIN >= 2 ? vget<T, VEC>(b[u], e) : T(0);
          const T r0 = READ_RES ? vget<T, VEC>(r[u], e) : T(0);
          vset<T, VEC>(o, e, op(x0, x1, r0));
        }
        stg<NT>(rv + base + (int64_t)u * kBlock, o);
      }
    } else {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + (int64_t)u * kBlock;
        if (i < nvec) {
          V o, aa, bb, rr;
          if constexpr (NIN >= 1) aa = ldg<NT>(av + i);
          if constexpr (NIN >= 2) bb = ldg<NT>(bv + i);
          if constexpr (READ_RES) rr = ldg<NT>(rv + i);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const T x0 = NIN >= 1 ? vget<T, VEC>(aa, e) : T(0);
            const T x1 = NIN >= 2 ? vget<T, VEC>(bb, e) : T(0);
            const T r0 = READ_RES ? vget<T, VEC>(rr, e) : T(0);
            vset<T, VEC>(o, e, op(x0, x1, r0));
          }
          stg<NT>(rv + i, o);
        }
      }
    }
  }
  if constexpr (VEC > 1) {
    // scalar head [0,head) and tail [head + nvec*VEC, n): at most 2*(VEC-1) elements
    if (blockIdx.x == gridDim.x - 1) {
      const int64_t tail0 = head + nvec * VEC;
      const int64_t cnt = head + (n - tail0);
      if (tid < cnt) {
        const int64_t i = tid < head ? tid : tail0 + (tid - head);
        const T x0 = NIN >= 1 ? in0[i] : T(0);
        const T x1 = NIN >= 2 ? in1[i] : T(0);
        const T r0 = READ_RES ? res[i] : T(0);
        res[i] = op(x0, x1, r0);
      }
    }
  }
}

constexpr int kStreamUnroll = 4;

// Host launcher: picks the 16-byte path when all operands share their alignment, nontemporal
// accesses when the streamed footprint exceeds Tune::nt_min_bytes, and the grid: one
// 256*UNROLL-vector chunk per workgroup (blocks_per_cu == 0, best measured) or a persistent
// grid of num_cu*blocks_per_cu workgroups.
template <typename T, int NIN, bool READ_RES, bool REVERSE, typename Op>
int32_t launch_map(mxlo_ctx *ctx, T *res, const T *in0, const T *in1, int64_t n, Op op) {
  if (n <= 0) return MXLO_OK;
  constexpr int VEC = Vec16<T>::N;
  const int64_t head0 = common_head<T>({res, NIN >= 1 ? in0 : nullptr, NIN >= 2 ? in1 : nullptr});
  const int64_t streamed = (int64_t)sizeof(T) * n * (1 + NIN + (READ_RES ? 1 : 0));
  const bool nt = streamed >= ctx->tune.nt_min_bytes;
  auto grid_of = [&](int64_t items) {
    const int64_t need = (items + (int64_t)kBlock * kStreamUnroll - 1) / ((int64_t)kBlock * kStreamUnroll);
    int64_t g = need;
    if (ctx->tune.blocks_per_cu > 0) {
      const int64_t cap = (int64_t)ctx->num_cu * ctx->tune.blocks_per_cu;
      if (g > cap) g = cap;
    }
    if (g > 0x7fffffffLL) g = 0x7fffffffLL;
    return (int)(g < 1 ? 1 : g);
  };
  if (head0 >= 0 && n >= 4 * VEC) {
    const int64_t head = head0 < n ? head0 : n;
    const int64_t nvec = (n - head) / VEC;
    const int grid = grid_of(nvec);
    if (nt)
      hipLaunchKernelGGL((map_kernel<T, VEC, kStreamUnroll, NIN, READ_RES, REVERSE, true, Op>), dim3(grid),
                         dim3(kBlock), 0, ctx->stream, res, in0, in1, head, nvec, n, op);
    else
      hipLaunchKernelGGL((map_kernel<T, VEC, kStreamUnroll, NIN, READ_RES, REVERSE, false, Op>), dim3(grid),
                         dim3(kBlock), 0, ctx->stream, res, in0, in1, head, nvec, n, op);
  } else {
    const int grid = grid_of(n);
    hipLaunchKernelGGL((map_kernel<T, 1, kStreamUnroll, NIN, READ_RES, REVERSE, false, Op>), dim3(grid),
                       dim3(kBlock), 0, ctx->stream, res, in0, in1, (int64_t)0, n, n, op);
  }
  MXLO_LAUNCH_CHECK();
  return MXLO_OK;
}

// ---- caller-scalar types ---------------------------------------------------------------------
// Julia does not convert the caller's alpha / beta to the element type T: in
//   res .= α .* d .* v .+ β .* res            (src/special-operators.jl:126-129 and every other leaf)
// the α-term is evaluated in CA = promote_type(typeof(α), T), the β-term in CB = promote_type(typeof(β), T),
// their sum in promote_type(CA, CB), and the result is rounded once on store. For Float64 data CA = CB = double;
// for Float32 data each scalar is float or double on its own (MXLO_ALPHA_F64 / MXLO_BETA_F64), so
// mul!(res32, op32, v32, α::Float32, β::Float64) rounds (α*d)*v in Float32 first, exactly like the reference.
template <typename CA, typename CB>
using Wider = std::conditional_t<(sizeof(CA) >= sizeof(CB)), CA, CB>;

// res = T(t (+ b*r)) with t already evaluated in CA
template <typename T, typename CA, typename CB, bool BETA0>
__device__ __forceinline__ T fin_ab(CA t, CB b, T r) {
  if constexpr (BETA0) return (T)t;
  else {
    using P = Wider<CA, CB>;
    return (T)((P)t + (P)(b * (CB)r));
  }
}

// ---- functors: one per reference statement ------------------------------------
// res = (a*d)*v (+ b*res)      src/special-operators.jl:126-129,146-148
template <typename T, typename CA, typename CB, bool BETA0>
struct DiagOp {
  CA a;
  CB b;
  __device__ void init() {}
  __device__ T operator()(T d, T v, T r) const { return fin_ab<T, CA, CB, BETA0>((a * (CA)d) * (CA)v, b, r); }
};
// 1-element d (SpectralGradient): d broadcast from device memory
template <typename T, typename CA, typename CB, bool BETA0>
struct DiagScalarOp {
  CA a;
  CB b;
  const T *dptr;
  CA ad;
  __device__ void init() { ad = a * (CA)(*dptr); }
  __device__ T operator()(T v, T, T r) const { return fin_ab<T, CA, CB, BETA0>(ad * (CA)v, b, r); }
};
// res = a*v (+ b*res)          src/special-operators.jl:38-41, src/operations.jl:18
template <typename T, typename CA, typename CB, bool BETA0>
struct AxpbyOp {
  CA a;
  CB b;
  __device__ void init() {}
  __device__ T operator()(T v, T, T r) const { return fin_ab<T, CA, CB, BETA0>(a * (CA)v, b, r); }
};
// res = res*b                  src/special-operators.jl:106, src/operations.jl:14
template <typename T, typename CT>
struct ScaleOp {
  CT b;
  __device__ void init() {}
  __device__ T operator()(T, T, T r) const { return (T)((CT)r * b); }
};
// res = const
template <typename T>
struct FillOp {
  T c;
  __device__ void init() {}
  __device__ T operator()(T, T, T) const { return c; }
};
// res = c0 (+ b*res) with c0 = a*sum(v) read from device memory   src/special-operators.jl:81-83
template <typename T, typename CA, typename CB, bool BETA0>
struct OnesOp {
  CA a;
  CB b;
  const double *sum;
  CA as;
  __device__ void init() { as = a * (CA)(T)(*sum); }
  __device__ T operator()(T, T, T r) const { return fin_ab<T, CA, CB, BETA0>(as, b, r); }
};
// res = a*(v - c*h) (+ b*res), c = 2*dot(h,v) from device memory   src/linalg.jl:79-81
template <typename T, typename CA, typename CB, bool BETA0>
struct HouseholderOp {
  CA a;
  CB b;
  const double *dot;
  T c;
  __device__ void init() { c = (T)2 * (T)(*dot); }
  __device__ T operator()(T h, T v, T r) const {
    const T inner = v - (c * h);
    return fin_ab<T, CA, CB, BETA0>(a * (CA)inner, b, r);
  }
};
// the same with the dot taken from the dots pass's per-workgroup partial sums: every workgroup adds them up itself, in
// finalize_kernel's order (lane t: t, t+256, ...; shuffle tree; the four waves pairwise) — bit-identical to the
// finalized value in every workgroup, and one dependent launch fewer (mid sizes, no all-reduce hook)
template <typename T, typename CA, typename CB, bool BETA0>
struct HouseholderPartialsOp {
  CA a;
  CB b;
  const double *partials;
  int nblocks;
  T c;
  __device__ void init() {
    __shared__ double hp_lds[kBlock / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) s += partials[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) hp_lds[threadIdx.x >> 6] = s;
    __syncthreads();
    c = (T)2 * (T)((hp_lds[0] + hp_lds[1]) + (hp_lds[2] + hp_lds[3]));
  }
  __device__ T operator()(T h, T v, T r) const {
    const T inner = v - (c * h);
    return fin_ab<T, CA, CB, BETA0>(a * (CA)inner, b, r);
  }
};

inline bool alpha_is_f64(size_t elt, int32_t flags) { return elt == 8 || (flags & MXLO_ALPHA_F64); }
inline bool beta_is_f64(size_t elt, int32_t flags) { return elt == 8 || (flags & MXLO_BETA_F64); }

// Dispatch over (type of alpha, type of beta, beta == 0): calls f.template operator()<CA, CB, BETA0>().
// beta == 0 never reads res, so CB is irrelevant there (instantiated as CA).
template <typename T, typename F>
int32_t dispatch_ab(double beta, int32_t flags, F &&f) {
  if constexpr (sizeof(T) == 8) {
    return beta == 0 ? f.template operator()<double, double, true>() : f.template operator()<double, double, false>();
  } else {
    const bool ad = alpha_is_f64(sizeof(T), flags), bd = beta_is_f64(sizeof(T), flags);
    if (beta == 0) return ad ? f.template operator()<double, double, true>() : f.template operator()<float, float, true>();
    if (ad) return bd ? f.template operator()<double, double, false>() : f.template operator()<double, float, false>();
    return bd ? f.template operator()<float, double, false>() : f.template operator()<float, float, false>();
  }
}

// Round caller scalars to the types the flags say they have (a Float32 scalar arrives widened to double).
inline void eff_scalars(size_t elt, int32_t flags, double &alpha, double &beta) {
  if (!alpha_is_f64(elt, flags)) alpha = (double)(float)alpha;
  if (!beta_is_f64(elt, flags)) beta = (double)(float)beta;
}

}  // namespace mxlo
