// sparse_kernels.h — device side of the sparse (SparseMatrixCSC) leaf: shared by sparse.hip (mxlo_csc_mul) and
// blockdiag.hip (MXLO_BLK_CSC blocks of a fused BlockDiagonalOperator).
//
// The reference hands a sparse M to LinearAlgebra.mul!(res, M, v, α, β) (src/constructors.jl:19-29), i.e. to the
// SparseArrays stdlib: a column sweep with scattered `+=` for A*x and a per-column gather-reduce for Aᵀ*x. On the
// device BOTH modes are a row gather-reduce over a compressed-row view of the operand that is applied:
//   N mode: rows of A   — the CSR view built once at construction (rowptr, colidx, values permuted into row order),
//   T / C mode: rows of Aᵀ = columns of A — the CSC arrays themselves (values read in place).
//
// Work decomposition (fixed at construction, so every apply sums in the same order: run-to-run bit-identical):
// the stored entries are cut into CHUNKS of at most kSpChunk = 2048 entries, one workgroup each —
//   * SP_ROWS    consecutive whole rows, none longer than kSpLongRow entries, together <= 2048 entries;
//   * SP_LONG1   ONE row of kSpLongRow < nnz <= 2048 entries;
//   * SP_PIECE   a 2048-entry piece of a row longer than that; its sum goes to a carry slot and a small second
//                launch adds the pieces of each such row in order (only issued when such rows exist).
// A workgroup first STREAMS its chunk into LDS: every lane loads 8 (value, index) pairs at stride 256 — whole cache
// lines per wave instruction, all 16 loads in flight, together with the row pointers its lane group will need — and
// parks them (24 KiB for Float64: six workgroups per CU; chunks of 4096 entries — three per CU — ran the stencils at
// 0.52 - 0.55 of HBM peak where 2048 gives 0.60 - 0.63, chunks of 1024 are slower again). So the HBM side is a coalesced stream independent of the
// row-length distribution (a few very long rows next to millions of short ones cost nothing extra).
// Then the ROWS are walked out of LDS by lane groups whose width depends only on the number of rows in the chunk
// (1 .. 64 lanes per row; long rows: the whole workgroup): lane l of a group takes its row's entries l, l + g, ... in
// ascending order, 8 at a time (8 x gathers in flight), adds the products in that order, and the group's partials are
// combined by one fixed xor tree. With one lane per row — the usual case: >= 256 rows in a chunk, or any chunk of rows
// of <= 8 entries — ADJACENT LANES GATHER x FOR ADJACENT ROWS: on the patterns solvers apply (stencils, bands, FEM
// meshes) those are adjacent addresses, so a wave's gather touches a handful of cache lines instead of one per lane.
// (Round-4 measurements, 7-point Laplacian 160^3, profiles/r04_pmc_sparse.txt: gathering in ENTRY order — adjacent
// lanes = adjacent entries of one row = far-apart columns — cost ~60 L1 accesses per gather instruction, the vector
// L1s were the limiter (0.57 accesses per cycle and CU, HBM traffic 1.03 x algorithmic at only 4.4 TB/s). Also
// measured and dropped: an LDS copy of the chunk's x window (2048-entry chunks + 33 KiB window: 410 us where entry-order
// gathers took 229 - 338 us on a random band), persistent workgroups with the next chunk's loads in flight (216 VGPRs,
// two workgroups per CU: 146 us vs 101 us), an XCD-banded chunk order (no change: x sits in the Infinity Cache).)
// Indices are 32-bit and 0-based inside the library whatever the caller stores: 12 B per entry (Float64).
#pragma once
#include "common.h"
#include "stream_kernels.h"
#include "complex_scalars.h"

namespace mxlo {

constexpr int kSpChunk = 2048;     // entries per chunk = kBlock lanes x kSpPerLane
constexpr int kSpPerLane = kSpChunk / kBlock;
constexpr int kSpLongRow = 512;    // a row with more entries gets chunks of its own
enum { SP_ROWS = 0, SP_LONG1 = 1, SP_PIECE = 2 };

struct SpChunk {          // 32 bytes, one per workgroup
  int64_t k0;             // first entry
  int32_t nz;             // entries (<= kSpChunk)
  int32_t row0;           // first row
  int32_t nr;             // rows (SP_ROWS), 1 otherwise
  int32_t kind;           // SP_*
  int32_t carry;          // SP_PIECE: slot of this piece's sum
  int32_t pad;
};
struct SpLongRow {        // a row cut into pieces: summed by the fix-up launch
  int32_t row, carry0, npieces, pad;
};

// Device-resident description of one sparse operand (owned by its mxlo_csc handle).
struct CscDev {
  const int64_t *rowptr;   // [m + 1] CSR view (N mode)
  const int32_t *colidx;   // [nnz]
  const void *csr_val;     // [nnz] values in row order (snapshot: mxlo_csc_refresh re-gathers them)
  const int64_t *colptr;   // [n + 1] CSC arrays, 0-based copies (T / C mode)
  const int32_t *rowidx;   // [nnz]
  const void *nzval;       // [nnz] the caller's values, read in place
  const SpChunk *chunks_n, *chunks_t;
  double *carry;           // piece sums of rows longer than a chunk (shared by both modes: applies are stream-ordered)
  int64_t m, n, nnz;
  int32_t nchunks_n, nchunks_t, nlong_n, nlong_t;
};

// LDS of a workgroup: kSpChunk values (element type) + kSpChunk 32-bit indices
template <typename T>
constexpr size_t sp_lds_bytes() { return (sizeof(T) + sizeof(int32_t)) * (size_t)kSpChunk; }
constexpr size_t kSpLdsBytesMax = sp_lds_bytes<double>();
constexpr int kSpBatch = 8;        // gathers in flight per lane while a row is walked (4: -5 ... -10 %, 16: -30 % on the stencils)

// lanes per row of a SP_ROWS chunk: from the row COUNT of the chunk only
__device__ __forceinline__ int sp_lanes_per_row(int nr) {
  if (nr >= kBlock) return 1;
  int p2 = 1;
  while (p2 < nr) p2 <<= 1;
  const int g = kBlock / p2;
  return g > 64 ? 64 : g;
}

// ---- element arithmetic: real (accumulator = one double) and complex (two); CONJ conjugates the stored value (A' * x)
template <typename T>
struct SpNum {                     // Float64 / Float32
  static constexpr int NACC = 1;
  struct Acc {
    double v;
  };
  __device__ static __forceinline__ Acc zero() { return Acc{0.0}; }
  template <bool CONJ>
  __device__ static __forceinline__ void madd(Acc &a, T val, T x) { a.v += (double)val * (double)x; }
  __device__ static __forceinline__ void add(Acc &a, const Acc &b) { a.v += b.v; }
  __device__ static __forceinline__ Acc shfl_xor(const Acc &a, int off) { return Acc{__shfl_xor(a.v, off, 64)}; }
  __device__ static __forceinline__ void store(double *p, const Acc &a) { p[0] = a.v; }
  __device__ static __forceinline__ Acc load(const double *p) { return Acc{p[0]}; }
};
template <typename R>
struct SpNum<cx<R>> {              // ComplexF64 / ComplexF32: Julia's component-wise product, nothing fused
  static constexpr int NACC = 2;
  struct Acc {
    double re, im;
  };
  __device__ static __forceinline__ Acc zero() { return Acc{0.0, 0.0}; }
  template <bool CONJ>
  __device__ static __forceinline__ void madd(Acc &a, cx<R> val, cx<R> x) {
    const double vr = (double)val.re, vi = CONJ ? -(double)val.im : (double)val.im, xr = (double)x.re, xi = (double)x.im;
    a.re += (vr * xr) - (vi * xi);
    a.im += (vr * xi) + (vi * xr);
  }
  __device__ static __forceinline__ void add(Acc &a, const Acc &b) { a.re += b.re; a.im += b.im; }
  __device__ static __forceinline__ Acc shfl_xor(const Acc &a, int off) {
    return Acc{__shfl_xor(a.re, off, 64), __shfl_xor(a.im, off, 64)};
  }
  __device__ static __forceinline__ void store(double *p, const Acc &a) { p[0] = a.re; p[1] = a.im; }
  __device__ static __forceinline__ Acc load(const double *p) { return Acc{p[0], p[1]}; }
};
// ---- the closing step res = alpha * sum (+ beta * res): the scalars in the types the caller passed them in
template <typename T, typename CA, typename CB, bool BETA0>
struct SpFinReal {
  CA alpha;
  CB beta;
  __device__ __forceinline__ T operator()(const typename SpNum<T>::Acc &a, T old) const {
    return fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)a.v, beta, old);
  }
};
template <typename R, typename RA, typename RB, bool BETA0>
struct SpFinCplx {
  Sc<RA> alpha;
  RB bre, bim;
  bool b_real;
  __device__ __forceinline__ cx<R> operator()(const typename SpNum<cx<R>>::Acc &a, cx<R> old) const {
    RA tr, ti;
    alpha.template mul<R>(cx<R>((R)a.re, (R)a.im), tr, ti);
    return cfin<R, RA, RB, BETA0>(tr, ti, bre, bim, b_real, old);
  }
};

// sum of (value * x[index]) over the LDS entries e0, e0 + step, ... < e1, in that order, kSpBatch gathers at a time
template <typename T, bool CONJ>
__device__ __forceinline__ typename SpNum<T>::Acc sp_walk(const T *__restrict__ x, const T *sval, const int32_t *sidx, int e0,
                                                           int e1, int step) {
  typename SpNum<T>::Acc acc = SpNum<T>::zero();
  for (int e = e0; e < e1; e += step * kSpBatch) {
    T v[kSpBatch], xv[kSpBatch];
#pragma unroll
    for (int u = 0; u < kSpBatch; ++u) {
      const int q = e + u * step;
      const bool in = q < e1;
      v[u] = in ? sval[q] : T(0);
      xv[u] = in ? x[sidx[q]] : T(0);
    }
#pragma unroll
    for (int u = 0; u < kSpBatch; ++u)
      if (e + u * step < e1) SpNum<T>::template madd<CONJ>(acc, v[u], xv[u]);
  }
  return acc;
}

// One chunk, start to end: called by all kBlock threads of a workgroup; `lds` is sp_lds_bytes<T>() of LDS.
constexpr int kSpPre = 3;          // rows per lane group whose pointers (and old res) are requested with the chunk's own loads
                                   // (5: no better on the 7-point pattern, -10 % on the 27-point one)
// ncols / ldx / ldr: the apply on an n x ncols block (`mul!` on matrices, src/operations.jl:34-36): the chunk is streamed
// into LDS ONCE and its rows are walked once per column of x (column j of x at x + j*ldx, of res at res + j*ldr) — A is
// read once for the whole block. Piece sums of column j go to carry[(piece * ncols + j) * NACC ...].
// BLOCK = false fixes ncols = 1 at compile time (the vector apply keeps its registers: 77-90 instead of 94-106).
// FIN: SpFinReal / SpFinCplx (the closing step); CONJ: the stored values are conjugated (complex A' * x).
template <typename T, typename FIN, bool BETA0, bool BLOCK = false, bool CONJ = false>
__device__ __forceinline__ void spmv_chunk(T *__restrict__ res, const T *__restrict__ x, const int64_t *__restrict__ ptr,
                                           const int32_t *__restrict__ idx, const T *__restrict__ val, const SpChunk c,
                                           double *__restrict__ carry, const FIN fin, void *lds, int ncols_ = 1,
                                           int64_t ldx = 0, int64_t ldr = 0) {
  using N = SpNum<T>;
  const int ncols = BLOCK ? ncols_ : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T *sval = (T *)lds;
  int32_t *sidx = (int32_t *)((char *)lds + sizeof(T) * kSpChunk);
  const int g = c.kind == SP_ROWS ? sp_lanes_per_row(c.nr) : 1;
  const int grp = tid / g, l = tid - grp * g, ngrp = kBlock / g;
  // ---- 1. stream the chunk into LDS; the first rows' pointers ride in the same round trip
  int64_t p0[kSpPre], p1[kSpPre];
  T old[kSpPre];
  {
    const T *vp = val + c.k0;
    const int32_t *ip = idx + c.k0;
    T vv[kSpPerLane];
    int32_t ii[kSpPerLane];
    if (c.nz == kSpChunk) {
#pragma unroll
      for (int j = 0; j < kSpPerLane; ++j) {
        vv[j] = vp[j * kBlock + tid];
        ii[j] = ip[j * kBlock + tid];
      }
    } else {
#pragma unroll
      for (int j = 0; j < kSpPerLane; ++j) {
        const int e = j * kBlock + tid;
        const bool in = e < c.nz;
        vv[j] = in ? vp[e] : T(0);
        ii[j] = in ? ip[e] : 0;
      }
    }
    if (c.kind == SP_ROWS) {
#pragma unroll
      for (int k = 0; k < kSpPre; ++k) {
        const int r = grp + k * ngrp;
        if (r < c.nr) {
          const int64_t row = (int64_t)c.row0 + r;
          p0[k] = ptr[row];
          p1[k] = ptr[row + 1];
          if constexpr (!BETA0) old[k] = res[row];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kSpPerLane; ++j) {
      const int e = j * kBlock + tid;
      if (c.nz == kSpChunk || e < c.nz) {
        sval[e] = vv[j];
        sidx[e] = ii[j];
      }
    }
  }
  __syncthreads();
  // ---- 2. walk the rows
  if (c.kind == SP_ROWS) {
    int k = 0;
    for (int r = grp; r < c.nr; r += ngrp, ++k) {
      const int64_t row = (int64_t)c.row0 + r;
      int64_t q0 = 0, q1 = 0;
      T ro = T(0);
      bool have = false;
#pragma unroll
      for (int t = 0; t < kSpPre; ++t)
        if (k == t) {
          q0 = p0[t];
          q1 = p1[t];
          if constexpr (!BETA0) ro = old[t];
          have = true;
        }
      if (!have) {
        q0 = ptr[row];
        q1 = ptr[row + 1];
        if constexpr (!BETA0) ro = res[row];
      }
      for (int j = 0; j < ncols; ++j) {
        typename N::Acc acc = sp_walk<T, CONJ>(x + j * ldx, sval, sidx, (int)(q0 - c.k0) + l, (int)(q1 - c.k0), g);
        for (int off = g >> 1; off > 0; off >>= 1) N::add(acc, N::shfl_xor(acc, off));
        if (l == 0) {
          T *rp = res + j * ldr + row;
          if constexpr (!BETA0) if (j > 0) ro = *rp;
          *rp = fin(acc, ro);
        }
      }
    }
  } else {                                                // one long row (or a piece of one): the whole workgroup
    __shared__ double red[(kBlock / 64) * 2];
    for (int j = 0; j < ncols; ++j) {
      typename N::Acc acc = sp_walk<T, CONJ>(x + j * ldx, sval, sidx, tid, c.nz, kBlock);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) N::add(acc, N::shfl_xor(acc, off));
      if (lane == 0) N::store(red + wave * N::NACC, acc);
      __syncthreads();
      if (tid == 0) {
        typename N::Acc a01 = N::load(red), a1 = N::load(red + N::NACC), a23 = N::load(red + 2 * N::NACC),
                        a3 = N::load(red + 3 * N::NACC);
        N::add(a01, a1);
        N::add(a23, a3);
        N::add(a01, a23);                                  // (w0 + w1) + (w2 + w3)
        T *rp = res + j * ldr + c.row0;
        if (c.kind == SP_LONG1) *rp = fin(a01, BETA0 ? T(0) : *rp);
        else N::store(carry + ((int64_t)c.carry * ncols + j) * N::NACC, a01);
      }
      __syncthreads();                                    // red is free for the next column
    }
  }
}

// rows longer than a chunk: the pieces' sums added in order
template <typename T, typename FIN, bool BETA0>
__global__ void __launch_bounds__(kBlock)
spmv_fixup_kernel(T *__restrict__ res, const double *__restrict__ carry, const SpLongRow *__restrict__ rows, int nrows,
                  const FIN fin, int ncols, int64_t ldr) {
  using N = SpNum<T>;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nrows * ncols) return;
  const SpLongRow lr = rows[i / ncols];
  const int j = i % ncols;
  typename N::Acc acc = N::zero();
  for (int p = 0; p < lr.npieces; ++p) N::add(acc, N::load(carry + ((int64_t)(lr.carry0 + p) * ncols + j) * N::NACC));
  T *rp = res + j * ldr + lr.row;
  *rp = fin(acc, BETA0 ? T(0) : *rp);
}

}  // namespace mxlo
