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
// the stored entries are cut into CHUNKS of at most kSpChunk = 4096 entries, one workgroup each —
//   * SP_ROWS    consecutive whole rows, none longer than kSpLongRow entries, together <= 4096 entries;
//   * SP_LONG1   ONE row of kSpLongRow < nnz <= 4096 entries;
//   * SP_PIECE   a 4096-entry piece of a row longer than that; its sum goes to a carry slot and a small second
//                launch adds the pieces of each such row in order (only issued when such rows exist).
// A workgroup first streams its chunk: every lane loads 16 (value, index) pairs at stride 256 — whole cache lines per
// wave instruction, all 32 loads in flight — gathers the 16 x elements (a second round trip) and parks the f64 products
// in LDS (32 KiB: four to five workgroups per CU).
// (Measured and dropped, round 4: staging the chunk's x window in LDS so that the gathers become LDS reads — with the
// window in the same round trip as the values. On a random-banded pattern, 16 entries per row within +-2000 columns,
// 2048-entry chunks + a 33 KiB window ran at 410 us where this form takes 229 - 338 us: the window is as large as
// the chunk's own matrix data and the bigger LDS footprint leaves three workgroups per CU.)
// Then the rows of the chunk are summed out of LDS by lane groups whose width depends only on the number of rows in the
// chunk (1 .. 64 lanes per row, ascending entry order per lane, one fixed xor tree), long rows by the whole workgroup.
// So the HBM side is a coalesced stream independent of the row-length distribution (a few very long rows next to
// millions of short ones cost nothing extra), and the irregular part happens in LDS.
// Indices are 32-bit and 0-based inside the library whatever the caller stores: 12 B per entry (Float64).
#pragma once
#include "common.h"
#include "stream_kernels.h"

namespace mxlo {

constexpr int kSpChunk = 4096;     // entries per chunk = kBlock lanes x kSpPerLane
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

// LDS of a workgroup: kSpChunk products = 32 KiB exactly, so that FIVE workgroups share a CU's 160 KiB (the wave sums of
// the long-row path re-use the first slots of the product buffer once the products have been consumed)
constexpr size_t kSpLdsBytes = sizeof(double) * kSpChunk;

// One chunk: called by all kBlock threads of a workgroup. `prod` is kSpChunk doubles of LDS.
template <typename T, typename CA, typename CB, bool BETA0>
__device__ __forceinline__ void spmv_chunk(T *__restrict__ res, const T *__restrict__ x, const int64_t *__restrict__ ptr,
                                           const int32_t *__restrict__ idx, const T *__restrict__ val, const SpChunk c,
                                           double *__restrict__ carry, CA alpha, CB beta, double *prod) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- 1. stream the chunk: products into LDS
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
      T xx[kSpPerLane];
#pragma unroll
      for (int j = 0; j < kSpPerLane; ++j) xx[j] = x[ii[j]];
#pragma unroll
      for (int j = 0; j < kSpPerLane; ++j) prod[j * kBlock + tid] = (double)vv[j] * (double)xx[j];
    } else {
#pragma unroll
      for (int j = 0; j < kSpPerLane; ++j) {
        const int e = j * kBlock + tid;
        const bool in = e < c.nz;
        vv[j] = in ? vp[e] : T(0);
        ii[j] = in ? ip[e] : 0;
      }
      T xx[kSpPerLane];
#pragma unroll
      for (int j = 0; j < kSpPerLane; ++j) xx[j] = (j * kBlock + tid) < c.nz ? x[ii[j]] : T(0);
#pragma unroll
      for (int j = 0; j < kSpPerLane; ++j) {
        const int e = j * kBlock + tid;
        if (e < c.nz) prod[e] = (double)vv[j] * (double)xx[j];
      }
    }
  }
  __syncthreads();
  // ---- 2. sum the rows out of LDS
  if (c.kind == SP_ROWS) {
    int g = 1;                                            // lanes per row: from the row COUNT of the chunk only
    if (c.nr < kBlock) {
      int p2 = 1;
      while (p2 < c.nr) p2 <<= 1;
      g = kBlock / p2;
      if (g > 64) g = 64;
    }
    const int grp = tid / g, l = tid - grp * g, ngrp = kBlock / g;
    for (int r = grp; r < c.nr; r += ngrp) {
      const int64_t row = (int64_t)c.row0 + r;
      const int s0 = (int)(ptr[row] - c.k0), s1 = (int)(ptr[row + 1] - c.k0);
      double acc = 0.0;
      for (int e = s0 + l; e < s1; e += g) acc += prod[e];
      for (int off = g >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (l == 0) res[row] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)acc, beta, BETA0 ? T(0) : res[row]);
    }
  } else {                                                // one long row (or a piece of one): the whole workgroup
    double acc = 0.0;
    for (int e = tid; e < c.nz; e += kBlock) acc += prod[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __syncthreads();                                      // every product has been read: the buffer's head is free
    if (lane == 0) prod[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      const double tot = (prod[0] + prod[1]) + (prod[2] + prod[3]);
      if (c.kind == SP_LONG1) res[c.row0] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)tot, beta, BETA0 ? T(0) : res[c.row0]);
      else carry[c.carry] = tot;
    }
  }
}

// rows longer than a chunk: the pieces' sums added in order
template <typename T, typename CA, typename CB, bool BETA0>
__global__ void __launch_bounds__(kBlock)
spmv_fixup_kernel(T *__restrict__ res, const double *__restrict__ carry, const SpLongRow *__restrict__ rows, int nrows,
                  CA alpha, CB beta) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nrows) return;
  const SpLongRow lr = rows[i];
  double acc = 0.0;
  for (int p = 0; p < lr.npieces; ++p) acc += carry[lr.carry0 + p];
  res[lr.row] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)acc, beta, BETA0 ? T(0) : res[lr.row]);
}

}  // namespace mxlo
