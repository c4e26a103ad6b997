// sparse_kernels.h — device side of the sparse (SparseMatrixCSC) leaf: shared by sparse.hip (mxlo_csc_mul) and
// blockdiag.hip (MXLO_BLK_CSC blocks of a fused BlockDiagonalOperator).
//
// The reference hands a sparse M to LinearAlgebra.mul!(res, M, v, α, β) (src/constructors.jl:19-29), i.e. to the
// SparseArrays stdlib: a column sweep with scattered `+=` for A*x and a per-column gather-reduce for Aᵀ*x. On the
// device BOTH modes are a row gather-reduce over a compressed-row view of the operand that is applied:
//   N mode: rows of A   — the CSR view built once at construction (rowptr, colidx, values permuted into row order),
//   T / C mode: rows of Aᵀ = columns of A — the CSC arrays themselves (values read in place).
// A group of LPR lanes (a power of two <= 64, chosen from the mean row length) owns one output row: lane l takes the
// row's entries l, l + LPR, ... in order (consecutive lanes read consecutive entries: coalesced index and value
// streams), accumulates in f64 with fma, and the group's partials are combined by one fixed xor tree — the result of a
// row depends only on (row, LPR), so applies are run-to-run bit-identical. Indices are 32-bit and 0-based inside the
// library whatever the caller stores (12 B per entry instead of 16).
#pragma once
#include "common.h"
#include "stream_kernels.h"

namespace mxlo {

// Device-resident description of one sparse operand (owned by its mxlo_csc handle).
struct CscDev {
  const int64_t *rowptr;   // [m + 1] CSR view (N mode)
  const int32_t *colidx;   // [nnz]
  const void *csr_val;     // [nnz] values in row order (snapshot: mxlo_csc_refresh re-gathers them)
  const int64_t *colptr;   // [n + 1] CSC arrays, 0-based copies (T / C mode)
  const int32_t *rowidx;   // [nnz]
  const void *nzval;       // [nnz] the caller's values, read in place
  int64_t m, n, nnz;
  int32_t lpr_n, lpr_t;    // lanes per row of the N / T sweeps
};

// Output rows [row0, row0 + nrows) of   res = α * (R x) + β * res   for the compressed-row operand (ptr, idx, val).
// Called by all kBlock threads of a workgroup; rows are dealt to lane groups of LPR consecutive lanes.
template <typename T, typename CA, typename CB, bool BETA0>
__device__ __forceinline__ void spmv_rows(T *__restrict__ res, const T *__restrict__ x, const int64_t *__restrict__ ptr,
                                          const int32_t *__restrict__ idx, const T *__restrict__ val, int64_t row0,
                                          int64_t nrows, int lpr, CA alpha, CB beta) {
  const int tid = threadIdx.x;
  const int rows_per_pass = kBlock / lpr;
  const int g = tid / lpr, l = tid - g * lpr;
  for (int64_t r = g; r < nrows; r += rows_per_pass) {
    const int64_t row = row0 + r;
    const int64_t k0 = ptr[row], k1 = ptr[row + 1];
    double acc = 0.0;
    for (int64_t k = k0 + l; k < k1; k += lpr) acc = fma((double)val[k], (double)x[idx[k]], acc);
    for (int off = lpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);   // fixed tree inside the lane group
    if (l == 0) res[row] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)acc, beta, BETA0 ? T(0) : res[row]);
  }
}

}  // namespace mxlo
