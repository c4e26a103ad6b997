// sparse.hip — LinearOperator(M::SparseMatrixCSC) prod!/tprod!/ctprod! (src/constructors.jl:19-29 hands M to
// LinearAlgebra.mul!, i.e. to the SparseArrays stdlib) and the sparse blocks of BlockDiagonalOperator
// (test/test_linop.jl:743-756 builds one from an operator, a Matrix and a sprand block).
// Kernel design: sparse_kernels.h. This file: the handle (compressed-row view and chunk tables built once at
// construction, values permuted into row order), refresh after in-place value updates, the apply.
#include <algorithm>
#include <numeric>
#include <vector>

#include "common.h"
#include "sparse_kernels.h"

using namespace mxlo;

struct mxlo_csc {
  mxlo_ctx *ctx = nullptr;
  int dtype = 0;
  int64_t m = 0, n = 0, nnz = 0;
  CscDev host{};            // device pointers + sizes (mirror of *dev)
  CscDev *dev = nullptr;    // device-resident copy (what MXLO_BLK_CSC blocks of a fused block-diagonal point at)
  int32_t *perm = nullptr;  // [nnz] CSR position -> CSC position (mxlo_csc_refresh)
  const void *nzval = nullptr;
  SpLongRow *long_n = nullptr, *long_t = nullptr;   // rows cut into pieces (fix-up launch), per mode
};

namespace mxlo {
const CscDev *csc_device_desc(const mxlo_csc *h) { return h ? h->dev : nullptr; }   // blockdiag.hip
const mxlo_ctx *csc_ctx(const mxlo_csc *h) { return h ? h->ctx : nullptr; }
void csc_shape(const mxlo_csc *h, int64_t *m, int64_t *n, int *dtype, int *nchunks_n, int *nchunks_t, int *nlong) {
  *m = h->m; *n = h->n; *dtype = h->dtype;
  *nchunks_n = h->host.nchunks_n; *nchunks_t = h->host.nchunks_t; *nlong = h->host.nlong_n + h->host.nlong_t;
}
}  // namespace mxlo

namespace {

template <typename T>
__global__ void __launch_bounds__(kBlock)
csc_gather_values_kernel(T *__restrict__ out, const T *__restrict__ nzval, const int32_t *__restrict__ perm, int64_t nnz) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nnz) out[i] = nzval[perm[i]];
}

template <typename T, typename FIN, bool BETA0, bool BLOCK, bool CONJ>
__global__ void __launch_bounds__(kBlock)
csc_mul_kernel(T *__restrict__ res, const T *__restrict__ x, const int64_t *__restrict__ ptr, const int32_t *__restrict__ idx,
               const T *__restrict__ val, const SpChunk *__restrict__ chunks, int nchunks, int nxcd, double *__restrict__ carry,
               const FIN fin, int ncols, int64_t ldx, int64_t ldr) {
  __shared__ __attribute__((aligned(16))) char lds[sp_lds_bytes<T>()];
  // XCD-aware order: workgroup w runs on XCD w % nxcd, and each XCD has its own L2. XCD k takes the k-th CONTIGUOUS
  // part of the chunk table, so the x lines its workgroups share (neighbouring rows gather neighbouring columns) are
  // fetched into ONE L2 instead of into all eight.
  const int w = (int)blockIdx.x, per = (nchunks + nxcd - 1) / nxcd;
  const int ci = (w % nxcd) * per + w / nxcd;
  if (ci >= nchunks || w / nxcd >= per) return;
  spmv_chunk<T, FIN, BETA0, BLOCK, CONJ>(res, x, ptr, idx, val, chunks[ci], carry, fin, lds, ncols, ldx, ldr);
}

// The chunk table of one compressed-row operand (see sparse_kernels.h): whole rows packed into chunks of <= kSpChunk
// entries (and <= kSpChunk rows: empty rows still need their beta step), rows above kSpLongRow entries on their own.
void build_chunks(const std::vector<int64_t> &ptr, int64_t nrows, std::vector<SpChunk> &chunks, std::vector<SpLongRow> &longs,
                  int *ncarry) {
  SpChunk cur{0, 0, 0, 0, SP_ROWS, 0, 0};
  auto flush = [&]() {
    if (cur.nr > 0) chunks.push_back(cur);
    cur.nz = 0;
    cur.nr = 0;
  };
  int carry = 0;
  for (int64_t r = 0; r < nrows; ++r) {
    const int64_t len = ptr[r + 1] - ptr[r];
    if (len > kSpLongRow) {
      flush();
      if (len <= kSpChunk) {
        chunks.push_back(SpChunk{ptr[r], (int32_t)len, (int32_t)r, 1, SP_LONG1, 0, 0});
      } else {
        SpLongRow lr{(int32_t)r, carry, 0, 0};
        for (int64_t k = ptr[r]; k < ptr[r + 1]; k += kSpChunk) {
          const int64_t piece = std::min<int64_t>(kSpChunk, ptr[r + 1] - k);
          chunks.push_back(SpChunk{k, (int32_t)piece, (int32_t)r, 1, SP_PIECE, carry++, 0});
          ++lr.npieces;
        }
        longs.push_back(lr);
      }
      continue;
    }
    if (cur.nr > 0 && (cur.nz + len > kSpChunk || cur.nr == kSpChunk)) flush();
    if (cur.nr == 0) {
      cur.k0 = ptr[r];
      cur.row0 = (int32_t)r;
    }
    cur.nz += (int32_t)len;
    ++cur.nr;
  }
  flush();
  *ncarry = carry;
}

constexpr int kSpMaxCols = 8;      // columns of a block apply per launch (carry slots are sized for it)

struct SpLaunch {                  // what one apply needs from the handle, by mode
  const int64_t *ptr;
  const int32_t *idx;
  const void *val;
  const SpChunk *chunks;
  const SpLongRow *longs;
  int nchunks, nlong;
};
SpLaunch sp_launch_of(const mxlo_csc *h, bool trans) {
  const CscDev &d = h->host;
  return trans ? SpLaunch{d.colptr, d.rowidx, d.nzval, d.chunks_t, h->long_t, d.nchunks_t, d.nlong_t}
               : SpLaunch{d.rowptr, d.colidx, d.csr_val, d.chunks_n, h->long_n, d.nchunks_n, d.nlong_n};
}

template <typename T, typename FIN, bool B0, bool CONJ>
int32_t csc_launch(mxlo_csc *h, const SpLaunch &L, T *res, const T *v, const FIN &fin, int ncols, int64_t ldx, int64_t ldr) {
  mxlo_ctx *ctx = h->ctx;
  const int nxcd = ctx->tune.sp_xcds > 0 ? ctx->tune.sp_xcds : 1;
  const int per = (L.nchunks + nxcd - 1) / nxcd;
  if (ncols > 1)
    hipLaunchKernelGGL((csc_mul_kernel<T, FIN, B0, true, CONJ>), dim3((unsigned)(per * nxcd)), dim3(kBlock), 0, ctx->stream, res, v,
                       L.ptr, L.idx, (const T *)L.val, L.chunks, L.nchunks, nxcd, h->host.carry, fin, ncols, ldx, ldr);
  else
    hipLaunchKernelGGL((csc_mul_kernel<T, FIN, B0, false, CONJ>), dim3((unsigned)(per * nxcd)), dim3(kBlock), 0, ctx->stream, res, v,
                       L.ptr, L.idx, (const T *)L.val, L.chunks, L.nchunks, nxcd, h->host.carry, fin, 1, ldx, ldr);
  MXLO_LAUNCH_CHECK();
  if (L.nlong > 0) {
    hipLaunchKernelGGL((spmv_fixup_kernel<T, FIN, B0>), dim3((unsigned)((L.nlong * ncols + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       ctx->stream, res, h->host.carry, L.longs, L.nlong, fin, ncols, ldr);
    MXLO_LAUNCH_CHECK();
  }
  return MXLO_OK;
}

template <typename T>
int32_t csc_mul_t(mxlo_csc *h, T *res, const T *v, double alpha, double beta, int32_t op_mode, int32_t flags, int ncols = 1,
                  int64_t ldx = 0, int64_t ldr = 0) {
  const SpLaunch L = sp_launch_of(h, op_mode != MXLO_OP_N);
  if (L.nchunks == 0) return MXLO_OK;
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    return csc_launch<T, SpFinReal<T, CA, CB, B0>, B0, false>(h, L, res, v, SpFinReal<T, CA, CB, B0>{(CA)alpha, (CB)beta}, ncols, ldx,
                                                              ldr);
  });
}

// complex element types: op_mode N / T walk the values as stored, C conjugates them (A' * x)
template <typename R>
int32_t csc_mul_c_t(mxlo_csc *h, cx<R> *res, const cx<R> *v, const ScalArgs &s, int32_t op_mode) {
  const SpLaunch L = sp_launch_of(h, op_mode != MXLO_OP_N);
  if (L.nchunks == 0) return MXLO_OK;
  return dispatch_c<R>(s, [&]<typename RA, typename RB, bool B0>() -> int32_t {
    const SpFinCplx<R, RA, RB, B0> fin{Sc<RA>{(RA)s.are, (RA)s.aim, s.a_real}, (RB)s.bre, (RB)s.bim, s.b_real};
    if (op_mode == MXLO_OP_C) return csc_launch<cx<R>, SpFinCplx<R, RA, RB, B0>, B0, true>(h, L, res, v, fin, 1, 0, 0);
    return csc_launch<cx<R>, SpFinCplx<R, RA, RB, B0>, B0, false>(h, L, res, v, fin, 1, 0, 0);
  });
}

}  // namespace

MXLO_API int32_t mxlo_csc_create(mxlo_ctx *ctx, int32_t dtype, int64_t m, int64_t n, const int64_t *colptr,
                                 const int64_t *rowval, const void *nzval, int32_t index_base, mxlo_csc **out) {
  MXLO_REQUIRE(ctx && out, MXLO_EINVAL, "mxlo_csc_create: NULL argument");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(dtype >= MXLO_F64 && dtype <= MXLO_C32, MXLO_EINVAL, "mxlo_csc_create: bad dtype %d", dtype);
  MXLO_REQUIRE(m >= 0 && n >= 0 && m < (1LL << 31) && n < (1LL << 31), MXLO_ESHAPE,
               "mxlo_csc_create: %lld x %lld (each dimension must be below 2^31: indices are 32-bit inside the library)",
               (long long)m, (long long)n);
  MXLO_REQUIRE(index_base == 0 || index_base == 1, MXLO_EINVAL, "mxlo_csc_create: index_base must be 0 or 1 (Julia)");
  MXLO_REQUIRE(colptr, MXLO_EINVAL, "mxlo_csc_create: colptr is NULL");
  const size_t es = dtype == MXLO_F64 || dtype == MXLO_C32 ? 8 : (dtype == MXLO_C64 ? 16 : 4);
  // ---- structure to the host, validated there (the reference's SparseMatrixCSC constructor checks the same invariants)
  std::vector<int64_t> cp((size_t)n + 1);
  MXLO_HIP(hipStreamSynchronize(ctx->stream));
  MXLO_HIP(hipMemcpy(cp.data(), colptr, sizeof(int64_t) * cp.size(), hipMemcpyDeviceToHost));
  for (auto &c : cp) c -= index_base;
  MXLO_REQUIRE(cp[0] == 0, MXLO_EINVAL, "mxlo_csc_create: colptr[0] must be %d", index_base);
  for (int64_t j = 0; j < n; ++j)
    MXLO_REQUIRE(cp[j + 1] >= cp[j], MXLO_EINVAL, "mxlo_csc_create: colptr decreases at column %lld", (long long)j);
  const int64_t nnz = cp[n];
  MXLO_REQUIRE(nnz < (1LL << 31), MXLO_ESHAPE, "mxlo_csc_create: %lld stored entries (the row-order permutation is 32-bit)",
               (long long)nnz);
  MXLO_REQUIRE(nnz == 0 || (rowval && nzval), MXLO_EINVAL, "mxlo_csc_create: rowval / nzval is NULL");
  std::vector<int64_t> rv((size_t)nnz);
  if (nnz) MXLO_HIP(hipMemcpy(rv.data(), rowval, sizeof(int64_t) * (size_t)nnz, hipMemcpyDeviceToHost));
  std::vector<int32_t> ri((size_t)nnz);
  std::vector<int64_t> rp((size_t)m + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t r = rv[k] - index_base;
    MXLO_REQUIRE(r >= 0 && r < m, MXLO_EINVAL, "mxlo_csc_create: row index %lld at position %lld outside %d..%lld",
                 (long long)rv[k], (long long)k, (int)index_base, (long long)(m - 1 + index_base));
    ri[k] = (int32_t)r;
    ++rp[r + 1];
  }
  std::partial_sum(rp.begin(), rp.end(), rp.begin());
  // ---- compressed-row view: a stable counting sort by row keeps the columns of a row ascending (fixed summation order)
  std::vector<int32_t> ci((size_t)nnz), perm((size_t)nnz);
  {
    std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
    for (int64_t j = 0; j < n; ++j)
      for (int64_t k = cp[j]; k < cp[j + 1]; ++k) {
        const int64_t p = fill[ri[k]]++;
        ci[p] = (int32_t)j;
        perm[p] = (int32_t)k;
      }
  }
  // ---- chunk tables of the two sweeps
  std::vector<SpChunk> chn, cht;
  std::vector<SpLongRow> lgn, lgt;
  int ncar_n = 0, ncar_t = 0;
  build_chunks(rp, m, chn, lgn, &ncar_n);
  build_chunks(cp, n, cht, lgt, &ncar_t);
  MXLO_REQUIRE(chn.size() < (1u << 31) && cht.size() < (1u << 31), MXLO_ESHAPE, "mxlo_csc_create: too many chunks");

  mxlo_csc *h = new mxlo_csc();
  h->ctx = ctx;
  h->dtype = dtype;
  h->m = m;
  h->n = n;
  h->nnz = nnz;
  h->nzval = nzval;
  hipError_t e = hipSuccess;
  auto up = [&](void **dst, const void *src, size_t bytes) {
    if (e != hipSuccess) return;
    e = hipMalloc(dst, bytes ? bytes : 16);
    if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  CscDev &d = h->host;
  up((void **)&d.rowptr, rp.data(), sizeof(int64_t) * rp.size());
  up((void **)&d.colidx, ci.data(), sizeof(int32_t) * (size_t)nnz);
  up((void **)&d.colptr, cp.data(), sizeof(int64_t) * cp.size());
  up((void **)&d.rowidx, ri.data(), sizeof(int32_t) * (size_t)nnz);
  up((void **)&h->perm, perm.data(), sizeof(int32_t) * (size_t)nnz);
  up((void **)&d.chunks_n, chn.data(), sizeof(SpChunk) * chn.size());
  up((void **)&d.chunks_t, cht.data(), sizeof(SpChunk) * cht.size());
  up((void **)&h->long_n, lgn.data(), sizeof(SpLongRow) * lgn.size());
  up((void **)&h->long_t, lgt.data(), sizeof(SpLongRow) * lgt.size());
  if (e == hipSuccess) e = hipMalloc((void **)&d.csr_val, nnz ? es * (size_t)nnz : 16);
  if (e == hipSuccess) e = hipMalloc((void **)&d.carry, sizeof(double) * 2 * kSpMaxCols * (size_t)std::max(1, std::max(ncar_n, ncar_t)));
  d.nzval = nzval;
  d.m = m;
  d.n = n;
  d.nnz = nnz;
  d.nchunks_n = (int32_t)chn.size();
  d.nchunks_t = (int32_t)cht.size();
  d.nlong_n = (int32_t)lgn.size();
  d.nlong_t = (int32_t)lgt.size();
  if (e == hipSuccess) e = hipMalloc((void **)&h->dev, sizeof(CscDev));
  if (e == hipSuccess) e = hipMemcpy(h->dev, &d, sizeof(CscDev), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_error("mxlo_csc_create: device allocation failed: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    mxlo_csc_destroy(h);
    return MXLO_ENOMEM;
  }
  const int32_t st = mxlo_csc_refresh(h);
  if (st != MXLO_OK) {
    mxlo_csc_destroy(h);
    return st;
  }
  *out = h;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_csc_refresh(mxlo_csc *h) {
  MXLO_REQUIRE(h, MXLO_EINVAL, "mxlo_csc_refresh: handle is NULL");
  MXLO_DEVICE_GUARD(h->ctx);
  if (h->nnz == 0) return MXLO_OK;
  const unsigned grid = (unsigned)((h->nnz + kBlock - 1) / kBlock);
  if (h->dtype == MXLO_C64)            // 16-byte elements
    hipLaunchKernelGGL(csc_gather_values_kernel<cx<double>>, dim3(grid), dim3(kBlock), 0, h->ctx->stream,
                       (cx<double> *)h->host.csr_val, (const cx<double> *)h->nzval, h->perm, h->nnz);
  else if (h->dtype == MXLO_F64 || h->dtype == MXLO_C32)     // 8-byte elements
    hipLaunchKernelGGL(csc_gather_values_kernel<double>, dim3(grid), dim3(kBlock), 0, h->ctx->stream, (double *)h->host.csr_val,
                       (const double *)h->nzval, h->perm, h->nnz);
  else
    hipLaunchKernelGGL(csc_gather_values_kernel<float>, dim3(grid), dim3(kBlock), 0, h->ctx->stream, (float *)h->host.csr_val,
                       (const float *)h->nzval, h->perm, h->nnz);
  MXLO_LAUNCH_CHECK();
  return MXLO_OK;
}

MXLO_API int32_t mxlo_csc_mul(mxlo_csc *h, void *res, const void *v, double alpha, double beta, int32_t op_mode,
                              int32_t flags) {
  MXLO_REQUIRE(h, MXLO_EINVAL, "mxlo_csc_mul: handle is NULL");
  MXLO_DEVICE_GUARD(h->ctx);
  MXLO_REQUIRE(op_mode >= MXLO_OP_N && op_mode <= MXLO_OP_C, MXLO_EINVAL, "bad op_mode");
  const int64_t nres = op_mode == MXLO_OP_N ? h->m : h->n, nin = op_mode == MXLO_OP_N ? h->n : h->m;
  if (nres == 0) return MXLO_OK;
  MXLO_REQUIRE(res && (v || nin == 0), MXLO_EINVAL, "mxlo_csc_mul: NULL operand");
  MXLO_REQUIRE(h->dtype == MXLO_F64 || h->dtype == MXLO_F32, MXLO_EINVAL,
               "mxlo_csc_mul: the handle holds a complex matrix: use mxlo_csc_mul_c");
  eff_scalars(h->dtype == MXLO_F64 ? 8 : 4, flags, alpha, beta);
  if (h->dtype == MXLO_F64) return csc_mul_t<double>(h, (double *)res, (const double *)v, alpha, beta, op_mode, flags);
  return csc_mul_t<float>(h, (float *)res, (const float *)v, alpha, beta, op_mode, flags);
}

MXLO_API int32_t mxlo_csc_mul_c(mxlo_csc *h, void *res, const void *v, double alpha_re, double alpha_im, double beta_re,
                                double beta_im, int32_t op_mode, int32_t flags) {
  MXLO_REQUIRE(h, MXLO_EINVAL, "mxlo_csc_mul_c: handle is NULL");
  MXLO_DEVICE_GUARD(h->ctx);
  MXLO_REQUIRE(op_mode >= MXLO_OP_N && op_mode <= MXLO_OP_C, MXLO_EINVAL, "bad op_mode");
  MXLO_REQUIRE(h->dtype == MXLO_C64 || h->dtype == MXLO_C32, MXLO_EINVAL,
               "mxlo_csc_mul_c: the handle holds a real matrix: use mxlo_csc_mul (a real operator on complex vectors is applied to "
               "the two planes, mxlo_split_c / mxlo_join_c)");
  const int64_t nres = op_mode == MXLO_OP_N ? h->m : h->n, nin = op_mode == MXLO_OP_N ? h->n : h->m;
  if (nres == 0) return MXLO_OK;
  MXLO_REQUIRE(res && (v || nin == 0), MXLO_EINVAL, "mxlo_csc_mul_c: NULL operand");
  if (h->dtype == MXLO_C64)
    return csc_mul_c_t<double>(h, (cx<double> *)res, (const cx<double> *)v, scal_args(8, alpha_re, alpha_im, beta_re, beta_im, flags),
                               op_mode);
  return csc_mul_c_t<float>(h, (cx<float> *)res, (const cx<float> *)v, scal_args(4, alpha_re, alpha_im, beta_re, beta_im, flags),
                            op_mode);
}

MXLO_API int32_t mxlo_csc_mul_block(mxlo_csc *h, void *res, int64_t ldr, const void *V, int64_t ldv, int64_t k, double alpha,
                                    double beta, int32_t op_mode, int32_t flags) {
  MXLO_REQUIRE(h, MXLO_EINVAL, "mxlo_csc_mul_block: handle is NULL");
  MXLO_DEVICE_GUARD(h->ctx);
  MXLO_REQUIRE(op_mode >= MXLO_OP_N && op_mode <= MXLO_OP_C, MXLO_EINVAL, "bad op_mode");
  const int64_t nres = op_mode == MXLO_OP_N ? h->m : h->n, nin = op_mode == MXLO_OP_N ? h->n : h->m;
  MXLO_REQUIRE(k >= 0, MXLO_ESHAPE, "mxlo_csc_mul_block: k < 0");
  if (nres == 0 || k == 0) return MXLO_OK;
  MXLO_REQUIRE(res && (V || nin == 0), MXLO_EINVAL, "mxlo_csc_mul_block: NULL operand");
  MXLO_REQUIRE(ldr >= nres && ldv >= (nin > 0 ? nin : 1), MXLO_ESHAPE, "mxlo_csc_mul_block: leading dimension below the column length");
  MXLO_REQUIRE(h->dtype == MXLO_F64 || h->dtype == MXLO_F32, MXLO_EINVAL, "mxlo_csc_mul_block: real element types (a complex block is applied column by column)");
  eff_scalars(h->dtype == MXLO_F64 ? 8 : 4, flags, alpha, beta);
  for (int64_t j0 = 0; j0 < k; j0 += kSpMaxCols) {                      // the stored matrix is read once per 8 columns
    const int nc = (int)std::min<int64_t>(kSpMaxCols, k - j0);
    int32_t st;
    if (h->dtype == MXLO_F64)
      st = csc_mul_t<double>(h, (double *)res + j0 * ldr, (const double *)V + j0 * ldv, alpha, beta, op_mode, flags, nc, ldv, ldr);
    else
      st = csc_mul_t<float>(h, (float *)res + j0 * ldr, (const float *)V + j0 * ldv, alpha, beta, op_mode, flags, nc, ldv, ldr);
    if (st != MXLO_OK) return st;
  }
  return MXLO_OK;
}

// Host-only view of the work decomposition (no device needed): the chunk table build_chunks() produces for a row-pointer
// array, as rows of {k0, nz, row0, nr, kind, carry}. What the CPU tests check the invariants of (every stored entry in
// exactly one chunk, rows whole unless longer than a chunk, pieces of a long row contiguous and in order).
MXLO_API int32_t mxlo_debug_csc_chunks(const int64_t *ptr_host, int64_t nrows, int64_t *out, int64_t cap, int64_t *nchunks,
                                       int64_t *nlong, int64_t *ncarry) {
  MXLO_REQUIRE(ptr_host && nchunks && nlong && ncarry && nrows >= 0, MXLO_EINVAL, "mxlo_debug_csc_chunks: bad argument");
  std::vector<int64_t> ptr(ptr_host, ptr_host + nrows + 1);
  std::vector<SpChunk> chunks;
  std::vector<SpLongRow> longs;
  int nc = 0;
  build_chunks(ptr, nrows, chunks, longs, &nc);
  *nchunks = (int64_t)chunks.size();
  *nlong = (int64_t)longs.size();
  *ncarry = nc;
  if (out)
    for (int64_t i = 0; i < (int64_t)chunks.size() && i < cap; ++i) {
      const SpChunk &c = chunks[(size_t)i];
      int64_t *o = out + 6 * i;
      o[0] = c.k0; o[1] = c.nz; o[2] = c.row0; o[3] = c.nr; o[4] = c.kind; o[5] = c.carry;
    }
  return MXLO_OK;
}

MXLO_API int32_t mxlo_csc_info(mxlo_csc *h, int64_t info[8]) {
  MXLO_REQUIRE(h && info, MXLO_EINVAL, "mxlo_csc_info: NULL argument");
  info[0] = h->m;
  info[1] = h->n;
  info[2] = h->nnz;
  info[3] = h->host.nchunks_n;
  info[4] = h->host.nchunks_t;
  info[5] = h->host.nlong_n;
  info[6] = h->host.nlong_t;
  info[7] = kSpChunk;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_csc_destroy(mxlo_csc *h) {
  if (!h) return MXLO_OK;
  MXLO_DEVICE_GUARD(h->ctx);
  (void)hipStreamSynchronize(h->ctx->stream);
  for (const void *p : {(const void *)h->host.rowptr, (const void *)h->host.colidx, h->host.csr_val,
                        (const void *)h->host.colptr, (const void *)h->host.rowidx, (const void *)h->perm,
                        (const void *)h->host.chunks_n, (const void *)h->host.chunks_t, (const void *)h->host.carry,
                        (const void *)h->long_n, (const void *)h->long_t, (const void *)h->dev})
    if (p) (void)hipFree(const_cast<void *>(p));
  delete h;
  return MXLO_OK;
}
