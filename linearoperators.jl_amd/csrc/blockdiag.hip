// blockdiag.hip — BlockDiagonalOperator prod!/tprod!/ctprod! (src/special-operators.jl:249-294)
// as ONE launch: a device-resident descriptor table (one entry per block) plus a tile table
// (one workgroup per tile) built once at construction. The reference issues one inner mul! per
// block (1024 launches at BASELINE config 4); here the launch count is 1 regardless of the
// number of blocks.
//
// Tiles: elementwise blocks (opDiagonal / opEye / opZeros) are cut into 2048-row tiles; a dense
// block contributes 256-output-row tiles in N mode (thread per row, coalesced down the columns)
// and 4-output tiles in T mode (one wave per output, coalesced down the column).
// A sparse block (MXLO_BLK_CSC: data = the device descriptor of an mxlo_csc handle) contributes one tile per CHUNK of its
// compressed-row sweep (sparse_kernels.h: <= 2048 stored entries streamed into LDS, rows walked from there) — the N-mode
// tiles walk the CSR view, the T-mode tiles the CSC arrays themselves. Operators with sparse blocks are launched with
// 24 KiB of dynamic LDS; the others with none (the elementwise tiles keep their occupancy).
// Blocks start at arbitrary row offsets (e.g. 97,657-row blocks), so each tile aligns its
// stores to 16 bytes by peeling and loads an operand with one 16-byte or two element accesses
// depending on that operand's own phase.
#include <algorithm>
#include <vector>

#include "common.h"
#include "stream_kernels.h"
#include "sparse_kernels.h"

using namespace mxlo;

struct mxlo_csc;
namespace mxlo {
const CscDev *csc_device_desc(const mxlo_csc *h);     // sparse.hip
void csc_shape(const mxlo_csc *h, int64_t *m, int64_t *n, int *dtype, int *nchunks_n, int *nchunks_t, int *nlong);
const mxlo_ctx *csc_ctx(const mxlo_csc *h);
}

namespace {

constexpr int kTileE = 2048;   // rows per elementwise tile
constexpr int kTileDN = 256;   // output rows per dense N-mode tile
constexpr int kTileDT = 4;     // outputs per dense T-mode tile (one per wave)

struct DevBlock {
  int32_t kind, pad;
  int64_t row_off, col_off, m, n;
  const void *data;
  int64_t ld;
};

struct Tile {        // 64 bytes, self-contained: ONE wave-uniform load per workgroup, no dependent
  DevBlock b;        // descriptor fetch before the streaming loads can start
  int64_t start;     // first output row of this tile inside its block
  int64_t cnt;       // outputs in this tile (elementwise tiles end on GLOBAL multiples of kTileE so that,
};                   // after the first short tile of a block, every tile is 16 KiB-aligned in res)

template <typename T, int VEC, bool NT>
__device__ __forceinline__ void load_vec(const T *p, bool aligned, T (&out)[VEC]) {
  using V = typename VecOf<T, VEC>::type;
  if (aligned) {
    const V v = ldg<NT>(reinterpret_cast<const V *>(p));
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[e] = v[e];
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[e] = p[e];
  }
}

// kind-specific scalar op: returns the new res element
template <typename T, typename CA, typename CB, bool BETA0>
__device__ __forceinline__ T ew_op(int kind, CA a, CB b, T d, T v, T r) {
  CA t;
  if (kind == MXLO_BLK_DIAG) t = (a * (CA)d) * (CA)v;          // special-operators.jl:127-129
  else if (kind == MXLO_BLK_EYE) t = a * (CA)v;                 // :38-41
  else {                                                        // zeros: res .= 0 | res .*= β (:104-106)
    if constexpr (BETA0) return (T)0;
    else return (T)((CB)r * b);
  }
  return fin_ab<T, CA, CB, BETA0>(t, b, r);
}

// One elementwise tile of a block of compile-time KIND (no per-load kind / alignment branches in the
// all-aligned fast path, so the compiler can issue every load of the tile back to back).
template <typename T, typename CA, typename CB, bool BETA0, bool NT, int KIND>
__device__ __forceinline__ void ew_tile(T *__restrict__ r0, const T *__restrict__ x0, const T *__restrict__ d0,
                                        int64_t cnt, int64_t tile_start, int64_t nmin, CA alpha, CB beta) {
  constexpr int VEC = Vec16<T>::N;
  using V = typename VecOf<T, VEC>::type;
  const int tid = threadIdx.x;
  // peel so stores are 16-byte aligned
  int64_t head = (int64_t)(((16 - ((uintptr_t)r0 & 15u)) & 15u) / sizeof(T));
  if (head > cnt) head = cnt;
  const bool fast = (tile_start + cnt <= nmin);   // whole tile inside the diagonal part
  if (fast) {
    const int64_t nv = (cnt - head) / VEC;
    const bool xa = KIND == MXLO_BLK_ZEROS || (((uintptr_t)(x0 + head)) & 15u) == 0;
    const bool da = KIND != MXLO_BLK_DIAG || (((uintptr_t)(d0 + head)) & 15u) == 0;
    constexpr int U = kTileE / VEC / kBlock;
    T dv[U][VEC], xv[U][VEC], rv[U][VEC];
    if (xa && da && head == 0 && cnt == kTileE) {
      // a whole, 16-byte aligned tile (all but the two boundary tiles of a block): no per-load predicates at all
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t o = ((int64_t)tid + (int64_t)u * kBlock) * VEC;
        if constexpr (KIND == MXLO_BLK_DIAG) load_vec<T, VEC, NT>(d0 + o, true, dv[u]);
        if constexpr (KIND != MXLO_BLK_ZEROS) load_vec<T, VEC, NT>(x0 + o, true, xv[u]);
        if constexpr (!BETA0) load_vec<T, VEC, NT>(r0 + o, true, rv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t o = ((int64_t)tid + (int64_t)u * kBlock) * VEC;
        V out;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          out[e] = ew_op<T, CA, CB, BETA0>(KIND, alpha, beta, KIND == MXLO_BLK_DIAG ? dv[u][e] : T(0),
                                       KIND != MXLO_BLK_ZEROS ? xv[u][e] : T(0), BETA0 ? T(0) : rv[u][e]);
        stg<NT>(reinterpret_cast<V *>(r0 + o), out);
      }
      return;
    }
    if (xa && da) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = tid + (int64_t)u * kBlock;
        if (i < nv) {
          const int64_t o = head + i * VEC;
          if constexpr (KIND == MXLO_BLK_DIAG) load_vec<T, VEC, NT>(d0 + o, true, dv[u]);
          if constexpr (KIND != MXLO_BLK_ZEROS) load_vec<T, VEC, NT>(x0 + o, true, xv[u]);
          if constexpr (!BETA0) load_vec<T, VEC, NT>(r0 + o, true, rv[u]);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = tid + (int64_t)u * kBlock;
        if (i < nv) {
          const int64_t o = head + i * VEC;
          if constexpr (KIND == MXLO_BLK_DIAG) load_vec<T, VEC, NT>(d0 + o, da, dv[u]);
          if constexpr (KIND != MXLO_BLK_ZEROS) load_vec<T, VEC, NT>(x0 + o, xa, xv[u]);
          if constexpr (!BETA0) load_vec<T, VEC, NT>(r0 + o, true, rv[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = tid + (int64_t)u * kBlock;
      if (i < nv) {
        const int64_t o = head + i * VEC;
        V out;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          out[e] = ew_op<T, CA, CB, BETA0>(KIND, alpha, beta, KIND == MXLO_BLK_DIAG ? dv[u][e] : T(0),
                                       KIND != MXLO_BLK_ZEROS ? xv[u][e] : T(0), BETA0 ? T(0) : rv[u][e]);
        stg<NT>(reinterpret_cast<V *>(r0 + o), out);
      }
    }
    const int64_t tail0 = head + nv * VEC;
    const int64_t nsc = head + (cnt - tail0);
    if (tid < nsc) {
      const int64_t o = tid < head ? tid : tail0 + (tid - head);
      r0[o] = ew_op<T, CA, CB, BETA0>(KIND, alpha, beta, KIND == MXLO_BLK_DIAG ? d0[o] : T(0),
                                  KIND != MXLO_BLK_ZEROS ? x0[o] : T(0), BETA0 ? T(0) : r0[o]);
    }
  } else {
    for (int64_t o = tid; o < cnt; o += kBlock) {
      const int64_t g = tile_start + o;
      if (g < nmin) {
        r0[o] = ew_op<T, CA, CB, BETA0>(KIND, alpha, beta, KIND == MXLO_BLK_DIAG ? d0[o] : T(0),
                                    KIND != MXLO_BLK_ZEROS ? x0[o] : T(0), BETA0 ? T(0) : r0[o]);
      } else {  // rectangular eye tail: 0 when β == 0 else β itself (special-operators.jl:39,42)
        r0[o] = BETA0 ? (T)0 : (T)beta;
      }
    }
  }
}

// SP: the operator has sparse blocks. A separate instantiation, so that operators without them keep the 40 registers
// (8 waves per SIMD) of the elementwise / dense tiles — the chunk sweep needs 100.
template <typename T, typename CA, typename CB, bool BETA0, bool TRANS, bool NT, bool SP>
__global__ void __launch_bounds__(kBlock)
blockdiag_kernel(T *__restrict__ res, const T *__restrict__ x, const DevBlock *__restrict__ blocks,
                 const Tile *__restrict__ tiles, CA alpha, CB beta) {
  const Tile tl = tiles[blockIdx.x];
  const DevBlock b = tl.b;
  (void)blocks;
  const int tid = threadIdx.x;
  // in T mode the roles of (row_off, m) and (col_off, n) swap
  const int64_t out_off = TRANS ? b.col_off : b.row_off;
  const int64_t in_off = TRANS ? b.row_off : b.col_off;
  const int64_t mo = TRANS ? b.n : b.m;   // outputs of this block
  const int64_t ni = TRANS ? b.m : b.n;   // inputs of this block
  T *rp = res + out_off;
  const T *xp = x + in_off;
  if constexpr (SP) if (b.kind == MXLO_BLK_CSC) {
    extern __shared__ double sp_lds[];                        // sp_lds_bytes<T>(): the chunk's values and indices
    const CscDev *S = (const CscDev *)b.data;                 // wave-uniform descriptor fetches
    const SpFinReal<T, CA, CB, BETA0> fin{alpha, beta};
    if constexpr (!TRANS)
      spmv_chunk<T, SpFinReal<T, CA, CB, BETA0>, BETA0>(rp, xp, S->rowptr, S->colidx, (const T *)S->csr_val, S->chunks_n[tl.start],
                                                        S->carry, fin, sp_lds);
    else
      spmv_chunk<T, SpFinReal<T, CA, CB, BETA0>, BETA0>(rp, xp, S->colptr, S->rowidx, (const T *)S->nzval, S->chunks_t[tl.start],
                                                        S->carry, fin, sp_lds);
    return;
  }
  if (b.kind != MXLO_BLK_DENSE) {
    // elementwise tile [tl.start, tl.start + cnt). Rectangular eye/zeros: outputs beyond
    // min(m,n) of an eye block follow mulOpEye!'s tail rule (0 | β).
    const int64_t cnt = tl.cnt;
    const int64_t nmin = b.kind == MXLO_BLK_ZEROS ? mo : (mo < ni ? mo : ni);
    const T *dp = (const T *)b.data;
    T *r0 = rp + tl.start;
    const T *x0 = xp + tl.start;
    const T *d0 = dp ? dp + tl.start : nullptr;
    if (b.kind == MXLO_BLK_DIAG)
      ew_tile<T, CA, CB, BETA0, NT, MXLO_BLK_DIAG>(r0, x0, d0, cnt, tl.start, nmin, alpha, beta);
    else if (b.kind == MXLO_BLK_EYE)
      ew_tile<T, CA, CB, BETA0, NT, MXLO_BLK_EYE>(r0, x0, d0, cnt, tl.start, nmin, alpha, beta);
    else
      ew_tile<T, CA, CB, BETA0, NT, MXLO_BLK_ZEROS>(r0, x0, d0, cnt, tl.start, nmin, alpha, beta);
    return;
  }
  // ---- dense block, column-major m x n with leading dimension ld
  const T *M = (const T *)b.data;
  if constexpr (!TRANS) {
    const int64_t i = tl.start + tid;
    if (i < b.m) {
      double acc = 0.0;
      for (int64_t j = 0; j < b.n; ++j) acc = fma((double)M[i + j * b.ld], (double)xp[j], acc);
      rp[i] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)acc, beta, BETA0 ? T(0) : rp[i]);
    }
  } else {
    const int lane = tid & 63;
    const int64_t j = tl.start + (tid >> 6);
    if (j < b.n) {
      const T *colp = M + j * b.ld;
      double acc = 0.0;
      for (int64_t i = lane; i < b.m; i += 64) acc = fma((double)colp[i], (double)xp[i], acc);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
      if (lane == 0) {
        rp[j] = fin_ab<T, CA, CB, BETA0>(alpha * (CA)(T)acc, beta, BETA0 ? T(0) : rp[j]);
      }
    }
  }
}

}  // namespace

struct mxlo_blockdiag {
  mxlo_ctx *ctx = nullptr;
  int dtype = 0;
  int64_t nblocks = 0, nrow = 0, ncol = 0;
  DevBlock *d_blocks = nullptr;
  Tile *d_tiles_n = nullptr, *d_tiles_t = nullptr;
  int64_t ntiles_n = 0, ntiles_t = 0;
  size_t lds_bytes = 0;     // dynamic LDS of a launch: the product buffer of the sparse tiles, 0 without sparse blocks
};

MXLO_API int32_t mxlo_blockdiag_create(mxlo_ctx *ctx, int32_t dtype, const mxlo_block_desc *blocks,
                                       int64_t nblocks, mxlo_blockdiag **out) {
  MXLO_REQUIRE(ctx && out && (nblocks == 0 || blocks), MXLO_EINVAL, "mxlo_blockdiag_create: NULL argument");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(dtype == MXLO_F64 || dtype == MXLO_F32, MXLO_EINVAL, "bad dtype");
  MXLO_REQUIRE(nblocks >= 0 && nblocks < (1LL << 31), MXLO_ESHAPE, "bad block count");
  std::vector<DevBlock> hb((size_t)nblocks);
  std::vector<Tile> tn, tt;
  int64_t nrow = 0, ncol = 0;
  bool has_sparse = false;
  for (int64_t k = 0; k < nblocks; ++k) {
    const mxlo_block_desc &b = blocks[k];
    MXLO_REQUIRE(b.kind >= MXLO_BLK_DIAG && b.kind <= MXLO_BLK_CSC, MXLO_EINVAL, "block %lld: bad kind", (long long)k);
    MXLO_REQUIRE(b.m >= 0 && b.n >= 0, MXLO_ESHAPE, "block %lld: negative size", (long long)k);
    if (b.kind == MXLO_BLK_DIAG) MXLO_REQUIRE(b.m == b.n && (b.m == 0 || b.data), MXLO_ESHAPE, "block %lld: diagonal blocks are square", (long long)k);
    if (b.kind == MXLO_BLK_DENSE) MXLO_REQUIRE((b.m == 0 || b.n == 0 || b.data) && b.ld >= (b.m > 1 ? b.m : 1), MXLO_ESHAPE, "block %lld: bad dense block", (long long)k);
    MXLO_REQUIRE(b.row_off == nrow && b.col_off == ncol, MXLO_ESHAPE, "block %lld: offsets must be cumulative", (long long)k);
    hb[k] = DevBlock{b.kind, 0, b.row_off, b.col_off, b.m, b.n, b.data, b.ld};
    int nch_n = 0, nch_t = 0;
    if (b.kind == MXLO_BLK_CSC) {
      MXLO_REQUIRE(b.data, MXLO_EINVAL, "block %lld: a sparse block needs its mxlo_csc handle in `data`", (long long)k);
      int64_t sm = 0, sn = 0;
      int sdt = 0, nlong = 0;
      csc_shape((const mxlo_csc *)b.data, &sm, &sn, &sdt, &nch_n, &nch_t, &nlong);
      // the one-launch kernel dereferences the handle's DEVICE descriptor, and mxlo_csc_refresh orders its value gather on
      // the handle's ctx stream: both only make sense when the handle was built on THIS ctx (same device, same stream order)
      MXLO_REQUIRE(csc_ctx((const mxlo_csc *)b.data) == ctx, MXLO_EINVAL,
                   "block %lld: the sparse handle belongs to another mxlo_ctx than the block-diagonal operator (build both on one ctx; "
                   "refreshes of the handle are ordered on ITS ctx stream)", (long long)k);
      MXLO_REQUIRE(nlong == 0, MXLO_EINVAL,
                   "block %lld: the sparse block has a row or column with more than %d stored entries, which needs the "
                   "two-launch apply of mxlo_csc_mul (mxlo_csc_info reports it): apply that block on its own",
                   (long long)k, kSpChunk);
      has_sparse = true;
      MXLO_REQUIRE(sm == b.m && sn == b.n, MXLO_ESHAPE, "block %lld: descriptor says %lld x %lld, the sparse handle %lld x %lld",
                   (long long)k, (long long)b.m, (long long)b.n, (long long)sm, (long long)sn);
      MXLO_REQUIRE(sdt == dtype, MXLO_EINVAL, "block %lld: element type of the sparse handle differs from the operator's", (long long)k);
      hb[k].data = csc_device_desc((const mxlo_csc *)b.data);
    }
    nrow += b.m;
    ncol += b.n;
    auto cut = [&](std::vector<Tile> &out, int64_t off, int64_t len, int64_t step) {
      if (b.kind == MXLO_BLK_CSC) {                      // one tile per chunk of the sweep (step = number of chunks)
        for (int64_t c = 0; c < step; ++c) out.push_back(Tile{hb[k], c, 0});
        return;
      }
      if (b.kind == MXLO_BLK_DENSE) {
        for (int64_t s = 0; s < len; s += step) out.push_back(Tile{hb[k], s, len - s < step ? len - s : step});
        return;
      }
      // elementwise: boundaries at global multiples of kTileE (res is normally >= 256-B aligned)
      int64_t s = 0;
      while (s < len) {
        const int64_t g = off + s;
        int64_t e = ((g / kTileE) + 1) * kTileE - off;
        if (e > len) e = len;
        out.push_back(Tile{hb[k], s, e - s});
        s = e;
      }
    };
    cut(tn, b.row_off, b.m, b.kind == MXLO_BLK_DENSE ? kTileDN : b.kind == MXLO_BLK_CSC ? nch_n : kTileE);
    cut(tt, b.col_off, b.n, b.kind == MXLO_BLK_DENSE ? kTileDT : b.kind == MXLO_BLK_CSC ? nch_t : kTileE);
  }
  mxlo_blockdiag *bd = new mxlo_blockdiag();
  bd->ctx = ctx;
  bd->dtype = dtype;
  bd->nblocks = nblocks;
  bd->nrow = nrow;
  bd->ncol = ncol;
  bd->ntiles_n = (int64_t)tn.size();
  bd->ntiles_t = (int64_t)tt.size();
  bd->lds_bytes = has_sparse ? sp_lds_bytes<double>() : 0;   // (the Float32 instantiation uses 16 of the 24 KiB)
  hipError_t e = hipSuccess;
  auto up = [&](void **dst, const void *src, size_t bytes) {
    if (e != hipSuccess || bytes == 0) return;
    e = hipMalloc(dst, bytes);
    if (e == hipSuccess) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  up((void **)&bd->d_blocks, hb.data(), hb.size() * sizeof(DevBlock));
  up((void **)&bd->d_tiles_n, tn.data(), tn.size() * sizeof(Tile));
  up((void **)&bd->d_tiles_t, tt.data(), tt.size() * sizeof(Tile));
  if (e != hipSuccess) {
    set_error("mxlo_blockdiag_create: %s", hipGetErrorString(e));
    mxlo_blockdiag_destroy(bd);
    return MXLO_ENOMEM;
  }
  *out = bd;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_blockdiag_destroy(mxlo_blockdiag *bd) {
  if (!bd) return MXLO_OK;
  MXLO_DEVICE_GUARD(bd->ctx);
  (void)hipStreamSynchronize(bd->ctx->stream);
  if (bd->d_blocks) (void)hipFree(bd->d_blocks);
  if (bd->d_tiles_n) (void)hipFree(bd->d_tiles_n);
  if (bd->d_tiles_t) (void)hipFree(bd->d_tiles_t);
  delete bd;
  return MXLO_OK;
}

template <typename T>
static int32_t blockdiag_mul_t(mxlo_blockdiag *bd, T *res, const T *v, double alpha, double beta,
                               int32_t op_mode, int32_t flags) {
  mxlo_ctx *ctx = bd->ctx;
  const bool trans = op_mode != MXLO_OP_N;
  const int64_t nt = trans ? bd->ntiles_t : bd->ntiles_n;
  const Tile *tiles = trans ? bd->d_tiles_t : bd->d_tiles_n;
  if (nt == 0) return MXLO_OK;
  MXLO_REQUIRE(nt < (1LL << 31), MXLO_ESHAPE, "too many tiles");
  return dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    const bool ntm = (int64_t)sizeof(T) * (bd->nrow + 2 * bd->ncol) >= ctx->tune.nt_min_bytes;
#define BD_GO(TR_, NT_)                                                                               \
  do {                                                                                                \
  if (bd->lds_bytes)                                                                                  \
    hipLaunchKernelGGL((blockdiag_kernel<T, CA, CB, B0, TR_, NT_, true>), dim3((unsigned)nt), dim3(kBlock),           \
                       bd->lds_bytes, ctx->stream, res, v, bd->d_blocks, tiles, (CA)alpha, (CB)beta);  \
  else                                                                                                \
    hipLaunchKernelGGL((blockdiag_kernel<T, CA, CB, B0, TR_, NT_, false>), dim3((unsigned)nt), dim3(kBlock), 0,       \
                       ctx->stream, res, v, bd->d_blocks, tiles, (CA)alpha, (CB)beta);                 \
  } while (0)
    if (trans && ntm) BD_GO(true, true);
    else if (trans) BD_GO(true, false);
    else if (ntm) BD_GO(false, true);
    else BD_GO(false, false);
#undef BD_GO
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
}

MXLO_API int32_t mxlo_blockdiag_mul(mxlo_blockdiag *bd, void *res, const void *v, double alpha,
                                    double beta, int32_t op_mode, int32_t flags) {
  MXLO_REQUIRE(bd, MXLO_EINVAL, "mxlo_blockdiag_mul: handle is NULL");
  MXLO_DEVICE_GUARD(bd->ctx);
  MXLO_REQUIRE(op_mode >= MXLO_OP_N && op_mode <= MXLO_OP_C, MXLO_EINVAL, "bad op_mode");
  const int64_t nres = op_mode == MXLO_OP_N ? bd->nrow : bd->ncol;
  if (nres == 0) return MXLO_OK;
  MXLO_REQUIRE(res && v, MXLO_EINVAL, "mxlo_blockdiag_mul: NULL operand");
  eff_scalars(bd->dtype == MXLO_F64 ? 8 : 4, flags, alpha, beta);
  if (bd->dtype == MXLO_F64) return blockdiag_mul_t<double>(bd, (double *)res, (const double *)v, alpha, beta, op_mode, flags);
  return blockdiag_mul_t<float>(bd, (float *)res, (const float *)v, alpha, beta, op_mode, flags);
}
