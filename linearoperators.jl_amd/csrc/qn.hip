// qn.hip — quasi-Newton operators with their state resident in HBM:
//   InverseLBFGSOperator (src/lbfgs.jl:112-158), LBFGSOperator (src/lbfgs.jl:168-206),
//   LSR1Operator (src/lsr1.jl:86-111), push! (src/lbfgs.jl:210-367, src/lsr1.jl:119-184),
//   diag! , reset!, solve_shifted_system! (src/utilities.jl:207-248).
//
// Data layout: s, y (and a, b) are n x mem column-major PANELS (column k = slot k, leading
// dimension ld = n rounded up to a 16-byte multiple) so every column streams coalesced.
//
// Every apply is the same three-stage shape (the "skinny panel" form):
//   1. panel_dots   : ONE pass over x and the active panel columns -> up to 2m dots (f64),
//                     fixed-order finalize, [all-reduce hook for row-sharded vectors]
//   2. coef kernel  : one tiny launch turning dots (+ Gram matrices kept up to date by push!)
//                     into per-column coefficients, all in device memory (no host sync)
//   3. panel_combine: ONE pass reading x and the columns, writing res, applying the columns in the
//                     reference's elementwise statement order (no FMA fusion).
// HBM traffic per apply: (4m+3)*8 B/elt for L-BFGS (vs (48+80m) for the reference's statement
// sequence). The reference-ordered inverse two-loop (2m chained dot/axpy passes) is kept as
// MXLO_INV_REFORDER for validation.
#include <chrono>
#include <thread>
#include <cmath>
#include <vector>

#include "common.h"

#include <cstdlib>
#include "stream_kernels.h"

using namespace mxlo;

namespace {

constexpr int kMaxMem = 64;          // slots of the inverse operator (its coefficient kernel owns one lane per slot)
constexpr int kMaxMemFwd = 32;       // forward L-BFGS / L-SR1: one lane per BASIS vector, 2*mem <= 64
constexpr int kMaxCols = 2 * kMaxMem;   // panel columns per combine
constexpr int kX0Slot = 64;          // CM_AXPYS (forward shifted solve, <= 64 columns): where the factor of x travels

// ---- device scalar region layout (doubles) ------------------------------------------
struct DscLayout {
  // fixed-size head for mem <= 64 (the by-value fast path); every region grows with mem beyond that
  int64_t dots = 0;    // [nd] panel_dots outputs of the current apply
  int64_t coef = 128;  // [nd] coefficients consumed by the combine pass (+1: the factor of x of the shifted solve)
  int64_t misc = 256;  // [16 + ns] push! scalars, then per-slot norms from misc + 16
  int64_t as_ = 320;   // [ns] L-SR1 a_k' s_k
  int64_t alpha = 384; // [ns] inverse two-loop alpha_k (data.α)
  int64_t SY = 448;    // [mem*mem] column j = S' y_j  (written when slot j is pushed)
  int64_t YS = 0, YY = 0, G = 0, g = 0, cx = 0, SS = 0, YSf = 0, Cm = 0, gtmp = 0, Wm = 0, total = 0;
  int64_t bigM = 0, bigT = 0, bigP = 0;   // [4 mem^2] each: shifted-solve work matrices of the big-memory path
  explicit DscLayout(int64_t mem, bool big = false) {
    const int64_t nd = big ? 2 * mem + 64 : 128;     // dots / coef
    const int64_t ns = big ? mem + 64 : 64;          // per-slot scalar arrays
    dots = 0;
    coef = nd;
    misc = 2 * nd;
    as_ = misc + (big ? 16 + ns : 64);
    alpha = as_ + ns;
    SY = alpha + ns;
    YS = SY + mem * mem;
    YY = YS + mem * mem;
    G = YY + mem * mem;
    g = G + 4 * mem * mem;
    cx = g + 2 * mem;
    SS = cx + 2 * mem + 64;   // [mem*mem] s_j's_k            (forward push!, Gram form)
    YSf = SS + mem * mem;     // [mem*mem] y_j's_k (row j, col k)
    Cm = YSf + mem * mem;     // [mem * 2mem] coefficients of a_k on [s_1..s_r, b_1..b_r]
    gtmp = Cm + 2 * mem * mem;  // [4*mem] scratch dots of one push
    Wm = gtmp + 4 * mem + 64;   // [2mem x 2mem] shifted solve: coefficients of u_t on the basis [s.. b..]
    total = Wm + 4 * mem * mem + 64;
    if (big) {
      bigM = total;
      bigT = bigM + 4 * mem * mem;
      bigP = bigT + 4 * mem * mem;
      total = bigP + 4 * mem * mem + 64;
    }
  }
};

}  // namespace

struct mxlo_qn {
  mxlo_ctx *ctx = nullptr;
  int kind = 0, dtype = 0;
  int64_t n = 0, mem = 0, ld = 0;
  bool scaling = true, damped = false;
  double sigma2 = 0.99, sigma3 = 10.0;
  int mode = MXLO_INV_TWOPASS;
  int push_mode = MXLO_PUSH_GRAM;
  // host mirrors of the small scalars that drive control flow
  int64_t insert0 = 0;  // 0-based next write slot
  double scaling_factor = 1.0;
  std::vector<double> ys;     // [mem]
  std::vector<int64_t> age;   // push counter per slot (Gram freshness)
  int64_t pushes = 0;
  bool G_valid = false;
  bool normA_valid = true;   // L-SR1 ||a_k||^2 (opnorm bound) computed lazily after a Gram-form push
  bool A_valid = true;       // forward L-BFGS: the a_k panel is materialised (false after a compact push!)
  bool gram_ok = true;       // forward / L-SR1: S'S, Y'S, Y'Y and Cm describe the current pairs (false after a
                             // reference-ordered push!, which does not maintain them)
  // device
  void *S = nullptr, *Y = nullptr, *A = nullptr, *B = nullptr;  // panels
  void *tmp = nullptr, *tmp2 = nullptr;                         // n-vectors (Ax / tmp)
  double *dsc = nullptr;
  DscLayout lay{1};
  // memories beyond the single-wave coefficient kernels (qn_big.h): slot order / ys / age live in device memory
  bool big = false;
  void *meta = nullptr;
  std::vector<char> meta_host;
  int64_t generation = 0;   // bumped by every state change: a captured hipGraph of an apply is stale afterwards
  // kPinnedScalars doubles (+ one sequence word) of pinned, device-mapped host memory: where push! reads its decision
  // scalars back to. pinned_dev is the device's address of the same bytes (NULL: not mapped — plain copies only).
  double *pinned = nullptr, *pinned_dev = nullptr;
  unsigned long long post_seq = 0;   // sequence number of the last posted read-back
};
constexpr int kPinnedScalars = 64;
constexpr int kPostPairs = kPinnedScalars + 8;        // 8 (value, sequence number) pairs posted by finalize launches
constexpr int kPinnedTotal = kPostPairs + 16;         // layout: [0,64) doubles | [64] their sequence number | [72,88) pairs

namespace {

template <typename T>
inline T *col(void *panel, int64_t ld, int64_t k) {
  return (T *)panel + k * ld;
}

// ---- panel_combine ---------------------------------------------------------------------
enum CombineMode {
  CM_FWD = 0,       // q = x (/g); pairs (b,a): q = q + ((cb*b) - (ca*a)); res = al*q (+ be*res)
  CM_INV = 1,       // q = x; cols y: q = q - c*y; q = q*g; cols s: q = q + c*s; res = al*q (+ be*res)
  CM_LSR1 = 2,      // q = (al*x)/g (+ be*res); cols a: q = T(q + c*a)          (c carries al)
  CM_AFWD = 3,      // q = x/g; pairs (b,a): q = q + c1*b; q = q - c2*a; res = q   (push!: a_k rebuild)
  CM_ASR1 = 4,      // q = x2 - x/g; cols a: q = q - c*a; res = q                  (L-SR1 a_k rebuild)
  CM_AXPYS = 5,     // q = c0*x; cols u: q = q + c*u; res = q                      (shifted solve)
  CM_DIAG_FWD = 6,  // q = 1 (/g); pairs (b,a): q = q + ((b*b) - (a*a)); res = q
  CM_DIAG_SR1 = 7,  // q = 1 (/g); cols a: q = q + ((a*a)/as); res = q
  CM_CFWD = 8,      // q = x (/g); cols u: q = q + c*u; res = al*q (+ be*res)   (compact forward L-BFGS)
  CM_LSR1R = 9,     // CM_LSR1 with nothing stored: r = q, per-workgroup partial sums of r·x, r·r and |res - x/sf|²
                    // (L-SR1 push!: r = y - B s with res = y, x = s, and the decision scalars of lsr1.jl:126-141)
};

template <typename T>
struct CombineArgs {
  const T *cols[kMaxCols];
  int ncol;        // total columns (pairs count 2)
  int nfirst;      // CM_INV: number of y columns (the rest are s columns)
  int use_gamma;   // divide / multiply by gamma
  double gamma;
  double alpha, beta;
  const double *coef;  // device coefficients, one per column (CM_DIAG_SR1: as_k per column)
  double shift = 0.0;  // CM_FWD/INV/LSR1: res += T(shift)*x after the epilogue (fused ShiftedOperator axpy!)
  int reverse = 0;     // walk the vectors back to front (set by the applies that follow a dots pass over the same panel)
  // CM_LSR1R only: valid elements of x / res (the last vector may be partial; panel columns are zero padded), the
  // device scalars sf = T(*sfnum) / T(*sfden) is formed from, and the [3][kMaxRedBlocks] partial sums
  int64_t n_valid = 0;
  const double *sfnum = nullptr, *sfden = nullptr;
  double *partials = nullptr;
};

template <typename T, typename CA, typename CB, int MODE, bool BETA0, int VEC, bool NT>
__global__ void __launch_bounds__(kBlock)
combine_kernel(T *__restrict__ res, const T *__restrict__ x, const T *__restrict__ x2,
               CombineArgs<T> A, int64_t nvec) {
  using V = typename VecOf<T, VEC>::type;
  const T g = (T)A.gamma;
  // per-column coefficients: device memory -> LDS once per workgroup (broadcast ds_reads in the column
  // loop instead of one extra VMEM instruction per column competing with the streaming loads)
  __shared__ double scoef[kMaxCols + 1];
  if (threadIdx.x <= kMaxCols)
    scoef[threadIdx.x] = (A.coef && (threadIdx.x < (unsigned)A.ncol || (MODE == CM_AXPYS && threadIdx.x == kX0Slot)))
                             ? A.coef[threadIdx.x] : 0.0;
  __syncthreads();
  const CA al = (CA)A.alpha;
  const CB be = (CB)A.beta;
  const int ncol = A.ncol;
  double racc[3] = {0.0, 0.0, 0.0};   // CM_LSR1R
  T sfac = T(1);
  if constexpr (MODE == CM_LSR1R) sfac = (T)(*A.sfnum) / (T)(*A.sfden);   // sf = ys/yy (lsr1.jl:139), as YmSOverDevOp
  // CM_LSR1R: x and res are caller vectors of n_valid elements — the last vector is read element by element
  auto ldt = [&](const T *p, int64_t i) -> V {
    if ((i + 1) * VEC <= A.n_valid) return ldg<NT>(reinterpret_cast<const V *>(p + i * VEC));
    V v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) vset<T, VEC>(v, e, i * VEC + e < A.n_valid ? p[i * VEC + e] : T(0));
    return v;
  };
  // A.reverse: workgroup 0 takes the END of the vectors — what the dots pass that ran just before touched last and the
  // Infinity Cache still holds (same elementwise arithmetic; only the order the workgroups walk the vectors in changes)
  const int64_t nblk = (nvec + kBlock - 1) / kBlock;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t i = (A.reverse ? nblk - 1 - blk : blk) * kBlock + threadIdx.x;
    if (i >= nvec) continue;
    T q[VEC];
    V xv, rvk;
    // ---- prologue
    if constexpr (MODE == CM_DIAG_FWD || MODE == CM_DIAG_SR1) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) q[e] = A.use_gamma ? (T)1 / g : (T)1;
    } else {
      if constexpr (MODE == CM_LSR1R) xv = ldt(x, i);
      else xv = ldg<NT>(reinterpret_cast<const V *>(x + i * VEC));
      V x2v, rv;
      if constexpr (MODE == CM_ASR1) x2v = ldg<NT>(reinterpret_cast<const V *>(x2 + i * VEC));
      if constexpr (MODE == CM_LSR1 && !BETA0) rv = ldg<NT>(reinterpret_cast<const V *>(res + i * VEC));
      if constexpr (MODE == CM_LSR1R) rv = rvk = ldt(res, i);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const T xe = vget<T, VEC>(xv, e);
        if constexpr (MODE == CM_FWD || MODE == CM_AFWD || MODE == CM_CFWD) q[e] = A.use_gamma ? xe / g : xe;
        else if constexpr (MODE == CM_INV) q[e] = xe;
        else if constexpr (MODE == CM_LSR1 || MODE == CM_LSR1R) {
          // (α*x)/γ : γ divided unconditionally (src/lsr1.jl:93)
          q[e] = fin_ab<T, CA, CB, BETA0>((al * (CA)xe) / (CA)g, be, BETA0 ? T(0) : vget<T, VEC>(rv, e));
        } else if constexpr (MODE == CM_ASR1) q[e] = vget<T, VEC>(x2v, e) - (xe / g);
        else if constexpr (MODE == CM_AXPYS) q[e] = (T)scoef[kX0Slot] * xe;  // c0 stored past the columns
      }
    }
    // ---- columns. Straight-line batches of UB columns: the UB column pointers (one wide scalar load), the UB
    // coefficients (LDS) and the UB 16-byte loads are all issued before the first use — the previous form guarded
    // every column with a scalar branch and waited on a scalar pointer load before EACH vector load. Batches of 8,
    // then 4 / 2 / 1 for the remainder, so no load is ever issued for a column that does not exist.
    auto batch = [&]<int UB>(int c0) {
      const T *pc[UB];
      double cf[UB];
      V cv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) pc[u] = A.cols[c0 + u];
#pragma unroll
      for (int u = 0; u < UB; ++u) cf[u] = scoef[c0 + u];
#pragma unroll
      for (int u = 0; u < UB; ++u) cv[u] = ldg<NT>(reinterpret_cast<const V *>(pc[u] + i * VEC));
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int c = c0 + u;
        if constexpr (MODE == CM_FWD || MODE == CM_AFWD || MODE == CM_DIAG_FWD) {
          if constexpr (UB >= 2) {
            if (u & 1) continue;  // pairs (b, a) are handled on the even member; batches of these modes are even
            const T cb = (T)cf[u], ca = (T)cf[u + 1];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const T b = vget<T, VEC>(cv[u], e), a = vget<T, VEC>(cv[u + 1], e);
              if constexpr (MODE == CM_FWD) q[e] = q[e] + ((cb * b) - (ca * a));       // lbfgs.jl:194
              else if constexpr (MODE == CM_AFWD) {
                q[e] = q[e] + (cb * b);                                                 // lbfgs.jl:244
                q[e] = q[e] - (ca * a);                                                 // lbfgs.jl:245
              } else q[e] = q[e] + ((b * b) - (a * a));                                 // lbfgs.jl:391
            }
          }
        } else if constexpr (MODE == CM_INV) {
          if (c == A.nfirst && A.use_gamma) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) q[e] = q[e] * g;                            // lbfgs.jl:139
          }
          const T cc = (T)cf[u];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const T ce = vget<T, VEC>(cv[u], e);
            if (c < A.nfirst) q[e] = q[e] - (cc * ce);                                // lbfgs.jl:135
            else q[e] = q[e] + (cc * ce);                                             // lbfgs.jl:146
          }
        } else if constexpr (MODE == CM_LSR1 || MODE == CM_LSR1R) {
          const CA cc = (CA)cf[u];  // (α*dot)/as evaluated in α's type by the coef kernel
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            q[e] = (T)((CA)q[e] + (cc * (CA)vget<T, VEC>(cv[u], e)));                 // lsr1.jl:103
        } else if constexpr (MODE == CM_ASR1) {
          const T cc = (T)cf[u];
#pragma unroll
          for (int e = 0; e < VEC; ++e) q[e] = q[e] - (cc * vget<T, VEC>(cv[u], e));  // lsr1.jl:174
        } else if constexpr (MODE == CM_AXPYS || MODE == CM_CFWD) {
          const T cc = (T)cf[u];
#pragma unroll
          for (int e = 0; e < VEC; ++e) q[e] = q[e] + (cc * vget<T, VEC>(cv[u], e));
        } else if constexpr (MODE == CM_DIAG_SR1) {
          const T as = (T)cf[u];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const T a = vget<T, VEC>(cv[u], e);
            q[e] = q[e] + ((a * a) / as);                                             // lsr1.jl:206
          }
        }
      }
    };
    int c0 = 0;
    for (; c0 + 8 <= ncol; c0 += 8) batch.template operator()<8>(c0);
    if (c0 + 4 <= ncol) { batch.template operator()<4>(c0); c0 += 4; }
    if (c0 + 2 <= ncol) { batch.template operator()<2>(c0); c0 += 2; }
    if (c0 < ncol) batch.template operator()<1>(c0);        // odd counts exist only in the single-column modes
    if constexpr (MODE == CM_INV) {
      if (A.nfirst == ncol && A.use_gamma) {  // no s column followed (cannot happen: nfirst*2 == ncol)
#pragma unroll
        for (int e = 0; e < VEC; ++e) q[e] = q[e] * g;
      }
    }
    // ---- epilogue
    if constexpr (MODE == CM_LSR1R) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if (i * VEC + e < A.n_valid) {
          const T xe = vget<T, VEC>(xv, e), t = vget<T, VEC>(rvk, e) - (xe / sfac);      // y - s/sf  (lsr1.jl:140)
          racc[0] = fma((double)q[e], (double)xe, racc[0]);                              // dot(ymBs, s)   (:126)
          racc[1] = fma((double)q[e], (double)q[e], racc[1]);                            // |ymBs|^2       (:127)
          racc[2] = fma((double)t, (double)t, racc[2]);
        }
      }
      continue;
    }
    V out;
    if constexpr (MODE == CM_FWD || MODE == CM_INV || MODE == CM_CFWD) {
      V rv;
      if constexpr (!BETA0) rv = ldg<NT>(reinterpret_cast<const V *>(res + i * VEC));
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        vset<T, VEC>(out, e, fin_ab<T, CA, CB, BETA0>(al * (CA)q[e], be, BETA0 ? T(0) : vget<T, VEC>(rv, e)));  // lbfgs.jl:150,198
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) vset<T, VEC>(out, e, q[e]);
    }
    if constexpr (MODE == CM_FWD || MODE == CM_INV || MODE == CM_LSR1 || MODE == CM_CFWD) {
      if (A.shift != 0.0) {  // axpy!(α σ, x, y) of shifted_prod! (src/shifted_operators.jl:21-23), in T like BLAS axpy
        const T sh = (T)A.shift;
#pragma unroll
        for (int e = 0; e < VEC; ++e) vset<T, VEC>(out, e, vget<T, VEC>(out, e) + (sh * vget<T, VEC>(xv, e)));
      }
    }
    stg<NT>(reinterpret_cast<V *>(res + i * VEC), out);
  }
  if constexpr (MODE == CM_LSR1R) {   // fixed order: DPP tree per wave, the four waves pairwise, one slot per workgroup
    __shared__ double sred[kBlock / 64][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double w = wave_allsum(racc[c]);
      if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6][c] = w;
    }
    __syncthreads();
    if (threadIdx.x < 3)
      A.partials[(int64_t)threadIdx.x * kMaxRedBlocks + blockIdx.x] =
          (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
  }
}

template <typename T, int MODE>
int32_t launch_combine_part(mxlo_ctx *ctx, T *res, const T *x, const T *x2, const CombineArgs<T> &A,
                            int64_t n, int32_t flags, bool vec) {
  if (n <= 0) return MXLO_OK;
  constexpr int VECF = Vec16<T>::N;
  const bool b0 = A.beta == 0;
  auto go = [&]<typename CA, typename CB, bool B0, int VEC>() -> int32_t {
    const int64_t nvec = n / VEC;
    const int grid = grid_for(ctx, nvec, kBlock, ctx->tune.combine_blocks_per_cu);
    const bool nt = (int64_t)sizeof(T) * n * (A.ncol + 2) >= ctx->tune.nt_min_bytes;
    if (nt)
      hipLaunchKernelGGL((combine_kernel<T, CA, CB, MODE, B0, VEC, true>), dim3(grid), dim3(kBlock), 0,
                         ctx->stream, res, x, x2, A, nvec);
    else
      hipLaunchKernelGGL((combine_kernel<T, CA, CB, MODE, B0, VEC, false>), dim3(grid), dim3(kBlock), 0,
                         ctx->stream, res, x, x2, A, nvec);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  };
  constexpr bool uses_ab = (MODE == CM_FWD || MODE == CM_INV || MODE == CM_LSR1 || MODE == CM_CFWD);
  if constexpr (!uses_ab) {
    return vec ? go.template operator()<T, T, true, VECF>() : go.template operator()<T, T, true, 1>();
  } else {
    return dispatch_ab<T>(b0 ? 0.0 : 1.0, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
      return vec ? go.template operator()<CA, CB, B0, VECF>() : go.template operator()<CA, CB, B0, 1>();
    });
  }
}

// Panel columns are 16-byte aligned by construction (ld is a multiple of the vector width); the
// vector path additionally needs res / x / x2 16-byte aligned. A non-multiple-of-VEC tail is a
// second, scalar launch on offset pointers.
template <typename T, int MODE>
int32_t launch_combine(mxlo_ctx *ctx, T *res, const T *x, const T *x2, CombineArgs<T> &A,
                       int64_t n, int32_t flags) {
  if (n <= 0) return MXLO_OK;
  constexpr int VECF = Vec16<T>::N;
  bool vec = (((uintptr_t)res & 15u) == 0) && (!x || ((uintptr_t)x & 15u) == 0) &&
             (!x2 || ((uintptr_t)x2 & 15u) == 0) && n >= VECF;
  for (int c = 0; c < A.ncol && vec; ++c) vec = (((uintptr_t)A.cols[c]) & 15u) == 0;
  if (!vec) return launch_combine_part<T, MODE>(ctx, res, x, x2, A, n, flags, false);
  const int64_t nbody = (n / VECF) * VECF;
  MXLO_TRY((launch_combine_part<T, MODE>(ctx, res, x, x2, A, nbody, flags, true)));
  if (nbody < n) {
    CombineArgs<T> At = A;
    for (int c = 0; c < A.ncol; ++c) At.cols[c] = A.cols[c] + nbody;
    return launch_combine_part<T, MODE>(ctx, res + nbody, x ? x + nbody : x, x2 ? x2 + nbody : x2, At,
                                        n - nbody, flags, false);
  }
  return MXLO_OK;
}

// ---- coefficient kernels (one wave, lane 0 does the O(m^2) scalar work) ------------------
struct OrdArgs {
  int na;
  int ord[kMaxMem];      // active slots in the order the dots / columns were issued
  double ys[kMaxMem];    // ys[slot]
  long long age[kMaxMem];
  int mem;
  double gamma;
  int use_gamma;
  int is_f32;            // round scalars to float like the reference's T arithmetic
};

// coef[i] = as[ord[i]] (L-SR1 diag!: the active slots' 1/(a_k's_k) in apply order)
__global__ void gather_as_kernel(const double *__restrict__ as_, double *__restrict__ coef, OrdArgs O) {
  for (int i = threadIdx.x; i < O.na; i += blockDim.x) coef[i] = as_[O.ord[i]];
}

__device__ __forceinline__ double rnd(double v, int is_f32) { return is_f32 ? (double)(float)v : v; }

// Inverse two-loop in coefficient space (SURVEY §8a equivalence (i)):
//   dots[i] = s_i'x, dots[na+i] = y_i'x for i = 0..na-1 in NEWEST->OLDEST order (ord[]).
//   coef[i]      = alpha_i  (column y_ord[i], newest->oldest)
//   coef[na + j] = beta for column s in OLDEST->NEWEST order (j = 0 is ord[na-1]).
// (body: ONE full wave, `lane` = lane id; sy(i, j) = s'y and yy(i, j) = y'y of the pairs at ORD POSITIONS i, j — the
// standalone kernel reads them from the Gram matrices in device memory, the single-launch apply from its LDS copies)
template <typename SYF, typename YYF>
__device__ __forceinline__ void inv_coef_generic(int lane, const double *dots, double *coef, double *alpha_out,
                                                 const OrdArgs &O, SYF &&syp, YYF &&yyp) {
  // one wave: lane j holds a_j / b_j; every inner sum of the recurrences is a wave reduction
  const int na = O.na;
  const int mypos = lane < na ? lane : 0;
  auto wsum = [](double v) { return wave_allsum(v); };   // DPP tree (common.h), total in every lane
  const int myslot = O.ord[mypos];
  // ys of ord position i is fetched from lane i's register (one kernel-argument lookup per lane, up front) instead of
  // O.ys[O.ord[i]] inside the loops: two dependent scalar loads per step of a chain that is latency from end to end
  const double ys_mine = O.ys[myslot];
  auto ys_at = [&](int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ys_mine), i),
                            __builtin_amdgcn_readlane(__double2loint(ys_mine), i));
#else
    return ys_mine;
#endif
  };
  double a = 0.0, b = 0.0;  // lane's own alpha / beta
  for (int i = 0; i < na; ++i) {
    const double part = (lane < i) ? a * syp(i, mypos) : 0.0;
    const double sq = dots[i] - wsum(part);
    const double ai = rnd(sq / ys_at(i), O.is_f32);
    if (lane == i) a = ai;
  }
  if (lane < na && alpha_out) alpha_out[myslot] = a;
  const double g = O.use_gamma ? O.gamma : 1.0;
  for (int i = na - 1; i >= 0; --i) {
    const double p1 = (lane < na) ? a * yyp(i, mypos) : 0.0;              // sum_j alpha_j y_k'y_j
    const double p2 = (lane < na && lane > i) ? b * syp(mypos, i) : 0.0;  // sum_{older j} beta_j s_j'y_k
    const double yq = dots[na + i] - wsum(p1);
    const double yr = g * yq + wsum(p2);
    const double ai = __shfl(a, i, 64);
    const double bi = rnd(ai - yr / ys_at(i), O.is_f32);
    if (lane == i) b = bi;
  }
  if (lane < na) {
    coef[lane] = a;                   // y columns, newest -> oldest
    coef[na + (na - 1 - lane)] = b;   // s columns, oldest -> newest
  }
}
// Standalone coefficient kernel of the four-launch apply. The Gram entries the recurrences read — chosen per pair by the
// pairs' ages — and the dots are first staged in LDS by ord position with all loads in flight (one workgroup, 256
// threads; wave 0 then runs the recurrences): read in place they cost one dependent global round trip per step of the
// two loops (6.3 us at na = 10, profiles/r04_bench_kernel_stats.csv). Same values, same arithmetic.
__global__ void __launch_bounds__(kBlock)
inv_coef_kernel(const double *__restrict__ dots, double *__restrict__ coef,
                const double *__restrict__ SY, const double *__restrict__ YS,
                const double *__restrict__ YY, double *__restrict__ alpha_out, OrdArgs O) {
  extern __shared__ double inv_lds[];            // [na*na] s_i'y_j | [na*na] y_i'y_j | [2 na] dots
  const int tid = threadIdx.x, na = O.na, mem = O.mem;
  double *sg1 = inv_lds, *sg2 = inv_lds + na * na, *sdots = inv_lds + 2 * na * na;
  for (int p = tid; p < na * na; p += kBlock) {
    const int i = O.ord[p / na], j = O.ord[p % na];                  // slots
    sg1[p] = O.age[j] >= O.age[i] ? SY[i + (int64_t)j * mem] : YS[j + (int64_t)i * mem];   // s_i'y_j
    sg2[p] = O.age[j] >= O.age[i] ? YY[i + (int64_t)j * mem] : YY[j + (int64_t)i * mem];   // y_i'y_j
  }
  if (tid < 2 * na) sdots[tid] = dots[tid];
  __syncthreads();
  if (tid < kWave)
    inv_coef_generic(tid, sdots, coef, alpha_out, O, [&](int i, int j) { return sg1[i * na + j]; },
                     [&](int i, int j) { return sg2[i * na + j]; });
}

// L-SR1: coef[i] = (alpha*dot_i)/as_k evaluated in CT (src/lsr1.jl:101)
__device__ __forceinline__ void lsr1_coef_body(int i, const double *dots, double *__restrict__ coef,
                                               const double *__restrict__ as_, const OrdArgs &O, double alpha, int ct_f32) {
  if (i >= O.na) return;
  const int k = O.ord[i];
  const double d = rnd(dots[i], O.is_f32), as = rnd(as_[k], O.is_f32);
  if (ct_f32) coef[i] = (double)(((float)alpha * (float)d) / (float)as);
  else coef[i] = (alpha * d) / as;
}
__global__ void lsr1_coef_kernel(const double *__restrict__ dots, double *__restrict__ coef,
                                 const double *__restrict__ as_, OrdArgs O, double alpha, int ct_f32) {
  lsr1_coef_body((int)threadIdx.x, dots, coef, as_, O, alpha, ct_f32);
}

// coef[i] = T(dots[i] / as[ord[i]])   (L-SR1 push!: as = dot(a_l, s_k)/as_l, src/lsr1.jl:173)
__global__ void div_as_kernel(const double *__restrict__ dots, double *__restrict__ coef,
                              const double *__restrict__ as_, OrdArgs O) {
  const int i = threadIdx.x;
  if (i >= O.na) return;
  coef[i] = rnd(rnd(dots[i], O.is_f32) / rnd(as_[O.ord[i]], O.is_f32), O.is_f32);
}

// coef[i] = T(dots[i])
__global__ void copy_coef_kernel(const double *__restrict__ dots, double *__restrict__ coef, int n,
                                 int is_f32) {
  const int i = threadIdx.x;
  if (i < n) coef[i] = rnd(dots[i], is_f32);
}

// v[i] = v[i] / sqrt(dot)  |  v[i] = y[i] / sqrt(ys)   (src/lbfgs.jl:232,248)
template <typename T>
struct DivSqrtDevOp {
  const double *dot;
  T s;
  __device__ void init() { s = sizeof(T) == 8 ? (T)sqrt(*dot) : (T)sqrtf((float)*dot); }
  __device__ T operator()(T, T, T r) const { return r / s; }
};
template <typename T>
struct DivConstOp {
  T s;
  __device__ void init() {}
  __device__ T operator()(T y, T, T) const { return y / s; }
};
// y = th*y + (1-th)*Bs    (Powell damping, src/lbfgs.jl:316,351)
template <typename T>
struct DampOp {
  T th, omth;
  __device__ void init() {}
  __device__ T operator()(T y, T bs, T) const { return (th * y) + (omth * bs); }
};
// Bs = (-alpha)*g   (src/lbfgs.jl:341)
template <typename T>
struct NegScaleOp {
  T na;
  __device__ void init() {}
  __device__ T operator()(T g, T, T) const { return na * g; }
};
// tmp = y - s/sf   (src/lsr1.jl:140)
// y - s/sf, sf = ys / yy formed ON THE DEVICE from the freshly reduced scalars (misc[0] = y's, misc[3] = y'y), so
// that the L-SR1 push! needs one read-back instead of two
template <typename T>
struct YmSOverDevOp {
  const double *misc;
  T sf;
  __device__ void init() { sf = (T)misc[0] / (T)misc[3]; }
  __device__ T operator()(T y, T s, T) const { return y - (s / sf); }
};
// reference-ordered two-loop steps: q = q - a*y with a = dot/ys (stored), q = q + (a_k - dot/ys)*s
template <typename T>
struct InvStep1Op {
  const double *dot;
  double *alpha_slot;
  T ys, c;
  __device__ void init() {
    c = (T)(*dot) / ys;
    if (blockIdx.x == 0 && threadIdx.x == 0) *alpha_slot = (double)c;
  }
  __device__ T operator()(T y, T, T q) const { return q - (c * y); }
};
template <typename T>
struct InvStep2Op {
  const double *dot;
  const double *alpha_slot;
  T ys, c;
  __device__ void init() { c = (T)(*alpha_slot) - ((T)(*dot) / ys); }
  __device__ T operator()(T s, T, T q) const { return q + (c * s); }
};
template <typename T>
struct CopyOp {
  __device__ void init() {}
  __device__ T operator()(T x, T, T) const { return x; }
};

template <typename T>
inline T eps_of() {
  return sizeof(T) == 8 ? (T)2.220446049250313e-16 : (T)1.1920929e-07f;
}

inline void fill_ord(const mxlo_qn *h, OrdArgs &O, bool newest_first) {
  O.na = 0;
  O.mem = (int)h->mem;
  O.gamma = h->scaling_factor;
  O.use_gamma = h->scaling ? 1 : 0;
  O.is_f32 = h->dtype == MXLO_F32;
  for (int k = 0; k < h->mem; ++k) {
    O.ys[k] = h->ys[k];
    O.age[k] = h->age[k];
  }
  for (int64_t i = 0; i < h->mem; ++i) {
    // oldest->newest: slot (insert0 + i) % mem ; newest->oldest: slot (insert0 - 1 - i) mod mem
    int64_t k = newest_first ? ((h->insert0 - 1 - i) % h->mem + h->mem) % h->mem
                             : (h->insert0 + i) % h->mem;
    if (h->ys[k] != 0) O.ord[O.na++] = (int)k;
  }
}

// active slots as a host vector (any mem): oldest->newest or newest->oldest
inline std::vector<int> ord_host(const mxlo_qn *h, bool newest_first) {
  std::vector<int> o;
  for (int64_t i = 0; i < h->mem; ++i) {
    const int64_t k = newest_first ? ((h->insert0 - 1 - i) % h->mem + h->mem) % h->mem : (h->insert0 + i) % h->mem;
    if (h->ys[k] != 0) o.push_back((int)k);
  }
  return o;
}

// big-memory path (qn_big.h, included at the end of this namespace)
int32_t sync_meta(mxlo_qn *h);
template <typename T>
int32_t inv_mul_big(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags, double shift);
template <typename T>
int32_t fwd_mul_big(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags, double shift);
template <typename T>
int32_t lsr1_mul_big(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags, double shift);
template <typename T>
int32_t lbfgs_push_common_big(mxlo_qn *h, const T *s, const T *y, double ys, double yy);
template <typename T>
int32_t lsr1_rebuild_big(mxlo_qn *h, int64_t ins);

// Compact forward L-BFGS apply. With a_k = [S B]·c_k (c_k = row k of Cm, built by push!) and d = [S B]'x:
//   B x = x/γ + Σ_k b_k (b_k'x) − a_k (a_k'x) = x/γ + [S B]·w,   w = −Cmᵀ (Cm d)  (+ d on the b half).
// One wave; lane j owns w_j (2r <= 64).
__device__ __forceinline__ void cfwd_coef_body(int lane, const double *dots, const double *__restrict__ Cm,
                                               double *__restrict__ coef, int r) {
  const int w2 = 2 * r;
  const double d = lane < w2 ? dots[lane] : 0.0;
  double w = (lane >= r && lane < w2) ? d : 0.0;                 // + b_j (b_j'x)
  for (int k = 0; k < r; ++k) {
    const double ck = lane < w2 ? Cm[(int64_t)k * w2 + lane] : 0.0;
    const double t = wave_allsum(ck * d);                        // a_k'x = c_k'd
    w -= ck * t;                                                 // − a_k (a_k'x)
  }
  if (lane < w2) coef[lane] = w;
}
__global__ void __launch_bounds__(64)
cfwd_coef_kernel(const double *__restrict__ dots, const double *__restrict__ Cm, double *__restrict__ coef, int r) {
  cfwd_coef_body((int)threadIdx.x, dots, Cm, coef, r);
}

// ---- launch-bound sizes: the WHOLE quasi-Newton apply in ONE launch -------------------------------------------------
// dots -> finalize -> coefficients -> combine are 4 dependent launches, ~3.5 us each on MI355X whatever their size; for
// n <= 2^17 the data movement is a fraction of that. Here (DESIGN §9 of round 2, built in round 3) every workgroup
//   1. keeps its slice of x in registers and reduces it against its slice of the <= 40 panel columns,
//   2. publishes the partial dots as self-validating 64-bit agent-scope stores into exchange slots (an empty slot is a
//      NaN payload arithmetic cannot produce; NaN partials are canonicalised) — NO fence: an agent-scope release would
//      write back the L2 of all 8 XCDs (profiles/r01_sweep_fuse.txt), the protocol of the single-launch Householder,
//   3. polls until all G <= 64 co-resident workgroups have published, sums every column's partials in a fixed order
//      (lane i <- workgroup i, one fixed DPP tree: bit-identical in every workgroup),
//   4. runs the operator's coefficient recurrence on one wave — redundantly per workgroup — from Gram data it staged
//      in LDS while the dots were in flight, and
//   5. applies the columns to its slice in the reference's elementwise order (the formulas of combine_kernel) and
//      writes res.
// Two slot sets alternate by an epoch word in device memory (graph-replay safe): a launch uses set e, re-arms set 1 - e
// for its successor, workgroup 0 flips e once it has seen every partial. One fused launch in flight per ctx (all work of
// a ctx is stream-ordered: mxlo_ctx_set_stream). Not used with an all-reduce hook (a host callback sits between 3 and 4).
// kQnfMaxCols = 40 (2*mem at mem <= 20: the launch-bound defaults of the callers), kQnfMaxGrid = 256 workgroups (round 4;
// 64 before: n <= 131072 doubles), kQnfSlots: common.h. The gathered partials live in DYNAMIC LDS (ncol x grid doubles,
// at most kQnfMaxPart = 6144: 48 KiB next to the 8 KiB of static LDS).
constexpr int kQnfMaxPart = 6144;
template <typename T>
struct QnfCols {
  const T *p[kQnfMaxCols];
};
struct QnfArgs {
  int kind;                                  // MXLO_QN_LBFGS_INV / _FWD (compact) / _LSR1
  int ncol, nfirst, use_gamma;
  double gamma, alpha, beta, shift;
  const double *SY, *YS, *YY;                // inverse: Gram matrices (mem x mem)
  double *alpha_out;
  const double *Cm;                          // compact forward: r x 2r coefficients
  const double *as_;                         // L-SR1
  int ct_f32;
};

// U vectors per lane and NB columns per batch: (4, 8) in general; (2, 12) when 8 < ncol <= 12 and the vector is short enough
// for 64 workgroups of half the slice — then ALL columns are one batch: one round trip in the dots phase and one in the
// combine phase instead of two each (L-BFGS m = 5 / 6: 10 / 12 columns); (1, 20) for 13 .. 40 columns on vectors of at most
// 64 workgroups of one vector per lane (L-BFGS m = 10: one batch instead of three; m = 20: two instead of five).
template <typename T, typename CA, typename CB, int KIND, bool BETA0, int U, int NB>
__global__ void __launch_bounds__(kBlock)
qn_apply_fused_kernel(T *__restrict__ res, QnfCols<T> cols, const T *__restrict__ x, int64_t n,
                      unsigned long long *__restrict__ slots, QnfArgs F, OrdArgs O, unsigned long long ticks,
                      unsigned *__restrict__ fault, int drop) {
  constexpr int VEC = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = (int)gridDim.x, b = (int)blockIdx.x;
  const int ncol = F.ncol, na = O.na;
  __shared__ double red[kBlock / 64][kQnfMaxCols];
  extern __shared__ double spart[];                         // [ncol][G] gathered partials
  __shared__ double sdots[kQnfMaxCols];
  __shared__ double scoef[kQnfMaxCols];
  __shared__ double sg1[kQnfMaxCols * kQnfMaxCols / 2];     // inverse: s_i'y_j by ord position (na x na); forward: Cm (r x 2r)
  __shared__ double sg2[kQnfMaxCols * kQnfMaxCols / 4];     // inverse: y_i'y_j by ord position; L-SR1: as by ord position
  unsigned long long *epoch = slots + 2 * kQnfSlots;
  const unsigned e = (unsigned)__hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u;
  unsigned long long *mine = slots + e * kQnfSlots, *other = slots + (1u - e) * kQnfSlots;
  for (int i = b * kBlock + tid; i < kQnfSlots; i += G * kBlock)       // re-arm the other set for the next launch
    __hip_atomic_store(other + i, kSlotEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- stage the Gram data the coefficient recurrence reads (loads overlap the dots below)
  if constexpr (KIND == MXLO_QN_LBFGS_INV) {
    const int mem = O.mem;
    for (int p = tid; p < na * na; p += kBlock) {
      const int i = O.ord[p / na], j = O.ord[p % na];                  // slots
      sg1[p] = O.age[j] >= O.age[i] ? F.SY[i + (int64_t)j * mem] : F.YS[j + (int64_t)i * mem];   // s_i'y_j
      sg2[p] = O.age[j] >= O.age[i] ? F.YY[i + (int64_t)j * mem] : F.YY[j + (int64_t)i * mem];   // y_i'y_j
    }
  } else if constexpr (KIND == MXLO_QN_LBFGS_FWD) {
    for (int p = tid; p < na * 2 * na; p += kBlock) sg1[p] = F.Cm[p];
  } else {
    if (tid < na) sg2[tid] = F.as_[O.ord[tid]];
  }
  // ---- 1. this workgroup's slice: U vectors per lane, x kept in registers
  const int64_t base = ((int64_t)b * kBlock * U + tid) * VEC;
  T xe[U][VEC];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + (int64_t)u * kBlock * VEC;
    if (i + VEC <= n) {
      const V xv = *reinterpret_cast<const V *>(x + i);
#pragma unroll
      for (int k = 0; k < VEC; ++k) xe[u][k] = xv[k];
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) xe[u][k] = i + k < n ? x[i + k] : T(0);
    }
  }
  // A workgroup whose whole slice lies inside n (all but possibly the last) runs branch-free: every load of a batch is
  // an unconditional 16-byte load, so the 8 x U loads of a batch really are in flight together (a per-load bounds
  // branch makes the compiler wait at every merge point: one memory round trip per LOAD). The last workgroup takes the
  // guarded form.
  const bool full = ((int64_t)(b + 1) * kBlock * U) * VEC <= n;
  auto ldv = [&](const T *p, int64_t i) -> V {   // guarded: 16-byte load, or element loads with zero fill
    if (i + VEC <= n) return *reinterpret_cast<const V *>(p + i);
    V v;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = i + k < n ? p[i + k] : T(0);
    return v;
  };
  // REUSE: the narrow shapes keep the LAST dots batch in registers through the exchange, and the combine phase applies
  // it from there instead of reading those columns again (forward L-BFGS and L-SR1: same column order in both phases;
  // with a single batch — the case these shapes exist for — the combine then issues no column load at all)
  constexpr bool REUSE = U < 4;
  // (the inverse operator's combine walks [y newest -> oldest, s oldest -> newest] where its dots are laid out
  //  [s.., y..] newest -> oldest: with REUSE the dots phase loads — and publishes — the columns in COMBINE order, and the
  //  sums are put back into dots order when they are written to sdots, which is all the coefficient recurrence reads)
  auto dots_index = [&](int c) -> int {        // position in dots order of the column the dots phase handles as number c
    if constexpr (KIND == MXLO_QN_LBFGS_INV && REUSE) return c < F.nfirst ? F.nfirst + c : 2 * F.nfirst - 1 - c;
    else return c;
  };
  V cv[NB][U];
  auto dots_batches = [&]<bool FULL>() {
    for (int c0 = 0; c0 < ncol; c0 += NB) {     // columns in batches of NB: one memory round trip per batch
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const T *p = cols.p[dots_index(c0 + t < ncol ? c0 + t : ncol - 1)];      // clamp: a valid (unused) column instead of a branch
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t i = base + (int64_t)u * kBlock * VEC;
          if constexpr (FULL) cv[t][u] = *reinterpret_cast<const V *>(p + i);
          else cv[t][u] = ldv(p, i);
        }
      }
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc = fma((double)cv[t][u][k], (double)xe[u][k], acc);
        acc = wave_allsum(acc);      // DPP tree (a 64-bit shuffle tree is 12 dependent ds_bpermute per column)
        if (lane == 0 && c0 + t < ncol) red[wave][c0 + t] = acc;
      }
    }
  };
  if (full) dots_batches.template operator()<true>();
  else dots_batches.template operator()<false>();
  __syncthreads();
  // ---- 2. publish
  if (tid < ncol) {
    const double sv = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    unsigned long long bits = (unsigned long long)__double_as_longlong(sv);
    if (sv != sv) bits = kCanonicalNaN;                               // canonical NaN: never the empty marker
    if (b != drop) __hip_atomic_store(mine + tid * kQnfMaxGrid + b, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- 3. gather: every (column, workgroup) slot is polled by exactly one lane; all polls of a lane are in flight together
  for (int p = tid; p < ncol * G; p += kBlock) {
    const int c = p / G, w = p - c * G;
    // bounded wait (poll_slot, common.h): a peer that never becomes resident ends as NaN + the ctx fault word
    spart[c * G + w] = __longlong_as_double((long long)poll_slot(mine + c * kQnfMaxGrid + w, ticks, fault, kFaultQn));
  }
  __syncthreads();
  if (b == 0 && tid == 0)                                             // every workgroup has read e: flip for the next launch
    __hip_atomic_store(epoch, (unsigned long long)(1u - e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int c = wave; c < ncol; c += kBlock / 64) {                    // fixed order: lane i <- workgroup i, one DPP tree
    double pv = lane < G ? spart[c * G + lane] : 0.0;                 // up to 64 workgroups: one partial per lane (the
    for (int w = lane + 64; w < G; w += 64) pv += spart[c * G + w];   // bits of round 3); beyond: lane l adds l, l+64, ...
    const double v = wave_allsum(pv);
    if (lane == 0) sdots[dots_index(c)] = v;
  }
  __syncthreads();
  // ---- 4. coefficients (one wave, redundantly per workgroup; same recurrences as the *_coef_kernel bodies)
  if (wave == 0) {
    if constexpr (KIND == MXLO_QN_LBFGS_INV) {
      inv_coef_generic(lane, sdots, scoef, b == 0 ? F.alpha_out : nullptr, O,
                       [&](int i, int j) { return sg1[i * na + j]; }, [&](int i, int j) { return sg2[i * na + j]; });
    } else if constexpr (KIND == MXLO_QN_LBFGS_FWD) {
      cfwd_coef_body(lane, sdots, sg1, scoef, na);
    } else {
      if (lane < na) {
        const double d = rnd(sdots[lane], O.is_f32), as = rnd(sg2[lane], O.is_f32);
        scoef[lane] = F.ct_f32 ? (double)(((float)F.alpha * (float)d) / (float)as) : (F.alpha * d) / as;
      }
    }
  }
  __syncthreads();
  // ---- 5. combine on this workgroup's slice: the elementwise formulas (and rounding points) of combine_kernel
  const T g = (T)F.gamma;
  const CA al = (CA)F.alpha;
  const CB be = (CB)F.beta;
  // combine column c in terms of the dots columns: the inverse operator's dots run over [s.., y..] newest -> oldest, its
  // combine over y newest -> oldest, then s oldest -> newest; the other two use one order for both
  auto ccol = [&](int c) -> const T * {
    if constexpr (KIND == MXLO_QN_LBFGS_INV) return c < F.nfirst ? cols.p[F.nfirst + c] : cols.p[2 * F.nfirst - 1 - c];
    else return cols.p[c];
  };
  // prologue / one column / epilogue of the combine on E elements (formulas and rounding points of combine_kernel)
  auto prologue = [&]<int E>(const T (&xq)[E], const T (&rq)[E], T (&q)[E]) {
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if constexpr (KIND == MXLO_QN_LBFGS_FWD) q[k] = F.use_gamma ? xq[k] / g : xq[k];
      else if constexpr (KIND == MXLO_QN_LBFGS_INV) q[k] = xq[k];
      else q[k] = fin_ab<T, CA, CB, BETA0>((al * (CA)xq[k]) / (CA)g, be, BETA0 ? T(0) : rq[k]);     // lsr1.jl:93
    }
  };
  auto column = [&]<int E>(int c, T (&q)[E], const T (&ce)[E]) {
    if constexpr (KIND == MXLO_QN_LBFGS_INV) {
      if (c == F.nfirst && F.use_gamma) {
#pragma unroll
        for (int k = 0; k < E; ++k) q[k] = q[k] * g;                                                 // lbfgs.jl:139
      }
      const T cc = (T)scoef[c];
      if (c < F.nfirst) {
#pragma unroll
        for (int k = 0; k < E; ++k) q[k] = q[k] - (cc * ce[k]);                                      // lbfgs.jl:135
      } else {
#pragma unroll
        for (int k = 0; k < E; ++k) q[k] = q[k] + (cc * ce[k]);                                      // lbfgs.jl:146
      }
    } else if constexpr (KIND == MXLO_QN_LBFGS_FWD) {
      const T cc = (T)scoef[c];
#pragma unroll
      for (int k = 0; k < E; ++k) q[k] = q[k] + (cc * ce[k]);
    } else {
      const CA cc = (CA)scoef[c];
#pragma unroll
      for (int k = 0; k < E; ++k) q[k] = (T)((CA)q[k] + (cc * (CA)ce[k]));                           // lsr1.jl:103
    }
  };
  auto epilogue = [&]<int E>(const T (&xq)[E], const T (&rq)[E], const T (&q)[E], T (&out)[E]) {
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if constexpr (KIND == MXLO_QN_LSR1) out[k] = q[k];
      else out[k] = fin_ab<T, CA, CB, BETA0>(al * (CA)q[k], be, BETA0 ? T(0) : rq[k]);               // lbfgs.jl:150,198
      if (F.shift != 0.0) out[k] = out[k] + ((T)F.shift * xq[k]);                                    // shifted_operators.jl:21-23
    }
  };
  if (full) {
    // all U vectors of the slice advance through the columns TOGETHER: a batch is 8 columns x U vectors = 32 loads in
    // flight and ncol / 8 round trips in all. Every register array below is indexed by compile-time constants only (a
    // run-time column index into the batch would put it in scratch memory).
    constexpr int E = U * VEC;
    T xq[E], rq[E], q[E], out[E];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      V rv;
      if constexpr (!BETA0) rv = *reinterpret_cast<const V *>(res + base + (int64_t)u * kBlock * VEC);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        xq[u * VEC + k] = xe[u][k];
        rq[u * VEC + k] = BETA0 ? T(0) : rv[k];
      }
    }
    prologue.template operator()<E>(xq, rq, q);
    for (int c0 = 0; c0 < ncol; c0 += NB) {
      V cvb[NB][U];
      const bool kept = REUSE && c0 + NB >= ncol;                      // the batch the dots phase left in registers
      if (!kept) {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const T *p = ccol(c0 + t < ncol ? c0 + t : ncol - 1);        // clamp: a valid (unused) column instead of a branch
#pragma unroll
          for (int u = 0; u < U; ++u) cvb[t][u] = *reinterpret_cast<const V *>(p + base + (int64_t)u * kBlock * VEC);
        }
      }
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        if (c0 + t < ncol) {
          T ce[E];
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < VEC; ++k) ce[u * VEC + k] = kept ? cv[t][u][k] : cvb[t][u][k];
          column.template operator()<E>(c0 + t, q, ce);
        }
      }
    }
    epilogue.template operator()<E>(xq, rq, q, out);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      V ov;
#pragma unroll
      for (int k = 0; k < VEC; ++k) ov[k] = out[u * VEC + k];
      *reinterpret_cast<V *>(res + base + (int64_t)u * kBlock * VEC) = ov;
    }
  } else {
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + (int64_t)u * kBlock * VEC;
      for (int k = 0; k < VEC; ++k) {
        if (i + k >= n) break;
        const T xq[1] = {xe[u][k]};
        const T rq[1] = {BETA0 ? T(0) : res[i + k]};
        T q[1], out[1];
        prologue.template operator()<1>(xq, rq, q);
        for (int c = 0; c < ncol; ++c) {
          const T ce[1] = {ccol(c)[i + k]};
          column.template operator()<1>(c, q, ce);
        }
        epilogue.template operator()<1>(xq, rq, q, out);
        res[i + k] = out[0];
      }
    }
  }
}

// true: the whole apply was issued as ONE launch
template <typename T>
bool try_fused_apply(mxlo_qn *h, T *res, const T *const *cols, const T *x, QnfArgs F, const OrdArgs &O, int32_t flags,
                     int32_t *status) {
  mxlo_ctx *ctx = h->ctx;
  constexpr int VEC = Vec16<T>::N;
  // 9 .. 12 columns on a vector short enough for 64 workgroups of 2 vectors per lane: all columns in ONE batch
  auto grid_of = [&](int u) { return (h->n + (int64_t)kBlock * u * VEC - 1) / ((int64_t)kBlock * u * VEC); };
  const bool small = ctx->tune.qn_fused_batch12 && F.ncol <= 12 && grid_of(2) <= kQnfMaxGrid;
  // (<= 8 columns: one batch either way, but this shape keeps it in registers for the combine phase)
  // 13 .. 40 columns on a vector short enough for 64 workgroups of ONE vector per lane: batches of 20 columns
  const bool tiny = ctx->tune.qn_fused_batch12 && F.ncol > 12 && grid_of(1) <= kQnfMaxGrid;
  const int U = small ? 2 : (tiny ? 1 : 4);
  const int64_t per = (int64_t)kBlock * U * VEC;
  const int64_t grid = (h->n + per - 1) / per;
  if (!ctx->tune.qn_fused_small || ctx->allreduce || F.ncol < 1 || F.ncol > kQnfMaxCols || grid > kQnfMaxGrid ||
      grid > ctx->tune.qn_fused_max_grid || (int64_t)F.ncol * grid > kQnfMaxPart ||
      grid > ctx->num_cu || !ctx->qslots || !ctx->fault_dev || ((((uintptr_t)x) | ((uintptr_t)res)) & 15u) != 0)
    return false;
  if ((*status = fused_fault_check(ctx)) != MXLO_OK) return true;    // an earlier timed-out single-launch apply: reported here
  QnfCols<T> fc;
  for (int c = 0; c < F.ncol; ++c) {
    if ((((uintptr_t)cols[c]) & 15u) != 0) return false;
    fc.p[c] = cols[c];
  }
  bool fits = true;
  *status = dispatch_ab<T>(F.beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    auto go2 = [&]<int KIND, int UU, int NB>() {
      const size_t lds = sizeof(double) * (size_t)F.ncol * (size_t)grid;
      // (occupancy is asked for the LARGEST dynamic LDS this kernel is ever launched with: the answer is cached)
      if (!(fits = coresident<qn_apply_fused_kernel<T, CA, CB, KIND, B0, UU, NB>>(ctx, grid, sizeof(double) * kQnfMaxPart))) return;
      hipLaunchKernelGGL((qn_apply_fused_kernel<T, CA, CB, KIND, B0, UU, NB>), dim3((unsigned)grid), dim3(kBlock), lds, ctx->stream,
                         res, fc, x, h->n, ctx->qslots, F, O, fused_timeout_ticks(ctx), ctx->fault_dev,
                         ctx->tune.fused_debug_drop);
    };
    auto go = [&]<int KIND>() {
      if (small) go2.template operator()<KIND, 2, 12>();
      else if (tiny) go2.template operator()<KIND, 1, 20>();
      else go2.template operator()<KIND, 4, 8>();
    };
    if (F.kind == MXLO_QN_LBFGS_INV) go.template operator()<MXLO_QN_LBFGS_INV>();
    else if (F.kind == MXLO_QN_LBFGS_FWD) go.template operator()<MXLO_QN_LBFGS_FWD>();
    else go.template operator()<MXLO_QN_LSR1>();
    if (!fits) return MXLO_OK;          // not co-resident on this device: the four-launch apply runs instead
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
  return fits;
}

// ---- cache-resident sizes: the whole apply as ONE PERSISTENT launch ----------------------------------------------------
// Between the single-launch sizes above (one slice per workgroup, n <= 2^19 doubles) and the HBM-bound ones the panel is
// tens to a few hundred MB: it fits, or nearly fits, the 256 MiB Infinity Cache, and the four-launch apply loses 10-20 %
// to two grid fills / drains and two dependent tiny launches (profiles/r05_tune_persist.txt: measured ceilings and the
// shape sweep this kernel's geometry comes from). Here ONE workgroup of kPersistBlock threads per CU owns a contiguous
// run of chunks (a chunk = one 16-byte vector per lane):
//   1. dots, batch-major: kPersistNB columns at a time over the workgroup's chunks, one f64 accumulator per column and
//      lane (x is re-read per batch: it sits in L2),
//   2. the exchange of qn_apply_fused_kernel (same slots, epoch word and bounded wait): partials published as
//      self-validating agent-scope stores, gathered by every workgroup in a fixed order,
//   3. the operator's coefficient recurrence, redundantly per workgroup,
//   4. combine, chunk-major and BACK TO FRONT: the chunks the dots phase touched last are the likeliest to still sit in
//      this XCD's L2 / the Infinity Cache (panels above 256 MiB: 146 -> 122 us at n = 2^22, 10 columns).
// Same elementwise formulas and rounding points as combine_kernel; the dots are summed in a different (fixed) order than
// the four-launch schedule's, so the two agree to rounding, not to the bit (both are deterministic run to run).
constexpr int kPersistBlock = 512;   // 8 waves per CU, one workgroup per CU
constexpr int kPersistNB = 10;       // columns per batch (10 independent 16-byte loads per lane in flight)
constexpr int kPersistTileBytes = kPersistBlock * 16, kPersistLdsTiles = 18;   // LDS keep: 18 tiles of 8 KiB next to ~13 KiB of static LDS

template <typename T, typename CA, typename CB, int KIND, bool BETA0>
__global__ void __launch_bounds__(kPersistBlock)
qn_apply_persist_kernel(T *__restrict__ res, QnfCols<T> cols, const T *__restrict__ x, int64_t n, int cpw,
                        unsigned long long *__restrict__ slots, QnfArgs F, OrdArgs O, unsigned long long ticks,
                        unsigned *__restrict__ fault, int drop, int reverse, int prefetch, int lds_k) {
  constexpr int VEC = Vec16<T>::N, NB = kPersistNB, NW = kPersistBlock / 64;
  using V = typename Vec16<T>::type;
  // LDS KEEP (round 6; lds_k >= 0): what the dots phase streams through its registers is also parked in the CU's LDS — x for
  // every full chunk of this workgroup and the first lds_k columns of the COMBINE order (one 8-KiB tile per (column, chunk),
  // a lane's own 16 bytes: conflict-free b128 accesses) — and the combine phase takes them from there instead of from the
  // L2 / the Infinity Cache; batches 2.. of the dots phase read x from it as well. Same loads' values, same arithmetic in the
  // same order: bit-identical to lds_k = -1. Tile (slot, j) = slot * cpw + j, slot 0 = x, slot 1 + c = combine column c.
  extern __shared__ __align__(16) unsigned char persist_dyn[];
  const bool lds_on = lds_k >= 0;
  auto ltile = [&](int slot, int64_t j) -> V * {
    return reinterpret_cast<V *>(persist_dyn) + ((int64_t)slot * cpw + j) * kPersistBlock + threadIdx.x;
  };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = (int)gridDim.x, b = (int)blockIdx.x;
  const int ncol = F.ncol, na = O.na;
  __shared__ double red[NW][kQnfMaxCols];
  __shared__ double sdots[kQnfMaxCols];
  __shared__ double scoef[kQnfMaxCols];
  __shared__ double sg1[kQnfMaxCols * kQnfMaxCols / 2];     // inverse: s_i'y_j by ord position; forward: Cm (r x 2r)
  __shared__ double sg2[kQnfMaxCols * kQnfMaxCols / 4];     // inverse: y_i'y_j by ord position; L-SR1: as by ord position
  unsigned long long *epoch = slots + 2 * kQnfSlots;
  const unsigned e = (unsigned)__hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u;
  unsigned long long *mine = slots + e * kQnfSlots, *other = slots + (1u - e) * kQnfSlots;
  for (int i = b * kPersistBlock + tid; i < kQnfSlots; i += G * kPersistBlock)      // re-arm the other set for the next launch
    __hip_atomic_store(other + i, kSlotEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if constexpr (KIND == MXLO_QN_LBFGS_INV) {
    const int mem = O.mem;
    for (int p = tid; p < na * na; p += kPersistBlock) {
      const int i = O.ord[p / na], j = O.ord[p % na];
      sg1[p] = O.age[j] >= O.age[i] ? F.SY[i + (int64_t)j * mem] : F.YS[j + (int64_t)i * mem];
      sg2[p] = O.age[j] >= O.age[i] ? F.YY[i + (int64_t)j * mem] : F.YY[j + (int64_t)i * mem];
    }
  } else if constexpr (KIND == MXLO_QN_LBFGS_FWD) {
    for (int p = tid; p < na * 2 * na; p += kPersistBlock) sg1[p] = F.Cm[p];
  } else {
    if (tid < na) sg2[tid] = F.as_[O.ord[tid]];
  }
  // this workgroup's chunks: [ch0, ch1) of kPersistBlock vectors each; only the very last chunk of the vector can hold
  // lanes beyond n (or a partial last vector): it takes the guarded loads
  const int64_t nvec = (n + VEC - 1) / VEC;
  const int64_t nchunks = (nvec + kPersistBlock - 1) / kPersistBlock;
  const int64_t ch0 = (int64_t)b * cpw < nchunks ? (int64_t)b * cpw : nchunks;
  const int64_t ch1 = ch0 + cpw < nchunks ? ch0 + cpw : nchunks;
  auto chunk_full = [&](int64_t ch) { return (ch + 1) * kPersistBlock * VEC <= n; };
  auto ldv = [&](const T *p, int64_t i) -> V {   // guarded: 16-byte load, or element loads with zero fill (i: element index)
    if (i + VEC <= n) return *reinterpret_cast<const V *>(p + i);
    V v;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = i + k < n ? p[i + k] : T(0);
    return v;
  };
  // ---- 1. dots (columns in DOTS order: cols.p[c])
  for (int c0 = 0; c0 < ncol; c0 += NB) {
    double acc[NB];
    const T *cp[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      acc[t] = 0.0;
      cp[t] = cols.p[c0 + t < ncol ? c0 + t : ncol - 1];      // clamp: a valid (unused) column instead of a branch
    }
    for (int64_t ch = ch0; ch < ch1; ++ch) {
      const int64_t i = (ch * kPersistBlock + tid) * VEC;
      V xv, cv[NB];
      const bool keep = lds_on && chunk_full(ch);
      if (chunk_full(ch)) {
        if (keep && c0 > 0) xv = *ltile(0, ch - ch0);
        else xv = *reinterpret_cast<const V *>(x + i);
#pragma unroll
        for (int t = 0; t < NB; ++t) cv[t] = *reinterpret_cast<const V *>(cp[t] + i);
      } else {
        xv = ldv(x, i);
#pragma unroll
        for (int t = 0; t < NB; ++t) cv[t] = ldv(cp[t], i);
      }
      if (keep && c0 == 0) *ltile(0, ch - ch0) = xv;
#pragma unroll
      for (int t = 0; t < NB; ++t) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[t] = fma((double)cv[t][k], (double)xv[k], acc[t]);
        if (keep) {                              // dots column c0 + t is combine column cc: parked when cc < lds_k
          const int d = c0 + t;
          int cc = d;
          if constexpr (KIND == MXLO_QN_LBFGS_INV) cc = d >= F.nfirst ? d - F.nfirst : 2 * F.nfirst - 1 - d;
          if (d < ncol && cc < lds_k) *ltile(1 + cc, ch - ch0) = cv[t];
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
  // combine column c in terms of the dots columns
  auto ccol = [&](int c) -> const T * {
    if constexpr (KIND == MXLO_QN_LBFGS_INV) return c < F.nfirst ? cols.p[F.nfirst + c] : cols.p[2 * F.nfirst - 1 - c];
    else return cols.p[c];
  };
  // PREFETCH across the exchange (tune key qn_persist_prefetch, OFF by default): x and the first batch of columns of the
  // chunk the combine phase takes first are requested NOW — they need no coefficient — and travel while the partials are
  // exchanged and the recurrence runs. Measured: no gain — 1.5 us slower at n = 2^19 .. 2^20 (loads return in order, so the
  // first poll of the exchange waits behind the 11 prefetch loads), neutral above (profiles/r05_bench_mid_apply.txt).
  const int64_t chp = reverse ? ch1 - 1 : ch0;
  const bool pre = prefetch && !lds_on && ch1 > ch0 && chunk_full(chp);
  V pxv, pcv[NB];
  if (pre) {
    const int64_t ip = (chp * kPersistBlock + tid) * VEC;
    pxv = *reinterpret_cast<const V *>(x + ip);
#pragma unroll
    for (int t = 0; t < NB; ++t) pcv[t] = *reinterpret_cast<const V *>(ccol(t < ncol ? t : ncol - 1) + ip);
  }
  // ---- 2. publish, gather (fixed order: lane l adds workgroups l, l + 64, ..., then one DPP tree)
  if (tid < ncol) {
    const double sv = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) +
                      ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
    unsigned long long bits = (unsigned long long)__double_as_longlong(sv);
    if (sv != sv) bits = kCanonicalNaN;
    if (b != drop) __hip_atomic_store(mine + tid * kQnfMaxGrid + b, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int c = wave; c < ncol; c += NW) {
    constexpr int kPer = kQnfMaxGrid / 64;
    const unsigned long long *row = mine + c * kQnfMaxGrid;
    unsigned long long bits[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j)      // first look: all of a lane's slots in flight together
      bits[j] = lane + 64 * j < G ? __hip_atomic_load(row + lane + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    double pv = 0.0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if (lane + 64 * j < G) {
        if (bits[j] == kSlotEmpty) bits[j] = poll_slot(row + lane + 64 * j, ticks, fault, kFaultQn);   // bounded wait
        pv += __longlong_as_double((long long)bits[j]);
      }
    }
    const double v = wave_allsum(pv);
    if (lane == 0) sdots[c] = v;
  }
  __syncthreads();
  if (b == 0 && tid == 0)                                             // every workgroup has read e: flip for the next launch
    __hip_atomic_store(epoch, (unsigned long long)(1u - e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- 3. coefficients (one wave, redundantly per workgroup; the bodies of the *_coef_kernel launches)
  if (wave == 0) {
    if constexpr (KIND == MXLO_QN_LBFGS_INV) {
      inv_coef_generic(lane, sdots, scoef, b == 0 ? F.alpha_out : nullptr, O,
                       [&](int i, int j) { return sg1[i * na + j]; }, [&](int i, int j) { return sg2[i * na + j]; });
    } else if constexpr (KIND == MXLO_QN_LBFGS_FWD) {
      cfwd_coef_body(lane, sdots, sg1, scoef, na);
    } else {
      if (lane < na) {
        const double d = rnd(sdots[lane], O.is_f32), as = rnd(sg2[lane], O.is_f32);
        scoef[lane] = F.ct_f32 ? (double)(((float)F.alpha * (float)d) / (float)as) : (F.alpha * d) / as;
      }
    }
  }
  __syncthreads();
  // ---- 4. combine, last chunk first (columns in COMBINE order)
  const T g = (T)F.gamma;
  const CA al = (CA)F.alpha;
  const CB be = (CB)F.beta;
  // One chunk; FULL: every lane's vector lies inside n — unconditional 16-byte accesses, so the NB column loads of a
  // batch are in flight together (a per-load bounds branch makes the compiler drain the memory queue before every load).
  // PRE (a FULL chunk): x and the first batch of columns are the registers prefetched across the exchange.
  // KEEP (a FULL chunk, lds_k >= 0): x and the first lds_k columns come out of the LDS tiles the dots phase parked.
  auto combine_chunk = [&]<bool FULL, bool PRE = false, bool KEEP = false>(int64_t ch) {
    const int64_t i = (ch * kPersistBlock + tid) * VEC;
    if constexpr (!FULL) {
      if (i >= n) return;
    }
    V xv, rv;
    if constexpr (FULL) {
      if constexpr (PRE) xv = pxv;
      else if constexpr (KEEP) xv = *ltile(0, ch - ch0);
      else xv = *reinterpret_cast<const V *>(x + i);
      if constexpr (!BETA0) rv = *reinterpret_cast<const V *>(res + i);
    } else {
      xv = ldv(x, i);
      if constexpr (!BETA0) rv = ldv(res, i);
    }
    T q[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if constexpr (KIND == MXLO_QN_LBFGS_FWD) q[k] = F.use_gamma ? xv[k] / g : xv[k];
      else if constexpr (KIND == MXLO_QN_LBFGS_INV) q[k] = xv[k];
      else q[k] = fin_ab<T, CA, CB, BETA0>((al * (CA)xv[k]) / (CA)g, be, BETA0 ? T(0) : rv[k]);     // lsr1.jl:93
    }
    for (int c0 = 0; c0 < ncol; c0 += NB) {
      V cv[NB];
      // first batch of a KEEP chunk: columns [0, KK) from the LDS, [KK, NB) from memory — KK a compile-time count per case, so
      // the loads of a batch stay a straight line (a per-load `parked ? LDS : memory` branch would drain the queue before each)
      auto first_batch = [&]<int KK>() {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          if (t < KK) cv[t] = *ltile(1 + t, ch - ch0);
          else cv[t] = *reinterpret_cast<const V *>(ccol(t < ncol ? t : ncol - 1) + i);
        }
      };
      if (KEEP && c0 == 0) {
        switch (lds_k) {
          case 10: first_batch.template operator()<10>(); break;
          case 8: first_batch.template operator()<8>(); break;
          case 5: first_batch.template operator()<5>(); break;
          case 3: first_batch.template operator()<3>(); break;
          case 2: first_batch.template operator()<2>(); break;
          case 1: first_batch.template operator()<1>(); break;
          default: first_batch.template operator()<0>(); break;
        }
      } else {
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const T *p = ccol(c0 + t < ncol ? c0 + t : ncol - 1);      // clamp: a valid (unused) column instead of a branch
          if constexpr (PRE) cv[t] = c0 == 0 ? pcv[t] : *reinterpret_cast<const V *>(p + i);
          else if constexpr (FULL) cv[t] = *reinterpret_cast<const V *>(p + i);
          else cv[t] = ldv(p, i);
        }
      }
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int c = c0 + t;
        if (c < ncol) {
          if constexpr (KIND == MXLO_QN_LBFGS_INV) {
            if (c == F.nfirst && F.use_gamma) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) q[k] = q[k] * g;                                         // lbfgs.jl:139
            }
            const T cc = (T)scoef[c];
            if (c < F.nfirst) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) q[k] = q[k] - (cc * cv[t][k]);                           // lbfgs.jl:135
            } else {
#pragma unroll
              for (int k = 0; k < VEC; ++k) q[k] = q[k] + (cc * cv[t][k]);                           // lbfgs.jl:146
            }
          } else if constexpr (KIND == MXLO_QN_LBFGS_FWD) {
            const T cc = (T)scoef[c];
#pragma unroll
            for (int k = 0; k < VEC; ++k) q[k] = q[k] + (cc * cv[t][k]);
          } else {
            const CA cc = (CA)scoef[c];
#pragma unroll
            for (int k = 0; k < VEC; ++k) q[k] = (T)((CA)q[k] + (cc * (CA)cv[t][k]));                // lsr1.jl:103
          }
        }
      }
    }
    V ov;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      T o;
      if constexpr (KIND == MXLO_QN_LSR1) o = q[k];
      else o = fin_ab<T, CA, CB, BETA0>(al * (CA)q[k], be, BETA0 ? T(0) : rv[k]);                    // lbfgs.jl:150,198
      if (F.shift != 0.0) o = o + ((T)F.shift * xv[k]);                                              // shifted_operators.jl:21-23
      ov[k] = o;
    }
    if constexpr (FULL) *reinterpret_cast<V *>(res + i) = ov;
    else {
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        if (i + k < n) res[i + k] = ov[k];
    }
  };
  for (int64_t kk = 0; kk < ch1 - ch0; ++kk) {
    const int64_t ch = reverse ? ch1 - 1 - kk : ch0 + kk;
    if (kk == 0 && pre) combine_chunk.template operator()<true, true>(ch);
    else if (chunk_full(ch) && lds_on) combine_chunk.template operator()<true, false, true>(ch);
    else if (chunk_full(ch)) combine_chunk.template operator()<true>(ch);
    else combine_chunk.template operator()<false>(ch);
  }
}

// raise the kernel's dynamic-LDS cap once per (instantiation, device)
template <auto Kernel>
bool persist_lds_attr_set(mxlo_ctx *ctx) {
  static std::atomic<int> state[64];                // 0 unknown, 1 set, 2 refused
  const int d = ctx->device & 63;
  int st = state[d].load(std::memory_order_relaxed);
  if (st == 0) {
    st = hipFuncSetAttribute((const void *)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLdsTiles * kPersistTileBytes) == hipSuccess ? 1 : 2;
    if (st == 2) (void)hipGetLastError();
    state[d].store(st, std::memory_order_relaxed);
  }
  return st == 1;
}

// true: the whole apply was issued as one persistent launch
template <typename T>
bool try_persist_apply(mxlo_qn *h, T *res, const T *const *cols, const T *x, QnfArgs F, const OrdArgs &O, int32_t flags,
                       int32_t *status) {
  mxlo_ctx *ctx = h->ctx;
  constexpr int VEC = Vec16<T>::N;
  const int64_t panel_bytes = (int64_t)sizeof(T) * h->n * F.ncol;
  const int grid = ctx->num_cu < kQnfMaxGrid ? ctx->num_cu : kQnfMaxGrid;
  if (!ctx->tune.qn_persist || ctx->allreduce || F.ncol < 1 || F.ncol > kQnfMaxCols || h->n < ctx->tune.qn_persist_min_n ||
      panel_bytes > ctx->tune.qn_persist_max_bytes || panel_bytes < ctx->tune.qn_persist_min_bytes || !ctx->qslots || !ctx->fault_dev ||
      ((((uintptr_t)x) | ((uintptr_t)res)) & 15u) != 0)
    return false;
  QnfCols<T> fc;
  for (int c = 0; c < F.ncol; ++c) {
    if ((((uintptr_t)cols[c]) & 15u) != 0) return false;
    fc.p[c] = cols[c];
  }
  if ((*status = fused_fault_check(ctx)) != MXLO_OK) return true;    // an earlier timed-out single-launch apply: reported here
  const int64_t nvec = (h->n + VEC - 1) / VEC;
  const int64_t nchunks = (nvec + kPersistBlock - 1) / kPersistBlock;
  const int cpw = (int)((nchunks + grid - 1) / grid);
  bool fits = true;
  *status = dispatch_ab<T>(F.beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    auto go = [&]<int KIND>() {
      if (!(fits = coresident<qn_apply_persist_kernel<T, CA, CB, KIND, B0>, kPersistBlock>(ctx, grid, (size_t)kPersistLdsTiles * kPersistTileBytes))) return;
      // LDS keep (tune key qn_persist_lds, default 1): x of every chunk + the first K combine columns stay in the CU's LDS across
      // the exchange — K = what 18 tiles leave after x, from the counts the kernel has a straight-line first batch for.
      int lds_k = -1;
      size_t dyn = (size_t)ctx->tune.qn_persist_lds_pad;   // (lds_pad: dynamic LDS the kernel never touches — a placement experiment)
      if (ctx->tune.qn_persist_lds && cpw <= kPersistLdsTiles) {
        int k = (kPersistLdsTiles - cpw) / cpw;
        if (k > F.ncol) k = F.ncol;
        lds_k = k >= 10 ? 10 : k >= 8 ? 8 : k >= 5 ? 5 : k >= 3 ? 3 : k;      // 0, 1, 2 as they are
        // worth it when the parked share is at least ~1/12 of a pass (profiles/r06_qn_persist_lds.txt: +5 ... 22 % at
        // n = 2^19 .. 2^21; x alone next to 20 - 40 columns measured -2 ... -3 %: the parking stores for nothing)
        if ((1 + lds_k) * 12 < F.ncol + 1) lds_k = -1;
        const size_t want = lds_k < 0 ? 0 : (size_t)(1 + lds_k) * cpw * kPersistTileBytes;
        if (want > dyn) dyn = want;
      }
      if (dyn > 48 * 1024 && !persist_lds_attr_set<qn_apply_persist_kernel<T, CA, CB, KIND, B0>>(ctx)) {
        if (lds_k < 0) { fits = false; return; }
        lds_k = -1;                                        // the cap could not be raised: the plain form
        dyn = 0;
      }
      hipLaunchKernelGGL((qn_apply_persist_kernel<T, CA, CB, KIND, B0>), dim3((unsigned)grid), dim3(kPersistBlock), dyn,
                         ctx->stream, res, fc, x, h->n, cpw, ctx->qslots, F, O, fused_timeout_ticks(ctx), ctx->fault_dev,
                         ctx->tune.fused_debug_drop, ctx->tune.qn_persist_reverse, ctx->tune.qn_persist_prefetch, lds_k);
    };
    if (F.kind == MXLO_QN_LBFGS_INV) go.template operator()<MXLO_QN_LBFGS_INV>();
    else if (F.kind == MXLO_QN_LBFGS_FWD) go.template operator()<MXLO_QN_LBFGS_FWD>();
    else go.template operator()<MXLO_QN_LSR1>();
    if (!fits) return MXLO_OK;
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  });
  return fits;
}

// ---- applies ------------------------------------------------------------------------------
template <typename T>
int32_t inv_mul_twopass(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags,
                 double shift = 0.0) {
  mxlo_ctx *ctx = h->ctx;
  OrdArgs O;
  fill_ord(h, O, /*newest_first=*/true);
  const int na = O.na;
  double *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef;
  CombineArgs<T> A;
  A.ncol = 2 * na;
  A.nfirst = na;
  A.use_gamma = h->scaling;
  A.gamma = h->scaling_factor;
  A.alpha = alpha;
  A.beta = beta;
  A.coef = coef;
  A.shift = shift;
  if (na > 0) {
    const T *cols[kMaxCols];
    for (int i = 0; i < na; ++i) {
      cols[i] = col<T>(h->S, h->ld, O.ord[i]);
      cols[na + i] = col<T>(h->Y, h->ld, O.ord[i]);
    }
    {  // launch-bound sizes: the whole apply in one launch
      QnfArgs F{};
      F.kind = MXLO_QN_LBFGS_INV;
      F.ncol = 2 * na;
      F.nfirst = na;
      F.use_gamma = h->scaling;
      F.gamma = h->scaling_factor;
      F.alpha = alpha;
      F.beta = beta;
      F.shift = shift;
      F.SY = h->dsc + h->lay.SY;
      F.YS = h->dsc + h->lay.YS;
      F.YY = h->dsc + h->lay.YY;
      F.alpha_out = h->dsc + h->lay.alpha;
      int32_t fst = MXLO_OK;
      if (try_persist_apply<T>(h, res, cols, x, F, O, flags, &fst)) return fst;
      if (try_fused_apply<T>(h, res, cols, x, F, O, flags, &fst)) return fst;
    }
    MXLO_TRY(panel_dots<T>(ctx, cols, 2 * na, x, h->n, dots));
    const size_t coef_lds = sizeof(double) * (size_t)(2 * na * na + 2 * na);
    if (coef_lds > 64 * 1024 && !ctx->inv_lds_attr_set) {   // na = 64 only: 65 KiB, above the default cap; the attribute is per device
      MXLO_HIP(hipFuncSetAttribute((const void *)inv_coef_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
      ctx->inv_lds_attr_set = true;
    }
    hipLaunchKernelGGL(inv_coef_kernel, dim3(1), dim3(kBlock), coef_lds, ctx->stream, dots, coef,
                       h->dsc + h->lay.SY, h->dsc + h->lay.YS, h->dsc + h->lay.YY,
                       h->dsc + h->lay.alpha, O);
    MXLO_LAUNCH_CHECK();
    for (int i = 0; i < na; ++i) {
      A.cols[i] = col<T>(h->Y, h->ld, O.ord[i]);               // y, newest -> oldest
      A.cols[na + i] = col<T>(h->S, h->ld, O.ord[na - 1 - i]); // s, oldest -> newest
    }
  }  // no pairs: ncol == 0 and `q .*= scaling_factor` still executes (γ == 1 after reset!)
  A.reverse = ctx->tune.combine_reverse;
  return launch_combine<T, CM_INV>(ctx, res, x, (const T *)nullptr, A, h->n, flags);
}

template <typename T>
int32_t inv_mul_reforder(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags,
                 double shift = 0.0) {
  // src/lbfgs.jl:127-153 statement by statement; q lives in h->tmp (data.Ax)
  mxlo_ctx *ctx = h->ctx;
  T *q = (T *)h->tmp;
  const int64_t n = h->n;
  MXLO_TRY((launch_map<T, 1, false, false>(ctx, q, x, (const T *)nullptr, n, CopyOp<T>{})));
  double *dots = h->dsc + h->lay.dots, *al = h->dsc + h->lay.alpha;
  const std::vector<int> ord = ord_host(h, true);
  const int na = (int)ord.size();
  for (int i = 0; i < na; ++i) {
    const int k = ord[i];
    const T *sk = col<T>(h->S, h->ld, k), *yk = col<T>(h->Y, h->ld, k);
    const T *cols[1] = {sk};
    MXLO_TRY(panel_dots<T>(ctx, cols, 1, q, n, dots));
    InvStep1Op<T> op{dots, al + k, (T)h->ys[k], T(0)};
    MXLO_TRY((launch_map<T, 1, true, false>(ctx, q, yk, (const T *)nullptr, n, op)));
  }
  if (h->scaling) {
    MXLO_TRY((launch_map<T, 0, true, false>(ctx, q, (const T *)nullptr, (const T *)nullptr, n,
                                            ScaleOp<T, T>{(T)h->scaling_factor})));
  }
  for (int i = na - 1; i >= 0; --i) {
    const int k = ord[i];
    const T *sk = col<T>(h->S, h->ld, k), *yk = col<T>(h->Y, h->ld, k);
    const T *cols[1] = {yk};
    MXLO_TRY(panel_dots<T>(ctx, cols, 1, q, n, dots));
    InvStep2Op<T> op{dots, al + k, (T)h->ys[k], T(0)};
    MXLO_TRY((launch_map<T, 1, true, false>(ctx, q, sk, (const T *)nullptr, n, op)));
  }
  MXLO_TRY((dispatch_ab<T>(beta, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
    AxpbyOp<T, CA, CB, B0> op{(CA)alpha, (CB)beta};
    return launch_map<T, 1, !B0, false>(ctx, res, q, (const T *)nullptr, n, op);
  })));
  if (shift != 0.0)  // reference order: the ShiftedOperator axpy! stays its own pass
    return mxlo_eye_mul(ctx, sizeof(T) == 8 ? MXLO_F64 : MXLO_F32, res, x, n, n, shift, 1.0, 0);
  return MXLO_OK;
}

template <typename T>
int32_t fwd_mul(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags,
                 double shift = 0.0) {
  mxlo_ctx *ctx = h->ctx;
  OrdArgs O;
  fill_ord(h, O, /*newest_first=*/false);
  const int na = O.na;
  double *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef;
  CombineArgs<T> A;
  A.ncol = 2 * na;
  A.nfirst = 0;
  A.use_gamma = h->scaling;
  A.gamma = h->scaling_factor;
  A.alpha = alpha;
  A.beta = beta;
  A.coef = coef;
  A.shift = shift;
  if (na > 0 && !h->A_valid) {  // compact form: the a_k panel does not exist; x/γ + [S B]·w
    for (int i = 0; i < na; ++i) {
      A.cols[i] = col<T>(h->S, h->ld, O.ord[i]);
      A.cols[na + i] = col<T>(h->B, h->ld, O.ord[i]);
    }
    {
      QnfArgs F{};
      F.kind = MXLO_QN_LBFGS_FWD;
      F.ncol = 2 * na;
      F.nfirst = 0;
      F.use_gamma = h->scaling;
      F.gamma = h->scaling_factor;
      F.alpha = alpha;
      F.beta = beta;
      F.shift = shift;
      F.Cm = h->dsc + h->lay.Cm;
      int32_t fst = MXLO_OK;
      if (try_persist_apply<T>(h, res, A.cols, x, F, O, flags, &fst)) return fst;
      if (try_fused_apply<T>(h, res, A.cols, x, F, O, flags, &fst)) return fst;
    }
    MXLO_TRY(panel_dots<T>(ctx, A.cols, 2 * na, x, h->n, dots));
    hipLaunchKernelGGL(cfwd_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, dots, h->dsc + h->lay.Cm, coef, na);
    MXLO_LAUNCH_CHECK();
    A.reverse = ctx->tune.combine_reverse;
    return launch_combine<T, CM_CFWD>(ctx, res, x, (const T *)nullptr, A, h->n, flags);
  }
  if (na > 0) {
    for (int i = 0; i < na; ++i) {  // pair order (b_k, a_k): coef = (bx, ax)
      A.cols[2 * i] = col<T>(h->B, h->ld, O.ord[i]);
      A.cols[2 * i + 1] = col<T>(h->A, h->ld, O.ord[i]);
    }
    MXLO_TRY(panel_dots<T>(ctx, A.cols, 2 * na, x, h->n, dots));  // dots against x: independent
    hipLaunchKernelGGL(copy_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, dots, coef, 2 * na,
                       (int)(h->dtype == MXLO_F32));
    MXLO_LAUNCH_CHECK();
  }
  A.reverse = ctx->tune.combine_reverse;
  return launch_combine<T, CM_FWD>(ctx, res, x, (const T *)nullptr, A, h->n, flags);
}

template <typename T>
int32_t lsr1_mul(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags,
                 double shift = 0.0) {
  mxlo_ctx *ctx = h->ctx;
  OrdArgs O;
  fill_ord(h, O, false);
  const int na = O.na;
  double *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef;
  CombineArgs<T> A;
  A.ncol = na;
  A.nfirst = 0;
  A.use_gamma = 1;
  A.gamma = h->scaling_factor;  // divided unconditionally (== 1 when scaling is off)
  A.alpha = alpha;
  A.beta = beta;
  A.coef = coef;
  A.shift = shift;
  if (na > 0) {
    for (int i = 0; i < na; ++i) A.cols[i] = col<T>(h->A, h->ld, O.ord[i]);
    const int ct_f32 = alpha_is_f64(sizeof(T), flags) ? 0 : 1;
    {
      QnfArgs F{};
      F.kind = MXLO_QN_LSR1;
      F.ncol = na;
      F.nfirst = 0;
      F.use_gamma = 1;
      F.gamma = h->scaling_factor;
      F.alpha = alpha;
      F.beta = beta;
      F.shift = shift;
      F.as_ = h->dsc + h->lay.as_;
      F.ct_f32 = ct_f32;
      int32_t fst = MXLO_OK;
      if (try_persist_apply<T>(h, res, A.cols, x, F, O, flags, &fst)) return fst;
      if (try_fused_apply<T>(h, res, A.cols, x, F, O, flags, &fst)) return fst;
    }
    MXLO_TRY(panel_dots<T>(ctx, A.cols, na, x, h->n, dots));
    hipLaunchKernelGGL(lsr1_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, dots, coef,
                       h->dsc + h->lay.as_, O, alpha, ct_f32);
    MXLO_LAUNCH_CHECK();
  }
  A.reverse = ctx->tune.combine_reverse;
  return launch_combine<T, CM_LSR1>(ctx, res, x, (const T *)nullptr, A, h->n, flags);
}

template <typename T>
int32_t qn_mul_t(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags,
                 double shift = 0.0) {
  eff_scalars(sizeof(T), flags, alpha, beta);
  if (h->big) {
    switch (h->kind) {
      case MXLO_QN_LBFGS_INV:
        return h->mode == MXLO_INV_REFORDER ? inv_mul_reforder<T>(h, res, x, alpha, beta, flags, shift)
                                            : inv_mul_big<T>(h, res, x, alpha, beta, flags, shift);
      case MXLO_QN_LBFGS_FWD: return fwd_mul_big<T>(h, res, x, alpha, beta, flags, shift);
      case MXLO_QN_LSR1: return lsr1_mul_big<T>(h, res, x, alpha, beta, flags, shift);
    }
    return MXLO_EINVAL;
  }
  switch (h->kind) {
    case MXLO_QN_LBFGS_INV:
      return h->mode == MXLO_INV_REFORDER ? inv_mul_reforder<T>(h, res, x, alpha, beta, flags, shift)
                                          : inv_mul_twopass<T>(h, res, x, alpha, beta, flags, shift);
    case MXLO_QN_LBFGS_FWD: return fwd_mul<T>(h, res, x, alpha, beta, flags, shift);
    case MXLO_QN_LSR1: return lsr1_mul<T>(h, res, x, alpha, beta, flags, shift);
  }
  return MXLO_EINVAL;
}

// ---- push! ------------------------------------------------------------------------------------
// One wave copies `count` doubles from device memory into the handle's mapped host buffer with system-scope stores, then
// publishes the sequence number behind them: the host sees the number only after the doubles.
__global__ void __launch_bounds__(kWave)
post_scalars_kernel(const double *__restrict__ dev, double *__restrict__ host, int count, unsigned long long seq) {
  const int t = threadIdx.x;
  if (t < count)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(host) + t, (unsigned long long)__double_as_longlong(dev[t]),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();   // every lane's store has left before lane 0 publishes (one wave: lockstep)
  if (t == 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(host) + kPinnedScalars, seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// Host side of a posted read-back: poll `nwords` sequence words (stride `stride` words) of the handle's pinned buffer
// until all carry `seq`. A poll that outlasts kPostSpinUs (a long push pass is still running, or the launch failed)
// hands over to the stream synchronisation, after which the words must be in place. If they are NOT (a platform whose
// mapped host memory does not see the device's stores — nothing observed, but nothing the library can rule out either)
// *arrived is false: the caller copies the same doubles from device memory (the stream is idle at that point) and
// posting is switched off for this handle. push_posted = 2 (debug) takes that path on purpose.
static int32_t await_posted(mxlo_qn *h, const unsigned long long *word, int nwords, int stride, unsigned long long seq,
                            int64_t bytes, bool *arrived) {
  mxlo_ctx *ctx = h->ctx;
  ApiCounters &c = api_counters();     // what the contract test counts: one device-to-host transfer, one wait
  ++c.n_d2h;
  c.n_d2h_bytes += bytes;
  constexpr int kPostSpinUs = 200;
  const bool pretend_lost = ctx->tune.push_posted == 2;
  const auto t0 = std::chrono::steady_clock::now();
  int have = 0;
  for (int spin = 0; have < nwords && !pretend_lost; ++spin) {
    if (__atomic_load_n(word + (int64_t)have * stride, __ATOMIC_ACQUIRE) == seq) { ++have; continue; }
    if ((spin & 63) == 63 &&
        std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kPostSpinUs)
      break;
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::this_thread::yield();
#endif
  }
  *arrived = true;
  if (have == nwords) {
    ++c.n_stream_sync;
    return MXLO_OK;
  }
  MXLO_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < nwords && *arrived; ++i)
    if (pretend_lost || __atomic_load_n(word + (int64_t)i * stride, __ATOMIC_ACQUIRE) != seq) *arrived = false;
  if (!*arrived) h->pinned_dev = nullptr;   // plain copies from now on
  return MXLO_OK;
}

// the doubles themselves from device memory after a posting that did not arrive (stream already synchronised)
static int32_t read_after_lost_post(mxlo_qn *h, const double *dev, double *host, int count) {
  --api_counters().n_d2h;   // still ONE transfer of these doubles as far as the contract counters go
  api_counters().n_d2h_bytes -= (int64_t)sizeof(double) * count;
  MXLO_HIP(hipMemcpyAsync(host, dev, sizeof(double) * count, hipMemcpyDeviceToHost, h->ctx->stream));
  MXLO_HIP(hipStreamSynchronize(h->ctx->stream));
  return MXLO_OK;
}

int32_t read_scalars(mxlo_qn *h, const double *dev, double *host, int count) {
  // A handful of doubles of replicated control state, once per push!. Round 3 copied them into the caller's pageable
  // stack (staged + synchronised by the runtime: 41-61 us per small push!), then into pinned memory of the handle
  // (31-37 us). Now (tune key push_posted) a one-wave kernel posts them into that memory — it is device-mapped — and the
  // host polls the sequence word: no copy engine, no completion signal, no driver call on the way back.
  mxlo_ctx *ctx = h->ctx;
  if (h->pinned_dev && count <= kPinnedScalars && ctx->tune.push_posted) {
    const unsigned long long seq = ++h->post_seq;
    hipLaunchKernelGGL(post_scalars_kernel, dim3(1), dim3(kWave), 0, ctx->stream, dev, h->pinned_dev, count, seq);
    MXLO_LAUNCH_CHECK();
    bool arrived = false;
    MXLO_TRY(await_posted(h, reinterpret_cast<const unsigned long long *>(h->pinned) + kPinnedScalars, 1, 1, seq,
                          (int64_t)sizeof(double) * count, &arrived));
    if (!arrived) return read_after_lost_post(h, dev, host, count);
    memcpy(host, h->pinned, sizeof(double) * count);
    return MXLO_OK;
  }
  double *dst = (h->pinned && count <= kPinnedScalars) ? h->pinned : host;
  MXLO_HIP(hipMemcpyAsync(dst, dev, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
  MXLO_HIP(hipStreamSynchronize(ctx->stream));
  if (dst != host) memcpy(host, dst, sizeof(double) * count);
  return MXLO_OK;
}

// ... the same for doubles that a finalize launch posted itself as (value, sequence number) pairs (PushExtras::post);
// `dev`: where the same doubles lie in device memory (contiguous), for a posting that did not arrive
static int32_t await_posted_pairs(mxlo_qn *h, const double *dev, double *host, int count, unsigned long long seq) {
  const unsigned long long *pairs = reinterpret_cast<const unsigned long long *>(h->pinned) + kPostPairs;
  bool arrived = false;
  MXLO_TRY(await_posted(h, pairs + 1, count, 2, seq, (int64_t)sizeof(double) * count, &arrived));
  if (!arrived) return read_after_lost_post(h, dev, host, count);
  for (int i = 0; i < count; ++i) memcpy(host + i, pairs + 2 * i, sizeof(double));
  return MXLO_OK;
}

template <typename T>
inline double rT(double v) { return sizeof(T) == 4 ? (double)(float)v : v; }

// ---- forward push!, Gram form --------------------------------------------------------------------
// The reference rebuilds every a_k with O(m^2) dot/axpy passes over n (src/lbfgs.jl:236-250). All of those
// dots are inner products between vectors of span{s_j, b_j}; with the Gram matrices S'S and Y'S kept up to
// date (3m dots per push) the recurrence runs on 2m-vectors of coefficients and ALL a_k are then formed
// in ONE pass  A = [S B] * C  (read 2r columns, write r columns).

// Z[lane][k] = <basis_lane, s_ord[k]> from the Gram matrices: lanes < r read rows of G0, lanes r .. 2r-1 rows of G1 (DIV:
// divided by sqrt(ys) of their slot — b_j = y_j / sqrt(ys_j)). The per-lane slot and its ys are looked up ONCE and all r
// loads of a lane are issued before the first use: written as a plain loop over k this fill was a chain of three
// dependent memory round trips per k (kernel-argument array at a per-lane index -> Gram entry -> ys) and, at r = 20,
// ~35 of the forward push!'s 68 us at launch-bound sizes.
template <bool DIV>
__device__ __forceinline__ void fill_Z(double (*Z)[kMaxMemFwd], const double *G0, const double *G1, const OrdArgs &O) {
  const int lane = threadIdx.x, r = O.na, mem = O.mem, w = 2 * r;
  if (lane < w) {
    const int j = lane < r ? lane : lane - r;
    const int oj = O.ord[j];
    const double *row = (lane < r ? G0 : G1) + oj * mem;
    double g[kMaxMemFwd];
#pragma unroll
    for (int k = 0; k < kMaxMemFwd; ++k)
      if (k < r) g[k] = row[O.ord[k]];
    if constexpr (DIV) {
      if (lane >= r) {
        const double sq = sqrt(O.ys[oj]);
#pragma unroll
        for (int k = 0; k < kMaxMemFwd; ++k)
          if (k < r) g[k] = g[k] / sq;
      }
    }
#pragma unroll
    for (int k = 0; k < kMaxMemFwd; ++k)
      if (k < r) Z[lane][k] = g[k];
  }
}

// The Gram update of a push! as a PROLOGUE of the single-wave coefficient kernel that consumes it (afwd_coef_kernel,
// asr1_coef_kernel): launch-bound pushes lose one dependent launch (two for L-SR1, whose own-pair entries ride along).
// Same stores as gram_update_kernel (+ gram_self_kernel); the coefficient code reads SS / YSf after a barrier.
struct GramUpd {
  double *SS = nullptr, *YSf = nullptr, *YY = nullptr, *tmp = nullptr;
  int on = 0, mem = 0, ins = 0;
  int self = 0;                 // L-SR1: tmp[ins], tmp[m+ins], tmp[2m+ins], tmp[3m+ins] <- ss, ys, ys, yy first
  double ss = 0, ys = 0, yy = 0;
  // transposed, zero-padded copy of the finished coefficients for panel_gemm_kernel (NULL: launch_panel_gemm transposes)
  double *Ct = nullptr;
  int noutmax = 0;
};
__device__ __forceinline__ void gram_update_prologue(const GramUpd &G) {
  if (!G.on) return;
  const int k = threadIdx.x, mem = G.mem, ins = G.ins;
  if (k < mem) {
    double t0 = G.tmp[k], t1 = G.tmp[mem + k], t2 = G.tmp[2 * mem + k], t3 = G.tmp[3 * mem + k];
    if (G.self && k == ins) {
      t0 = G.ss; t1 = G.ys; t2 = G.ys; t3 = G.yy;
      G.tmp[k] = t0; G.tmp[mem + k] = t1; G.tmp[2 * mem + k] = t2; G.tmp[3 * mem + k] = t3;
    }
    G.SS[k * mem + ins] = t0;
    G.SS[ins * mem + k] = t0;
    G.YSf[k * mem + ins] = t1;
    G.YSf[ins * mem + k] = t2;
    G.YY[k * mem + ins] = t3;
    G.YY[ins * mem + k] = t3;
  }
  __syncthreads();
}
// ... and the transposition panel_gemm_kernel wants (transpose_coef_kernel) as an EPILOGUE: Ct[j][k] = Cm[k][j], rows padded
__device__ __forceinline__ void coef_transpose_epilogue(const GramUpd &G, const double *Cm, int nout, int nin) {
  if (!G.Ct) return;
  __syncthreads();
  for (int idx = threadIdx.x; idx < nin * G.noutmax; idx += blockDim.x) {
    const int j = idx / G.noutmax, k = idx % G.noutmax;
    G.Ct[idx] = k < nout ? Cm[(int64_t)k * nin + j] : 0.0;
  }
}

// place the dots of one pair: tmp[0..m) = S's_new, tmp[m..2m) = Y's_new, tmp[2m..3m) = S'y_new, tmp[3m..4m) = Y'y_new
__global__ void gram_update_kernel(double *__restrict__ SS, double *__restrict__ YSf, double *__restrict__ YY,
                                   const double *__restrict__ tmp, int mem, int ins) {
  const int k = threadIdx.x;
  if (k >= mem) return;
  SS[k * mem + ins] = tmp[k];
  SS[ins * mem + k] = tmp[k];
  YSf[k * mem + ins] = tmp[mem + k];       // y_k ' s_new
  YSf[ins * mem + k] = tmp[2 * mem + k];   // y_new ' s_k
  YY[k * mem + ins] = tmp[3 * mem + k];
  YY[ins * mem + k] = tmp[3 * mem + k];
}

// coefficients of a_k (active slots, oldest -> newest) on the basis [s_ord[0..r), b_ord[0..r)]:
//   Cm[k*(2r) + j] (j < r: on s_j; j >= r: on b_{j-r}).  b_j = y_j / sqrt(ys_j)  (src/lbfgs.jl:232)
// NB: `c += Z[r+l][k]` must precede the `as` reduction of the same l in the reference (:244 then :245 uses
// dot(a_l, s_k), independent of the running a_k), so the order inside the loop is immaterial.
__global__ void __launch_bounds__(64)
afwd_coef_kernel(const double *SS, const double *YSf, double *Cm, OrdArgs O, GramUpd G = GramUpd{}) {
  // one wave; lane j owns coefficient j of the 2r-vector (2r <= 64). Z[j][k] = <basis_j, s_k>.
  // (SS / YSf are NOT __restrict__: the prologue writes them)
  __shared__ double Z[2 * kMaxMemFwd][kMaxMemFwd];
  __shared__ double Cl[kMaxMemFwd][2 * kMaxMemFwd];
  gram_update_prologue(G);
  const int lane = threadIdx.x;
  const int r = O.na, w = 2 * r;
  fill_Z<true>(Z, SS, YSf, O);
  __syncthreads();
  auto wsum = [](double v) { return wave_allsum(v); };   // DPP tree (common.h), total in every lane
  for (int k = 0; k < r; ++k) {
    double c = (lane == k) ? 1.0 / O.gamma : 0.0;             // a_k = s_k / γ                 (:239)
    const double zk = lane < w ? Z[lane][k] : 0.0;
    // dot(a_l, s_k) does not involve the running a_k: four reductions are issued together (their DPP chains overlap —
    // a lone chain is ~250 cycles of latency, and there are r²/2 of them: 22 of the 30 us of this kernel at r = 20),
    // the updates are then applied in the reference's order. Bit-identical to the one-at-a-time loop.
    int l = 0;
    for (; l + 4 <= k; l += 4) {
      double cl[4], as[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) cl[u] = lane < w ? Cl[l + u][lane] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) as[u] = wsum(cl[u] * zk);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (lane == r + l + u) c += zk;                       // Z[r + l + u][k] IS this lane's zk: no LDS read in a branch
        c -= as[u] * cl[u];
      }
    }
    for (; l < k; ++l) {
      const double cl = lane < w ? Cl[l][lane] : 0.0;
      if (lane == r + l) c += zk;                             // += dot(b_l, s_k) b_l = Z[r + l][k], this lane's zk (:244)
      const double as = wsum(cl * zk);                        // dot(a_l, s_k)
      c -= as * cl;                                           // -= dot(a_l, s_k) a_l           (:245)
    }
    const double nn = wsum(c * zk);                           // dot(s_k, a_k)                  (:248)
    c *= 1.0 / sqrt(nn);
    if (lane < w) {
      Cl[k][lane] = c;
      Cm[(int64_t)k * w + lane] = c;
    }
    __syncthreads();
  }
  coef_transpose_epilogue(G, Cm, r, w);
}

constexpr int kGemmIn = 64, kGemmOut = 32;
template <typename T>
struct PanelGemmArgs {
  const T *in[kGemmIn];
  T *out[kGemmOut];
  int nin, nout;
  const double *C;  // nout x nin, row-major with row stride cstride (0: nin), as produced by the coefficient kernel
  int64_t cstride = 0;
  int accum = 0;    // start from the current contents of the outputs (input chunks of the big-memory path)
  // push!: up to two caller vectors copied into their panel slots by the same pass (they are inputs of it anyway)
  const T *cp_src[2] = {nullptr, nullptr};
  T *cp_dst[2] = {nullptr, nullptr};
  int ncp = 0;
};

// out_k[i] = sum_j C[k][j] * in_j[i].  j-outer streaming form: the nout accumulators of a lane's two rows
// live in registers, input columns are streamed 4 at a time (each read exactly once), the coefficients
// of one input column are wave-uniform and fetched with scalar loads from a transposed copy Ct[j][k].
// Products accumulate in f64.
template <typename T, int NOUTMAX>
__global__ void __launch_bounds__(kBlock)
panel_gemm_kernel(PanelGemmArgs<T> A, const double *__restrict__ Ct, int64_t nvec) {
  typedef T V __attribute__((ext_vector_type(2)));
  constexpr int UJ = 4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kBlock) {
    double acc[NOUTMAX][2];
#pragma unroll
    for (int k = 0; k < NOUTMAX; ++k) {
      acc[k][0] = acc[k][1] = 0.0;
      if (A.accum && k < A.nout) {
        const V o = *reinterpret_cast<const V *>(A.out[k] + i * 2);
        acc[k][0] = (double)o[0];
        acc[k][1] = (double)o[1];
      }
    }
    for (int j0 = 0; j0 < A.nin; j0 += UJ) {
      V in[UJ];
#pragma unroll
      for (int u = 0; u < UJ; ++u)
        if (j0 + u < A.nin) in[u] = __builtin_nontemporal_load(reinterpret_cast<const V *>(A.in[j0 + u] + i * 2));
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        if (j0 + u < A.nin) {
          const double x0 = (double)in[u][0], x1 = (double)in[u][1];
          const double *c = Ct + (int64_t)(j0 + u) * NOUTMAX;
#pragma unroll
          for (int k = 0; k < NOUTMAX; ++k) {
            const double ck = c[k];               // wave-uniform: scalar load (padded rows are zero)
            acc[k][0] = fma(ck, x0, acc[k][0]);
            acc[k][1] = fma(ck, x1, acc[k][1]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NOUTMAX; ++k) {
      if (k < A.nout) {
        V o;
        o[0] = (T)acc[k][0];
        o[1] = (T)acc[k][1];
        __builtin_nontemporal_store(o, reinterpret_cast<V *>(A.out[k] + i * 2));
      }
    }
    for (int c = 0; c < A.ncp; ++c)   // the lines were read a moment ago as inputs: this re-read is served by the cache
      __builtin_nontemporal_store(*reinterpret_cast<const V *>(A.cp_src[c] + i * 2), reinterpret_cast<V *>(A.cp_dst[c] + i * 2));
  }
}

// Ct[j][k] (row length NOUTMAX, zero padded) from C[k][j]
__global__ void transpose_coef_kernel(const double *__restrict__ C, double *__restrict__ Ct, int nout, int nin,
                                      int noutmax, int64_t cstride) {
  for (int idx = threadIdx.x; idx < nin * noutmax; idx += blockDim.x) {
    const int j = idx / noutmax, k = idx % noutmax;
    Ct[idx] = k < nout ? C[(int64_t)k * cstride + j] : 0.0;
  }
}

inline int gemm_noutmax(int nout) { return nout <= 8 ? 8 : (nout <= 16 ? 16 : (nout <= 24 ? 24 : 32)); }
inline double *gemm_ct(mxlo_ctx *ctx) { return ctx->scalars + 1024; }   // kGemmIn * kGemmOut = 2048 doubles inside the ctx scalar buffer

// ct_ready: the coefficient kernel already left the transposed copy in gemm_ct(ctx) (coef_transpose_epilogue)
template <typename T>
int32_t launch_panel_gemm(mxlo_ctx *ctx, PanelGemmArgs<T> &A, int64_t n, bool ct_ready = false) {
  if (n <= 0 || A.nout <= 0) return MXLO_OK;
  // panel columns are 16-byte aligned and padded (ld) to a whole number of 16-byte vectors: run pairs of rows
  const int64_t nvec = (n + 1) / 2;
  const int grid = grid_for(ctx, nvec, kBlock, 0);
  double *Ct = gemm_ct(ctx);
  auto go = [&]<int NOUTMAX>() -> int32_t {
    if (!ct_ready) {
      hipLaunchKernelGGL(transpose_coef_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, A.C, Ct, A.nout, A.nin, NOUTMAX,
                         A.cstride > 0 ? A.cstride : (int64_t)A.nin);
      MXLO_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((panel_gemm_kernel<T, NOUTMAX>), dim3(grid), dim3(kBlock), 0, ctx->stream, A, Ct, nvec);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  };
  if (A.nout <= 8) return go.template operator()<8>();
  if (A.nout <= 16) return go.template operator()<16>();
  if (A.nout <= 24) return go.template operator()<24>();
  return go.template operator()<32>();
}

// forward L-BFGS in compact form never touches the a_k panel: allocate it (n x mem, zeroed) on first need, so a
// handle that only ever sees push!/mul!/solve_shifted_system! holds three panels instead of four
inline int32_t alloc_A(mxlo_qn *h) {
  if (h->A) return MXLO_OK;
  const size_t bytes = (size_t)h->ld * h->mem * (h->dtype == MXLO_F64 ? 8 : 4);
  hipError_t e = hipMalloc(&h->A, bytes > 0 ? bytes : 16);
  MXLO_REQUIRE(e == hipSuccess, MXLO_ENOMEM, "a_k panel: %s", hipGetErrorString(e));
  MXLO_HIP(hipMemsetAsync(h->A, 0, bytes, h->ctx->stream));
  return MXLO_OK;
}

// materialise the a_k panel from the coefficients of the last compact push! (diag!, solve_shifted_system!,
// mxlo_qn_column and the reference-ordered push! read it)
template <typename T>
int32_t ensure_A(mxlo_qn *h) {
  if (h->kind != MXLO_QN_LBFGS_FWD) return MXLO_OK;
  MXLO_TRY(alloc_A(h));
  if (h->A_valid) return MXLO_OK;
  OrdArgs O;
  fill_ord(h, O, /*newest_first=*/false);
  PanelGemmArgs<T> G;
  G.nin = 2 * O.na;
  G.nout = O.na;
  G.C = h->dsc + h->lay.Cm;
  for (int j = 0; j < O.na; ++j) {
    G.in[j] = col<T>(h->S, h->ld, O.ord[j]);
    G.in[O.na + j] = col<T>(h->B, h->ld, O.ord[j]);
    G.out[j] = col<T>(h->A, h->ld, O.ord[j]);
  }
  MXLO_TRY(launch_panel_gemm<T>(h->ctx, G, h->n));
  h->A_valid = true;
  return MXLO_OK;
}

// L-SR1 rank-one terms in coefficient space (src/lsr1.jl:166-178): basis [y_ord[0..r), s_ord[0..r)],
//   a_k = y_k - s_k/γ - sum_{l<k} (a_l's_k / as_l) a_l ,  as_k = a_k's_k.  One wave, lane j owns coefficient j.
__global__ void __launch_bounds__(64)
asr1_coef_kernel(const double *SS, const double *YSf, double *Cm, double *as_out, OrdArgs O, GramUpd G = GramUpd{}) {
  __shared__ double Z[2 * kMaxMemFwd][kMaxMemFwd];      // Z[j][k] = <basis_j, s_k>
  __shared__ double Cl[kMaxMemFwd][2 * kMaxMemFwd];
  __shared__ double asl[kMaxMemFwd];
  gram_update_prologue(G);                               // (SS / YSf not __restrict__: the prologue writes them)
  const int lane = threadIdx.x;
  const int r = O.na, w = 2 * r;
  fill_Z<false>(Z, YSf, SS, O);
  __syncthreads();
  auto wsum = [](double v) { return wave_allsum(v); };   // DPP tree (common.h), total in every lane
  for (int k = 0; k < r; ++k) {
    double c = (lane == k) ? 1.0 : ((lane == r + k) ? -1.0 / O.gamma : 0.0);   // y_k - s_k/γ       (:169)
    const double zk = lane < w ? Z[lane][k] : 0.0;
    int l = 0;
    for (; l + 4 <= k; l += 4) {   // four independent reductions in flight, updates in the reference's order (see afwd_coef_kernel)
      double cl[4], as[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) cl[u] = lane < w ? Cl[l + u][lane] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) as[u] = wsum(cl[u] * zk) / asl[l + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) c -= as[u] * cl[u];
    }
    for (; l < k; ++l) {
      const double cl = lane < w ? Cl[l][lane] : 0.0;
      const double as = wsum(cl * zk) / asl[l];                                // dot(a_l,s_k)/as_l (:173)
      c -= as * cl;                                                            //                   (:174)
    }
    const double ask = wsum(c * zk);                                           // as_k = a_k's_k    (:177)
    if (lane < w) {
      Cl[k][lane] = c;
      Cm[(int64_t)k * w + lane] = c;
    }
    if (lane == 0) {
      asl[k] = ask;
      as_out[O.ord[k]] = ask;
    }
    __syncthreads();
  }
  coef_transpose_epilogue(G, Cm, r, w);
}

// Gram rows/columns of one stored pair (slot): S'[s y] in one dual-x pass over S, Y'[s y] in one over Y.
template <typename T>
int32_t gram_update_slot(mxlo_qn *h, int64_t slot) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n, mem = h->mem;
  double *tmp = h->dsc + h->lay.gtmp;
  const T *cols[kMaxCols];
  for (int k = 0; k < mem; ++k) {
    cols[k] = col<T>(h->S, h->ld, k);
    cols[mem + k] = col<T>(h->Y, h->ld, k);
  }
  constexpr int VECP = Vec16<T>::N;
  const int64_t npad = (n + VECP - 1) / VECP * VECP;   // panel columns are zero-padded to whole vectors (<= ld)
  const T *ss = col<T>(h->S, h->ld, slot), *yy = col<T>(h->Y, h->ld, slot);
  MXLO_TRY(panel_dots2<T>(ctx, cols, (int)mem, ss, yy, npad, tmp, tmp + 2 * mem));            // S's, S'y
  MXLO_TRY(panel_dots2<T>(ctx, cols + mem, (int)mem, ss, yy, npad, tmp + mem, tmp + 3 * mem)); // Y's, Y'y
  hipLaunchKernelGGL(gram_update_kernel, dim3(1), dim3(64), 0, ctx->stream, h->dsc + h->lay.SS,
                     h->dsc + h->lay.YSf, h->dsc + h->lay.YY, tmp, (int)mem, (int)slot);
  MXLO_LAUNCH_CHECK();
  return MXLO_OK;
}

// after reference-ordered pushes (which do not maintain them) the Gram matrices are rebuilt pair by pair
template <typename T>
int32_t gram_make_consistent(mxlo_qn *h, int64_t except_slot) {
  if (h->gram_ok) return MXLO_OK;
  for (int64_t k = 0; k < h->mem; ++k)
    if (h->ys[k] != 0 && k != except_slot) MXLO_TRY(gram_update_slot<T>(h, k));
  h->gram_ok = true;
  return MXLO_OK;
}

template <typename T>
int32_t fwd_rebuild_after_gram(mxlo_qn *h, int64_t ins, const double *gram_rows = nullptr);

// Gram-form rebuild of the forward panel after (s, y) was copied into slot `ins` and b[ins] was formed.
template <typename T>
int32_t fwd_rebuild_gram(mxlo_qn *h, int64_t ins) {
  MXLO_TRY(gram_make_consistent<T>(h, ins));
  MXLO_TRY(gram_update_slot<T>(h, ins));
  return fwd_rebuild_after_gram<T>(h, ins);
}

// ... the part after the Gram rows of slot `ins` are in place (shared with the one-pass push!)
// gram_rows (one-pass push!): [S's | Y's | S'y | Y'y] of the new pair, still to be placed in the Gram matrices — done by the
// coefficient kernel's prologue instead of a gram_update_kernel launch
template <typename T>
int32_t fwd_rebuild_after_gram(mxlo_qn *h, int64_t ins, const double *gram_rows) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n, mem = h->mem;
  OrdArgs O;
  fill_ord(h, O, /*newest_first=*/false);   // called before insert0 advances: the new pair must come last
  // oldest -> newest with the freshly written slot last: slots (ins+1 .. ins+mem) mod mem
  O.na = 0;
  for (int64_t i = 1; i <= mem; ++i) {
    const int64_t k = (ins + i) % mem;
    if (h->ys[k] != 0) O.ord[O.na++] = (int)k;
  }
  O.gamma = h->scaling_factor;  // (:239) divides unconditionally; γ == 1 without scaling
  const bool compact = h->push_mode == MXLO_PUSH_COMPACT;
  if (!compact) MXLO_TRY(alloc_A(h));   // (before the launches: its first call allocates and clears the panel)
  GramUpd U;
  if (gram_rows) {
    U.on = 1; U.mem = (int)mem; U.ins = (int)ins;
    U.SS = h->dsc + h->lay.SS; U.YSf = h->dsc + h->lay.YSf; U.YY = h->dsc + h->lay.YY;
    U.tmp = const_cast<double *>(gram_rows);
  }
  if (!compact && n > 0 && O.na > 0) {
    U.Ct = gemm_ct(ctx);
    U.noutmax = gemm_noutmax(O.na);
  }
  hipLaunchKernelGGL(afwd_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, h->dsc + h->lay.SS,
                     h->dsc + h->lay.YSf, h->dsc + h->lay.Cm, O, U);
  MXLO_LAUNCH_CHECK();
  if (compact) {  // a_k = [S B]·c_k stays implicit: the apply works on [S B] and Cm
    h->A_valid = false;
    return MXLO_OK;
  }
  PanelGemmArgs<T> A;
  A.nin = 2 * O.na;
  A.nout = O.na;
  A.C = h->dsc + h->lay.Cm;
  for (int j = 0; j < O.na; ++j) {
    A.in[j] = col<T>(h->S, h->ld, O.ord[j]);
    A.in[O.na + j] = col<T>(h->B, h->ld, O.ord[j]);
    A.out[j] = col<T>(h->A, h->ld, O.ord[j]);
  }
  h->A_valid = true;
  return launch_panel_gemm<T>(ctx, A, n, /*ct_ready=*/U.Ct != nullptr);
}

// push_common! — src/lbfgs.jl:210-255 (ys, yy already known on the host)
template <typename T>
int32_t lbfgs_push_common(mxlo_qn *h, const T *s, const T *y, double ys, double yy) {
  if (h->big) return lbfgs_push_common_big<T>(h, s, y, ys, yy);
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n, ins = h->insert0, mem = h->mem;
  T *si = col<T>(h->S, h->ld, ins), *yi = col<T>(h->Y, h->ld, ins);
  MXLO_HIP(hipMemcpyAsync(si, s, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));  // :220
  MXLO_HIP(hipMemcpyAsync(yi, y, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));  // :221
  h->ys[ins] = ys;                                                                       // :222
  h->age[ins] = ++h->pushes;
  if (h->scaling) h->scaling_factor = rT<T>(ys / yy);                                    // :225
  h->G_valid = false;
  double *dots = h->dsc + h->lay.dots;
  if (h->kind == MXLO_QN_LBFGS_INV) {
    // Gram columns for the two-pass apply: S'y_new, Y'y_new (x = y_new), Y's_new (x = s_new)
    const T *cols[kMaxCols];
    for (int k = 0; k < mem; ++k) cols[k] = col<T>(h->S, h->ld, k);
    MXLO_TRY(panel_dots<T>(ctx, cols, (int)mem, yi, n, h->dsc + h->lay.SY + ins * mem));
    for (int k = 0; k < mem; ++k) cols[k] = col<T>(h->Y, h->ld, k);
    // Y'y_new and Y's_new in ONE pass over Y (dual-x dots)
    constexpr int VECP = Vec16<T>::N;
    const int64_t npad = (n + VECP - 1) / VECP * VECP;
    MXLO_TRY(panel_dots2<T>(ctx, cols, (int)mem, yi, si, npad, h->dsc + h->lay.YY + ins * mem,
                            h->dsc + h->lay.YS + ins * mem));
    return MXLO_OK;
  }
  // forward: b[insert] = y ./ sqrt(ys) (:232), then rebuild every a[k] (:236-250)
  T *bi = col<T>(h->B, h->ld, ins);
  const T sq = sizeof(T) == 8 ? (T)std::sqrt(ys) : (T)sqrtf((float)ys);
  MXLO_TRY((launch_map<T, 1, false, false>(ctx, bi, yi, (const T *)nullptr, n, DivConstOp<T>{sq})));
  {  // norm_b[insert]^2 kept on the device (lazy: read by get_scalars)
    const T *cols[1] = {bi};
    MXLO_TRY(panel_dots<T>(ctx, cols, 1, bi, n, h->dsc + h->lay.misc + 16 + ins));
  }
  if (h->push_mode != MXLO_PUSH_REFORDER) return fwd_rebuild_gram<T>(h, ins);
  MXLO_TRY(ensure_A<T>(h));   // the reference-ordered rebuild reads the older a_l
  h->gram_ok = false;         // ... and does not maintain the Gram matrices / Cm
  double *coef = h->dsc + h->lay.coef;
  int older[kMaxMem];
  int nold = 0;
  for (int64_t i = 1; i <= mem; ++i) {
    const int64_t k = (ins + i) % mem;  // :237 (0-based)
    if (h->ys[k] == 0) continue;
    T *ak = col<T>(h->A, h->ld, k);
    const T *sk = col<T>(h->S, h->ld, k);
    CombineArgs<T> A;
    A.ncol = 2 * nold;
    A.nfirst = 0;
    A.use_gamma = 1;
    A.gamma = h->scaling_factor;  // :239 divides unconditionally (gamma == 1 without scaling)
    A.alpha = 1;
    A.beta = 0;
    A.coef = coef;
    if (nold > 0) {
      for (int j = 0; j < nold; ++j) {
        A.cols[2 * j] = col<T>(h->B, h->ld, older[j]);
        A.cols[2 * j + 1] = col<T>(h->A, h->ld, older[j]);
      }
      MXLO_TRY(panel_dots<T>(ctx, A.cols, 2 * nold, sk, n, dots));  // dot(b[l], s[k]), dot(a[l], s[k])
      hipLaunchKernelGGL(copy_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, dots, coef, 2 * nold,
                         (int)(sizeof(T) == 4));
      MXLO_LAUNCH_CHECK();
    }
    MXLO_TRY((launch_combine<T, CM_AFWD>(ctx, ak, sk, (const T *)nullptr, A, n, 0)));
    const T *cols[1] = {sk};
    MXLO_TRY(panel_dots<T>(ctx, cols, 1, ak, n, dots + 100));        // dot(s[k], a[k])
    MXLO_TRY((launch_map<T, 0, true, false>(ctx, ak, (const T *)nullptr, (const T *)nullptr, n,
                                            DivSqrtDevOp<T>{dots + 100, T(0)})));  // :248
    older[nold++] = (int)k;
  }
  return MXLO_OK;
}

// push!(op, s, y) in ONE streaming schedule (small-memory layout, aligned s / y, not the
// reference-ordered forward mode). The two-kernel schedule below moved, for the inverse operator at mem = m:
//   dots(s, y | y) 3 vectors + 2 device-to-device inserts 4 vectors + S'y over S (m + 1) + dual-x dots over Y (m + 2)
//   = 2m + 10 vector passes; here:
//   pass A over S (m - 1 columns + s, y; column `ins` IS s, held per lane) -> S'y_new AND y's, y'y for the decision
//   [accept / reject on the host, as the reference: src/lbfgs.jl:281-284]
//   pass B over Y (m - 1 columns + s, y) -> Y's_new, Y'y_new, storing s -> S[:,ins], y -> Y[:,ins] from registers
//   = 2m + 4 vector passes (m = 10: 9.6 GB instead of 12 GB at n = 5e7). Forward (Gram / compact modes): both panels
// give both dots (S's, S'y, Y's, Y'y = the four Gram rows of gram_update_slot), pass B also writes b = y ./ sqrt(ys)
// and |b|^2: 4m + 2*ceil(m/10)*2 + 3 - 2 passes instead of 4m + 15.
// Nothing of the operator's state is written before the decision (pass A's rows go to scratch), so a rejected pair
// leaves the operator untouched exactly like the reference.
inline int32_t push_schedule_agree(mxlo_qn *h, bool local_ok, bool *all_ok);

// The streaming push! schedules read the caller's s and y WHILE kernels store into S[:,ins], Y[:,ins], B[:,ins] and the
// a_k panel (the copy-based schedules copied the pair into its slots first, so any aliasing was harmless there). A pair
// that lives inside the operator's own storage — a view of a column handed out by mxlo_qn_column, the tmp vectors —
// would race: such calls take the copy-based schedule.
template <typename T>
inline bool pair_aliases_handle(const mxlo_qn *h, const T *s, const T *y) {
  const size_t vb = sizeof(T) * (size_t)h->n, pb = sizeof(T) * (size_t)h->ld * (size_t)h->mem, tb = sizeof(T) * (size_t)h->ld;
  auto hit = [](const void *p, size_t pbytes, const void *q, size_t qbytes) {
    const uintptr_t a = (uintptr_t)p, b = (uintptr_t)q;
    return p && q && pbytes && qbytes && a < b + qbytes && b < a + pbytes;
  };
  for (const T *v : {s, y}) {
    for (const void *panel : {h->S, h->Y, h->A, h->B})
      if (hit(v, vb, panel, pb)) return true;
    if (hit(v, vb, h->tmp, tb) || hit(v, vb, h->tmp2, tb)) return true;
  }
  return false;
}

template <typename T>
int32_t lbfgs_push_fused(mxlo_qn *h, const T *s, const T *y, int32_t *accepted) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n, ins = h->insert0, mem = h->mem;
  constexpr int VECP = Vec16<T>::N;
  const int64_t npad = (n + VECP - 1) / VECP * VECP;
  const bool inverse = h->kind == MXLO_QN_LBFGS_INV;
  double *misc = h->dsc + h->lay.misc, *scratch = h->dsc + h->lay.dots, *gt = inverse ? nullptr : h->dsc + h->lay.gtmp;
  const T *cols[kMaxCols];
  // columns per pass: 20 while >= 20 remain (s, y are re-read once per pass: wide passes re-read them less), then <= 10
  int c0s[kMaxMem / 10 + 2], ncs[kMaxMem / 10 + 2], nchunk = 0;
  for (int64_t c = 0; c < mem;) {
    const int w = (ctx->tune.push_wide && mem - c >= 20) ? 20 : (int)std::min<int64_t>(10, mem - c);
    c0s[nchunk] = (int)c;
    ncs[nchunk++] = w;
    c += w;
  }
  auto pass = [&](void *panel, int chunk, int slot_src, T *st1, T *st2, T *stb, double sq, double *o1, double *o2,
                  double *oxy, double *oyy, double *obb, const PushExtras *ex = nullptr) -> int32_t {
    const int c0 = c0s[chunk], nc = ncs[chunk];
    for (int c = 0; c < nc; ++c) cols[c] = col<T>(panel, h->ld, c0 + c);
    const int slot = (ins >= c0 && ins < c0 + nc) ? (int)(ins - c0) : -1;
    return panel_push_pass<T>(ctx, cols, nc, slot, slot_src, s, y, n, npad, st1, st2, stb, sq, o1 ? o1 + c0 : nullptr,
                              o2 ? o2 + c0 : nullptr, oxy, oyy, obb, nullptr, ex);
  };
  // ---- pass A, first chunk of S: its Gram dots + the decision scalars misc[0] = y's, misc[1] = y'y. Without an
  // all-reduce hook (the scalars are final as the pass's finalize launch writes them) that launch also posts them to the host.
  const bool post = !ctx->allreduce && ctx->tune.push_posted && h->pinned_dev;
  PushExtras exA;
  if (post) {
    exA.post = h->pinned_dev + kPostPairs;
    exA.post_seq = ++h->post_seq;
  }
  MXLO_TRY(pass(h->S, 0, 1, nullptr, nullptr, nullptr, 1.0, inverse ? nullptr : gt, inverse ? scratch : gt + 2 * mem,
                misc, misc + 1, nullptr, post ? &exA : nullptr));
  double hs[2];
  if (post) {
    MXLO_TRY(await_posted_pairs(h, misc, hs, 2, exA.post_seq));
  } else {
    MXLO_TRY(allreduce_hook(ctx, misc, 2));
    MXLO_TRY(read_scalars(h, misc, hs, 2));
  }
  const double ys = rT<T>(hs[0]), yy = rT<T>(hs[1]);
  if (ys <= (double)eps_of<T>()) {  // src/lbfgs.jl:281-284
    *accepted = 0;
    return MXLO_OK;
  }
  *accepted = 1;
  for (int ch = 1; ch < nchunk; ++ch)   // rest of S
    MXLO_TRY(pass(h->S, ch, 1, nullptr, nullptr, nullptr, 1.0, inverse ? nullptr : gt, inverse ? scratch : gt + 2 * mem,
                  nullptr, nullptr, nullptr));
  T *si = col<T>(h->S, h->ld, ins), *yi = col<T>(h->Y, h->ld, ins);
  h->ys[ins] = ys;                                                                       // :222
  h->age[ins] = ++h->pushes;
  if (h->scaling) h->scaling_factor = rT<T>(ys / yy);                                    // :225
  h->G_valid = false;
  if (inverse) {
    // S'y_new -> column `ins` of SY; Y's_new, Y'y_new straight into their rows; the inserts ride in the first Y pass
    MXLO_TRY(allreduce_hook(ctx, scratch, mem));
    PushExtras exY;                       // (the copy scratch -> SY[:, ins] rides in the first Y pass's finalize launch)
    exY.cp_src = scratch;
    exY.cp_dst = h->dsc + h->lay.SY + ins * mem;
    exY.cp_n = (int)mem;
    for (int ch = 0; ch < nchunk; ++ch)
      MXLO_TRY(pass(h->Y, ch, 2, ch == 0 ? si : nullptr, ch == 0 ? yi : nullptr, nullptr, 1.0,
                    h->dsc + h->lay.YS + ins * mem, h->dsc + h->lay.YY + ins * mem, nullptr, nullptr, nullptr,
                    ch == 0 ? &exY : nullptr));
    MXLO_TRY(allreduce_hook(ctx, h->dsc + h->lay.YS + ins * mem, mem));
    MXLO_TRY(allreduce_hook(ctx, h->dsc + h->lay.YY + ins * mem, mem));
  } else {
    // forward: b[insert] = y ./ sqrt(ys) (:232) and |b|^2 ride in the first Y pass with the two inserts
    T *bi = col<T>(h->B, h->ld, ins);
    const double sq = sizeof(T) == 8 ? std::sqrt(ys) : (double)sqrtf((float)ys);
    for (int ch = 0; ch < nchunk; ++ch)
      MXLO_TRY(pass(h->Y, ch, 2, ch == 0 ? si : nullptr, ch == 0 ? yi : nullptr, ch == 0 ? bi : nullptr, sq, gt + mem,
                    gt + 3 * mem, nullptr, nullptr, ch == 0 ? misc + 16 + ins : nullptr));
    MXLO_TRY(allreduce_hook(ctx, gt, 4 * mem));
    MXLO_TRY(allreduce_hook(ctx, misc + 16 + ins, 1));
    MXLO_TRY(fwd_rebuild_after_gram<T>(h, ins, gt));   // (places the Gram rows gt in the coefficient kernel's prologue)
  }
  h->insert0 = (h->insert0 + 1) % h->mem;  // :253
  ++h->generation;
  return sync_meta(h);
}

template <typename T>
int32_t lbfgs_push(mxlo_qn *h, const T *s, const T *y, int32_t *accepted) {
  mxlo_ctx *ctx = h->ctx;
  // (forward operator whose Gram matrices are stale after reference-ordered pushes: the two-kernel schedule below
  //  rebuilds them, and ONLY for an accepted pair — a rejected push! must leave every piece of state, gram_ok and
  //  the a_k coefficients included, exactly as it found it)
  if (ctx->tune.push_fused && !h->big &&
      !(h->kind == MXLO_QN_LBFGS_FWD && (h->push_mode == MXLO_PUSH_REFORDER || !h->gram_ok))) {
    // (row-sharded: an empty or differently aligned shard on ONE rank must not send the ranks down different schedules)
    bool all_ok = false;
    MXLO_TRY(push_schedule_agree(h, (ctx->allreduce || h->n >= 1) && ((((uintptr_t)s) | ((uintptr_t)y)) & 15u) == 0 &&
                                        !pair_aliases_handle<T>(h, s, y), &all_ok));
    if (all_ok) return lbfgs_push_fused<T>(h, s, y, accepted);
  }
  double *misc = h->dsc + h->lay.misc;
  const T *cols[2] = {s, y};
  MXLO_TRY(panel_dots<T>(ctx, cols, 2, y, h->n, misc));  // misc[0] = dot(y,s), misc[1] = dot(y,y)
  double hs[2];
  MXLO_TRY(read_scalars(h, misc, hs, 2));
  const double ys = rT<T>(hs[0]), yy = rT<T>(hs[1]);
  if (ys <= (double)eps_of<T>()) {  // src/lbfgs.jl:281-284
    *accepted = 0;
    return MXLO_OK;
  }
  *accepted = 1;
  MXLO_TRY(lbfgs_push_common<T>(h, s, y, ys, yy));
  h->insert0 = (h->insert0 + 1) % h->mem;  // :253
  ++h->generation;
  return sync_meta(h);
}

// Powell damping shared by both damped pushes: given ys and sBs decide theta (src/lbfgs.jl:307-314)
template <typename T>
bool powell_theta(const mxlo_qn *h, double ys, double sBs, T *theta) {
  const T s2 = (T)h->sigma2, s3 = (T)h->sigma3, ysT = (T)ys, sBsT = (T)sBs;
  if (ysT < ((T)1 - s2) * sBsT) {
    *theta = s2 * sBsT / (sBsT - ysT);
    return true;
  }
  if (ysT > ((T)1 + s3) * sBsT) {
    *theta = s3 * sBsT / (ysT - sBsT);
    return true;
  }
  return false;
}

template <typename T>
int32_t lbfgs_push_damped(mxlo_qn *h, const T *s, T *y_mut, const T *y_const, double alpha,
                          const T *g, T *Bs, int32_t *accepted) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n;
  const bool inverse = h->kind == MXLO_QN_LBFGS_INV;
  const T *y = inverse ? y_mut : y_const;
  if (inverse) {  // Bs .= -α .* g  (:341)
    MXLO_TRY((launch_map<T, 1, false, false>(ctx, Bs, g, (const T *)nullptr, n, NegScaleOp<T>{(T)(-alpha)})));
  } else {        // mul!(Bs, op, s, one(T), zero(T))  (:305)
    if (h->big) MXLO_TRY(fwd_mul_big<T>(h, Bs, s, 1.0, 0.0, 0, 0.0));
    else MXLO_TRY(fwd_mul<T>(h, Bs, s, 1.0, 0.0, 0));
  }
  double *misc = h->dsc + h->lay.misc;
  const T *cols[3] = {s, y, Bs};
  MXLO_TRY(panel_dots<T>(ctx, cols, 1, y, n, misc));          // dot(y, s)
  MXLO_TRY(panel_dots<T>(ctx, cols + 2, 1, s, n, misc + 1));  // dot(s, Bs)
  double hs[2];
  MXLO_TRY(read_scalars(h, misc, hs, 2));
  double ys = rT<T>(hs[0]);
  const double sBs = rT<T>(hs[1]);
  T theta;
  const T *yuse = y;
  if (powell_theta<T>(h, ys, sBs, &theta)) {
    T *ydst = inverse ? y_mut : (T *)h->tmp2;  // forward rebinds y (:316), inverse mutates it (:351)
    MXLO_TRY((launch_map<T, 2, false, false>(ctx, ydst, y, Bs, n, DampOp<T>{theta, (T)1 - theta})));
    ys = (double)((theta * (T)ys) + (((T)1 - theta) * (T)sBs));
    yuse = ydst;
  }
  const T *cy[1] = {yuse};
  MXLO_TRY(panel_dots<T>(ctx, cy, 1, yuse, n, misc + 2));     // dot(y, y) for scaling (:225)
  double yyh[1];
  MXLO_TRY(read_scalars(h, misc + 2, yyh, 1));
  *accepted = 1;  // the damped pushes have no ys <= eps rejection in the reference
  MXLO_TRY(lbfgs_push_common<T>(h, s, yuse, ys, rT<T>(yyh[0])));
  h->insert0 = (h->insert0 + 1) % h->mem;
  ++h->generation;
  return sync_meta(h);
}

// ---- L-SR1 push!, streaming schedule -------------------------------------------------------------
// The reference forms y - B s with an apply, takes five dots for its three tests, copies the pair in and rebuilds every
// a_k (src/lsr1.jl:124-181). In Gram form all of that needs: the S and Y panels once (their dots with s and y are the
// new pair's Gram rows AND, through the coefficients of the last rebuild, the a_k's of B s), the a_k panel once
// (r = y - B s is never stored: its three sums come out of the pass that forms it), and the rebuild A = [Y S] C, which
// reads the new pair straight from the caller's vectors and drops it into its slots on the way. (6m + 6) n elements
// instead of (7m + 24) n; nothing of the operator's state is written before the decision.

// dots[k] = a_k's from the Gram rows (gt: S's at [0, mem), Y's at [mem, 2 mem)) and the coefficients Cm of the last
// rebuild (a_k = sum_j Cm[k][j] y_ord[j] + Cm[k][r + j] s_ord[j]); coef[k] as lsr1_coef_kernel. One wave.
__global__ void __launch_bounds__(64)
lsr1_bs_coef_kernel(const double *__restrict__ gt, const double *__restrict__ Cm, double *__restrict__ dots,
                    double *__restrict__ coef, const double *__restrict__ as_, OrdArgs O, double alpha, int ct_f32) {
  const int lane = threadIdx.x, r = O.na, w = 2 * r;
  const double z = lane < r ? gt[O.mem + O.ord[lane]] : (lane < w ? gt[O.ord[lane - r]] : 0.0);
  for (int k = 0; k < r; ++k) {
    const double d = wave_allsum(lane < w ? Cm[(int64_t)k * w + lane] * z : 0.0);
    if (lane == 0) {
      dots[k] = d;
      lsr1_coef_body(k, dots, coef, as_, O, alpha, ct_f32);
    }
  }
}

__global__ void set_scalar_kernel(double *p, double v) { *p = v; }

// Every rank of a row-sharded operator must take the same push! schedule (the schedules issue different all-reduces):
// with a hook installed the ranks first agree that ALL of them are eligible (one summed scalar).
inline int32_t push_schedule_agree(mxlo_qn *h, bool local_ok, bool *all_ok) {
  mxlo_ctx *ctx = h->ctx;
  *all_ok = local_ok;
  if (!ctx->allreduce) return MXLO_OK;
  double *flag = h->dsc + h->lay.misc + 15;
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, ctx->stream, flag, local_ok ? 0.0 : 1.0);
  MXLO_LAUNCH_CHECK();
  MXLO_TRY(allreduce_hook(ctx, flag, 1));
  double v = 1.0;
  MXLO_TRY(read_scalars(h, flag, &v, 1));
  *all_ok = v == 0.0;
  return MXLO_OK;
}

// the three tests of src/lsr1.jl:131-149 on the reduced scalars
template <typename T>
bool lsr1_accepts(const mxlo_qn *h, double ys_d, double ss_d, double ymBs_s_d, double yy_d, double ymBs2_d, double t2_d) {
  const T ys = (T)ys_d, sNorm = (T)std::sqrt(rT<T>(ss_d)), yy = (T)yy_d;
  const T ymBs_s = (T)ymBs_s_d, ymBsNorm = (T)std::sqrt(rT<T>(ymBs2_d));
  const T eps = eps_of<T>();
  const bool well_defined = std::fabs((double)ymBs_s) >= (double)(eps + eps * ymBsNorm * sNorm);  // :131
  bool sufficient_curvature = true, scaling_condition = true;
  if (h->scaling) {
    const T yNorm = (T)std::sqrt((double)yy);                                              // :136
    sufficient_curvature = std::fabs((double)ys) >= (double)(eps * yNorm * sNorm);         // :137
    if (sufficient_curvature)
      scaling_condition = (T)std::sqrt(rT<T>(t2_d)) >= eps * yNorm * sNorm;                // :141
  }
  return well_defined && sufficient_curvature && scaling_condition;                        // :145-149
}

// true when one of the tests lsr1_accepts evaluates could come out differently under the rounding of another evaluation
// order. The streaming schedule forms r = y - B s with a_k's taken from the Gram data (Cm [Y's; S's]) instead of dots on
// the stored a_k panel; both are exact in exact arithmetic, but r's = y's - s'B s is a cancellation whose ABSOLUTE error is
// of the order eps (|y| + |B s|) |s| whatever the size of the result, while the thresholds of src/lsr1.jl:131-141 are of the
// order eps: a pair the memory (nearly) reproduces has an r that is rounding noise, and its accept / reject is decided by
// that noise — differently in the two schedules, after which insert and ys diverge for good. A test counts as decided
// only when its left-hand side clears the threshold by 2^8 times that error budget; everything else is re-evaluated by
// the apply-based schedule (nothing but scratch has been written at that point). The scalars are all-reduced, so every
// rank of a sharded operator takes the same route. Pairs an optimiser normally produces are far outside the band.
template <typename T>
bool lsr1_decision_is_marginal(const mxlo_qn *h, double ys_d, double ss_d, double ymBs_s_d, double yy_d, double ymBs2_d,
                               double t2_d) {
  const double K = 256.0, eps = (double)eps_of<T>();
  const double sN = std::sqrt(ss_d), yN = std::sqrt(yy_d), rN = std::sqrt(ymBs2_d);
  if (!(std::isfinite(sN) && std::isfinite(yN) && std::isfinite(rN) && std::isfinite(ymBs_s_d) && std::isfinite(ys_d)))
    return true;      // NaN / Inf: an operator blown up by ill-conditioned pairs can overflow in the Gram-form coefficients
                      // where the stored a_k's still give finite dots (or the reverse) — let the apply-based schedule decide
  const double budget = K * eps * ((2.0 * yN + rN) * sN + 1.0);   // |B s| <= |y| + |r|
  if (std::fabs(ymBs_s_d) <= (eps + eps * rN * sN) + budget) return true;                     // :131
  if (h->scaling) {
    const double thr = eps * yN * sN;
    if (std::fabs(ys_d) <= thr * (1.0 + K)) return true;                                      // :137
    const double t = std::sqrt(t2_d);                                                         // |y - s/sf|, sf = ys/yy
    if (!std::isfinite(t) || t <= thr + K * eps * (yN + sN * yy_d / std::fabs(ys_d))) return true;   // :141
  }
  return false;
}

template <typename T>
int32_t lsr1_push_copies(mxlo_qn *h, const T *s, const T *y, int32_t *accepted);

template <typename T>
int32_t lsr1_push_fused(mxlo_qn *h, const T *s, const T *y, int32_t *accepted) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n, mem = h->mem, ins = h->insert0;
  constexpr int VECP = Vec16<T>::N;
  const int64_t npad = (n + VECP - 1) / VECP * VECP;
  double *misc = h->dsc + h->lay.misc, *gt = h->dsc + h->lay.gtmp;
  double *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef, *as_ = h->dsc + h->lay.as_;
  const T *cols[kMaxCols];
  // ---- pass 1: the OLD S and Y panels against (s, y): gt = [S's | Y's | S'y | Y'y]; misc[0] = s'y, [1] = s's, [2] = y'y.
  // One launch when both panels fit one pass (2 mem <= 10, or = 20), else one launch per panel chunk.
  const int both = (int)(2 * mem);
  if (both <= 10 || both == 20) {
    for (int c = 0; c < mem; ++c) {
      cols[c] = col<T>(h->S, h->ld, c);
      cols[mem + c] = col<T>(h->Y, h->ld, c);
    }
    MXLO_TRY(panel_push_pass<T>(ctx, cols, both, -1, 0, s, y, n, npad, nullptr, nullptr, nullptr, 1.0, gt, gt + 2 * mem, misc,
                                misc + 2, nullptr, misc + 1));
  } else {
    for (int64_t c0 = 0; c0 < mem;) {
      const int nc = (ctx->tune.push_wide && mem - c0 >= 20) ? 20 : (int)std::min<int64_t>(10, mem - c0);
      for (int c = 0; c < nc; ++c) cols[c] = col<T>(h->S, h->ld, c0 + c);
      MXLO_TRY(panel_push_pass<T>(ctx, cols, nc, -1, 0, s, y, n, npad, nullptr, nullptr, nullptr, 1.0, gt + c0, gt + 2 * mem + c0,
                                  c0 == 0 ? misc : nullptr, c0 == 0 ? misc + 2 : nullptr, nullptr, c0 == 0 ? misc + 1 : nullptr));
      for (int c = 0; c < nc; ++c) cols[c] = col<T>(h->Y, h->ld, c0 + c);
      MXLO_TRY(panel_push_pass<T>(ctx, cols, nc, -1, 0, s, y, n, npad, nullptr, nullptr, nullptr, 1.0, gt + mem + c0, gt + 3 * mem + c0,
                                  nullptr, nullptr, nullptr));
      c0 += nc;
    }
  }
  MXLO_TRY(allreduce_hook(ctx, gt, 4 * mem));
  MXLO_TRY(allreduce_hook(ctx, misc, 3));
  // ---- pass 2: r = y - B s over the a_k panel, nothing stored; misc[3] = r's, [4] = |r|^2, [5] = |y - s/sf|^2
  const bool post = !ctx->allreduce && ctx->tune.push_posted && h->pinned_dev;
  unsigned long long post_seq = 0;
  OrdArgs O;
  fill_ord(h, O, false);
  CombineArgs<T> A;
  A.ncol = O.na;
  A.nfirst = 0;
  A.use_gamma = 1;
  A.gamma = h->scaling_factor;
  A.alpha = -1.0;
  A.beta = 1.0;
  A.coef = coef;
  A.n_valid = n;
  A.sfnum = misc;        // sf = ys / yy (:139); evaluated even when it will not be used, like the two-kernel schedule
  A.sfden = misc + 2;
  A.partials = ctx->partials;
  if (O.na > 0) {
    for (int i = 0; i < O.na; ++i) A.cols[i] = col<T>(h->A, h->ld, O.ord[i]);
    hipLaunchKernelGGL(lsr1_bs_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, gt, h->dsc + h->lay.Cm, dots, coef, as_, O,
                       -1.0, sizeof(T) == 4 ? 1 : 0);
    MXLO_LAUNCH_CHECK();
  }
  {
    const int64_t nvec = (n + VECP - 1) / VECP;
    int grid = grid_for(ctx, nvec, kBlock, ctx->tune.combine_blocks_per_cu);
    if (grid > kMaxRedBlocks) grid = kMaxRedBlocks;
    const bool nt = (int64_t)sizeof(T) * n * (A.ncol + 2) >= ctx->tune.nt_min_bytes;
    if (nt)
      hipLaunchKernelGGL((combine_kernel<T, T, T, CM_LSR1R, false, VECP, true>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         const_cast<T *>(y), s, (const T *)nullptr, A, nvec);
    else
      hipLaunchKernelGGL((combine_kernel<T, T, T, CM_LSR1R, false, VECP, false>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         const_cast<T *>(y), s, (const T *)nullptr, A, nvec);
    MXLO_LAUNCH_CHECK();
    if (post) {   // no hook: misc[0..2] and these three sums are final — this finalize launch posts all six to the host
      post_seq = ++h->post_seq;
      MXLO_TRY(finalize_and_post(ctx, 3, grid, misc + 3, h->pinned_dev + kPostPairs, post_seq, misc, 3));
    } else {
      MXLO_TRY(finalize_and_reduce(ctx, 3, grid, misc + 3));
      MXLO_TRY(allreduce_hook(ctx, misc + 3, 3));
    }
  }
  double hs[6];
  if (post) MXLO_TRY(await_posted_pairs(h, misc, hs, 6, post_seq));
  else MXLO_TRY(read_scalars(h, misc, hs, 6));   // the push's one device-to-host transfer (48 bytes)
  if (lsr1_decision_is_marginal<T>(h, hs[0], hs[1], hs[3], hs[2], hs[4], hs[5]))
    return lsr1_push_copies<T>(h, s, y, accepted);      // nothing but scratch has been written so far
  if (!lsr1_accepts<T>(h, hs[0], hs[1], hs[3], hs[2], hs[4], hs[5])) {
    *accepted = 0;
    return MXLO_OK;
  }
  *accepted = 1;
  const T ys = (T)hs[0], yy = (T)hs[2];
  h->ys[ins] = (double)ys;                                                                 // :153
  h->age[ins] = ++h->pushes;
  if (h->scaling) h->scaling_factor = (double)(ys / yy);                                   // :158
  h->insert0 = (ins + 1) % mem;                                                            // :163
  ++h->generation;
  // ---- Gram rows of the new pair, coefficients of every a_k, then ONE pass A = [Y S] C (:166-181) that also drops the
  // pair into its slots
  fill_ord(h, O, false);
  GramUpd U;                // own-pair entries + Gram rows in the coefficient kernel's prologue, the GEMM's transposed
  U.on = 1; U.self = 1;     // coefficients in its epilogue: one launch where there were four
  U.mem = (int)mem; U.ins = (int)ins;
  U.ss = hs[1]; U.ys = hs[0]; U.yy = hs[2];
  U.SS = h->dsc + h->lay.SS; U.YSf = h->dsc + h->lay.YSf; U.YY = h->dsc + h->lay.YY;
  U.tmp = gt;
  if (n > 0 && O.na > 0) {
    U.Ct = gemm_ct(ctx);
    U.noutmax = gemm_noutmax(O.na);
  }
  hipLaunchKernelGGL(asr1_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, h->dsc + h->lay.SS, h->dsc + h->lay.YSf,
                     h->dsc + h->lay.Cm, as_, O, U);
  MXLO_LAUNCH_CHECK();
  T *si = col<T>(h->S, h->ld, ins), *yi = col<T>(h->Y, h->ld, ins);
  const bool ride = n % 2 == 0;   // the rebuild reads pairs of rows: an odd n would read one element past s and y
  if (!ride) {
    MXLO_HIP(hipMemcpyAsync(si, s, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));
    MXLO_HIP(hipMemcpyAsync(yi, y, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));
  }
  PanelGemmArgs<T> G;
  G.nin = 2 * O.na;
  G.nout = O.na;
  G.C = h->dsc + h->lay.Cm;
  for (int j = 0; j < O.na; ++j) {
    const bool fresh = ride && O.ord[j] == ins;
    G.in[j] = fresh ? y : col<T>(h->Y, h->ld, O.ord[j]);
    G.in[O.na + j] = fresh ? s : col<T>(h->S, h->ld, O.ord[j]);
    G.out[j] = col<T>(h->A, h->ld, O.ord[j]);
  }
  if (ride) {
    G.ncp = 2;
    G.cp_src[0] = s; G.cp_dst[0] = si;
    G.cp_src[1] = y; G.cp_dst[1] = yi;
  }
  h->normA_valid = false;
  return launch_panel_gemm<T>(ctx, G, n, /*ct_ready=*/U.Ct != nullptr);
}

// push!(op::LSR1Operator, s, y) — src/lsr1.jl:119-184
template <typename T>
int32_t lsr1_push(mxlo_qn *h, const T *s, const T *y, int32_t *accepted) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n;
  if (ctx->tune.push_fused && !h->big && h->push_mode == MXLO_PUSH_GRAM && h->gram_ok) {
    bool all_ok = false;
    MXLO_TRY(push_schedule_agree(h, (ctx->allreduce || n >= 1) && ((((uintptr_t)s) | ((uintptr_t)y)) & 15u) == 0 &&
                                        !pair_aliases_handle<T>(h, s, y), &all_ok));
    if (all_ok) return lsr1_push_fused<T>(h, s, y, accepted);
  }
  return lsr1_push_copies<T>(h, s, y, accepted);
}

// the apply-based schedule: r = y - B s through lsr1_mul into tmp, decision dots on the stored vectors, the pair copied into
// its slots before the rebuild (also what a marginal decision or an aliased pair of the streaming schedule falls back to)
template <typename T>
int32_t lsr1_push_copies(mxlo_qn *h, const T *s, const T *y, int32_t *accepted) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n, mem = h->mem;
  T *ymBs = (T *)h->tmp;
  MXLO_HIP(hipMemcpyAsync(ymBs, y, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));  // :124
  if (h->big) MXLO_TRY(lsr1_mul_big<T>(h, ymBs, s, -1.0, 1.0, 0, 0.0));                    // :125
  else MXLO_TRY(lsr1_mul<T>(h, ymBs, s, -1.0, 1.0, 0));
  double *misc = h->dsc + h->lay.misc;
  const T *c3[3] = {y, s, ymBs};
  MXLO_TRY(panel_dots<T>(ctx, c3, 3, s, n, misc));          // y's, s's, ymBs's
  const T *c1[1] = {y};
  MXLO_TRY(panel_dots<T>(ctx, c1, 1, y, n, misc + 3));      // y'y
  const T *c2[1] = {ymBs};
  MXLO_TRY(panel_dots<T>(ctx, c2, 1, ymBs, n, misc + 4));   // ||ymBs||^2
  // scaling: ||y - s/sf|| with sf = ys/yy (:139-141) is evaluated speculatively on the device (sf is formed there from
  // the reduced scalars), so that ALL the push's control scalars come back in ONE device-to-host copy; the host then
  // takes the reference's decisions in the reference's order. (y = 0 gives sf = NaN on both paths: rejected.)
  if (h->scaling) {
    T *t2 = (T *)h->tmp2;
    MXLO_TRY((launch_map<T, 2, false, false>(ctx, t2, y, s, n, YmSOverDevOp<T>{misc, T(0)})));   // :140
    const T *ct[1] = {t2};
    MXLO_TRY(panel_dots<T>(ctx, ct, 1, t2, n, misc + 5));
  }
  double hs[6] = {0, 0, 0, 0, 0, 0};
  MXLO_TRY(read_scalars(h, misc, hs, h->scaling ? 6 : 5));
  if (!lsr1_accepts<T>(h, hs[0], hs[1], hs[2], hs[3], hs[4], hs[5])) {                     // :131-149
    *accepted = 0;
    return MXLO_OK;
  }
  const T ys = (T)hs[0], yy = (T)hs[3];
  *accepted = 1;
  const int64_t ins = h->insert0;
  MXLO_HIP(hipMemcpyAsync(col<T>(h->S, h->ld, ins), s, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));
  MXLO_HIP(hipMemcpyAsync(col<T>(h->Y, h->ld, ins), y, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));
  h->ys[ins] = (double)ys;                                                                 // :153
  h->age[ins] = ++h->pushes;
  if (h->scaling) h->scaling_factor = (double)(ys / yy);                                   // :158
  h->insert0 = (ins + 1) % mem;                                                            // :163
  ++h->generation;
  if (h->big) return lsr1_rebuild_big<T>(h, ins);
  // rebuild the rank-1 terms (:166-181)
  double *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef, *as_ = h->dsc + h->lay.as_;
  OrdArgs O;
  fill_ord(h, O, false);
  if (h->push_mode == MXLO_PUSH_GRAM) {
    // Gram form: 2m column reads of dots, an O(r^3) coefficient recurrence on one wave, ONE pass A = [Y S] * C
    MXLO_TRY(gram_make_consistent<T>(h, ins));
    MXLO_TRY(gram_update_slot<T>(h, ins));
    hipLaunchKernelGGL(asr1_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, h->dsc + h->lay.SS,
                       h->dsc + h->lay.YSf, h->dsc + h->lay.Cm, as_, O);
    MXLO_LAUNCH_CHECK();
    PanelGemmArgs<T> G;
    G.nin = 2 * O.na;
    G.nout = O.na;
    G.C = h->dsc + h->lay.Cm;
    for (int j = 0; j < O.na; ++j) {
      G.in[j] = col<T>(h->Y, h->ld, O.ord[j]);
      G.in[O.na + j] = col<T>(h->S, h->ld, O.ord[j]);
      G.out[j] = col<T>(h->A, h->ld, O.ord[j]);
    }
    h->normA_valid = false;
    return launch_panel_gemm<T>(ctx, G, n);
  }
  h->gram_ok = false;
  int nold = 0;
  for (int i = 0; i < O.na; ++i) {
    const int k = O.ord[i];
    T *ak = col<T>(h->A, h->ld, k);
    const T *sk = col<T>(h->S, h->ld, k), *yk = col<T>(h->Y, h->ld, k);
    CombineArgs<T> A;
    A.ncol = nold;
    A.nfirst = 0;
    A.use_gamma = 1;
    A.gamma = h->scaling_factor;  // :169 divides unconditionally
    A.alpha = 1;
    A.beta = 0;
    A.coef = coef;
    if (nold > 0) {
      OrdArgs Ol = O;
      Ol.na = nold;  // the first `nold` entries of ord are the older slots
      for (int j = 0; j < nold; ++j) A.cols[j] = col<T>(h->A, h->ld, O.ord[j]);
      MXLO_TRY(panel_dots<T>(ctx, A.cols, nold, sk, n, dots));                             // dot(a[l], s[k])
      hipLaunchKernelGGL(div_as_kernel, dim3(1), dim3(64), 0, ctx->stream, dots, coef, as_, Ol);  // :173
      MXLO_LAUNCH_CHECK();
    }
    MXLO_TRY((launch_combine<T, CM_ASR1>(ctx, ak, sk, yk, A, n, 0)));                      // :169,174
    const T *ca[1] = {ak};
    MXLO_TRY(panel_dots<T>(ctx, ca, 1, sk, n, as_ + k));                                   // :177 as[k]
    MXLO_TRY(panel_dots<T>(ctx, ca, 1, ak, n, h->dsc + h->lay.misc + 16 + k));             // ||a_k||^2 (opnorm)
    nold++;
  }
  return MXLO_OK;
}

// ---- solve_shifted_system! in coefficient space (SURVEY §8a equivalence (iii)) -------------------
// u_t, t = 0..nu-1: active slots oldest->newest, each contributing (a_k, b_k) with signs (+1, -1).
//   G = U'U (cached per state), g = U'b, then the reference's recursion (src/utilities.jl:226-246)
//   on coefficient vectors P[i] (p_i = sum_j P[i][j] u_j), finally x = x0*b + U cx.
__global__ void __launch_bounds__(64)
shifted_coef_kernel(const double *__restrict__ G, const double *__restrict__ gvec,
                    double *__restrict__ cx, int nu, double x0, int is_f32) {
  // one wave; lane j owns column j of every coefficient vector. P (nu x nu) and G live in LDS.
  extern __shared__ double sh[];  // P[nu*nu], Gs[nu*nu], v[nu], c2[nu]
  double *P = sh, *Gs = sh + nu * nu, *v = Gs + nu * nu, *c2 = v + nu;
  const int lane = threadIdx.x;
  for (int idx = lane; idx < nu * nu; idx += 64) {
    Gs[idx] = G[idx];
    P[idx] = 0.0;
  }
  __syncthreads();
  auto wsum = [](double val) { return wave_allsum(val); };   // DPP tree (common.h), total in every lane
  const double gl = lane < nu ? gvec[lane] : 0.0;
  double cxl = 0.0;
  for (int i = 0; i < nu; ++i) {
    const int sign_i = (i & 1) ? -1 : 1;
    // c2[t] = (sign_t v_t) * dot(p_t, u_i), t < i  — lane t does its own dot          (:235-237)
    if (lane < i) {
      double c0 = 0.0;
      for (int j = 0; j <= lane; ++j) c0 += P[lane * nu + j] * Gs[j * nu + i];
      c2[lane] = (((lane & 1) ? -1.0 : 1.0) * v[lane]) * c0;
    }
    __syncthreads();
    // p_i = x0 u_i + sum_t c2[t] p_t  — lane j does column j                          (:231,:238)
    double pij = (lane == i) ? x0 : 0.0;
    if (lane < i)
      for (int t = lane; t < i; ++t) pij += c2[t] * P[t * nu + lane];
    if (lane < nu) P[i * nu + lane] = pij;
    const double up = wsum(lane <= i ? pij * Gs[i * nu + lane] : 0.0);                 // dot(u_i, p_i)
    const double vi = rnd(1.0 / (1.0 - sign_i * up), is_f32);                          // (:242)
    if (lane == 0) v[i] = vi;
    const double pb = wsum(lane <= i ? pij * gl : 0.0);                                // p_i' b
    cxl += ((sign_i * vi) * pb) * pij;                                                 // (:243-244)
    __syncthreads();
  }
  if (lane < nu) cx[lane] = cxl;
  if (lane == 0) cx[kX0Slot] = x0;  // c0 slot read by CM_AXPYS
}

// Everything the recursion needs about U = [a_1 b_1 a_2 b_2 ...] follows from the Gram matrices push! maintains:
// u_t = [S B]·W[t] (W[t] = c_k for an a column, a unit vector for a b column), so G = U'U = W M W' with
// M = [S B]'[S B] assembled from S'S, Y'S, Y'Y and ys — no pass over n — and U'b = W ([S B]'b).
struct ShiftMap {
  int r, mem;
  int ord[kMaxMem];    // active slots, oldest -> newest (the order of Cm's rows and of the basis)
  int pos2[kMaxMem];   // position in `ord` of the i-th pair of the reference's solve order (utilities.jl:228)
  double ys[kMaxMem];  // by slot
};

__global__ void __launch_bounds__(kBlock)
shifted_gram_kernel(const double *__restrict__ SS, const double *__restrict__ YSf, const double *__restrict__ YY,
                    const double *__restrict__ Cm, ShiftMap mp, double *__restrict__ G, double *__restrict__ Wout) {
  extern __shared__ double shg[];  // M[w2*w2], W[nu*w2], Tm[nu*w2]
  const int r = mp.r, w2 = 2 * r, nu = 2 * r, mem = mp.mem, tid = threadIdx.x;
  double *M = shg, *W = M + w2 * w2, *Tm = W + nu * w2;
  for (int idx = tid; idx < w2 * w2; idx += kBlock) {
    const int p = idx / w2, q = idx % w2;
    const int sp = mp.ord[p < r ? p : p - r], sq = mp.ord[q < r ? q : q - r];
    double v;
    if (p < r && q < r) v = SS[sp * mem + sq];
    else if (p < r) v = YSf[sq * mem + sp] / sqrt(mp.ys[sq]);          // s_p'b_q, b = y / sqrt(ys) (lbfgs.jl:232)
    else if (q < r) v = YSf[sp * mem + sq] / sqrt(mp.ys[sp]);
    else v = YY[sp * mem + sq] / (sqrt(mp.ys[sp]) * sqrt(mp.ys[sq]));
    M[idx] = v;
  }
  for (int idx = tid; idx < nu * w2; idx += kBlock) {
    const int t = idx / w2, p = idx % w2, P = mp.pos2[t >> 1];
    W[idx] = (t & 1) ? (p == r + P ? 1.0 : 0.0) : Cm[(int64_t)P * w2 + p];   // even: a_k (sign +1), odd: b_k (-1)
  }
  __syncthreads();
  for (int idx = tid; idx < nu * w2; idx += kBlock) {
    const int t = idx / w2, q = idx % w2;
    double acc = 0.0;
    for (int p = 0; p < w2; ++p) acc = fma(W[t * w2 + p], M[p * w2 + q], acc);
    Tm[idx] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < nu * nu; idx += kBlock) {
    const int t = idx / nu, u = idx % nu;
    double acc = 0.0;
    for (int q = 0; q < w2; ++q) acc = fma(Tm[t * w2 + q], W[u * w2 + q], acc);
    G[idx] = acc;
  }
  for (int idx = tid; idx < nu * w2; idx += kBlock) Wout[idx] = W[idx];
}

// gv = W d  (U'b from the basis dots)
__global__ void shifted_gv_kernel(const double *__restrict__ W, const double *__restrict__ d, double *__restrict__ gv,
                                  int nu, int w2) {
  const int t = threadIdx.x;
  if (t >= nu) return;
  double acc = 0.0;
  for (int p = 0; p < w2; ++p) acc = fma(W[t * w2 + p], d[p], acc);
  gv[t] = acc;
}

// basis coefficients of U cx: coef = W' cx; the x0 factor of b travels in the slot CM_AXPYS reads it from
__global__ void shifted_back_kernel(const double *__restrict__ W, const double *__restrict__ cx,
                                    double *__restrict__ coef, int nu, int w2) {
  const int p = threadIdx.x;
  if (p < w2) {
    double acc = 0.0;
    for (int t = 0; t < nu; ++t) acc = fma(W[t * w2 + p], cx[t], acc);
    coef[p] = acc;
  }
  if (p == 0) coef[kX0Slot] = cx[kX0Slot];
}

template <typename T>
int32_t solve_shifted_t(mxlo_qn *h, T *x, const T *b, double sigma) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n;
  OrdArgs O;
  fill_ord(h, O, false);
  const int r = O.na, nu = 2 * r, w2 = 2 * r;
  if (!h->gram_ok) {  // reference-ordered pushes: bring S'S, Y'S, Y'Y and the a_k coefficients up to date
    MXLO_TRY(gram_make_consistent<T>(h, -1));
    O.gamma = h->scaling_factor;
    hipLaunchKernelGGL(afwd_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, h->dsc + h->lay.SS,
                       h->dsc + h->lay.YSf, h->dsc + h->lay.Cm, O);
    MXLO_LAUNCH_CHECK();
    h->G_valid = false;
  }
  // NOTE: the reference walks slots k = mod(insert + j - 1, mem) + 1, j = 1..mem (utilities.jl:228),
  // i.e. starting one slot AFTER the oldest-first order used by mul!; with a full memory that is
  // (oldest+1 ... newest, oldest). The recursion is order-dependent only through rounding; we follow it.
  ShiftMap mp;
  mp.r = r;
  mp.mem = (int)h->mem;
  for (int k = 0; k < h->mem; ++k) mp.ys[k] = h->ys[k];
  for (int i = 0; i < r; ++i) mp.ord[i] = O.ord[i];
  int na = 0;
  for (int64_t j = 1; j <= h->mem; ++j) {
    const int64_t k = (h->insert0 + j) % h->mem;
    if (h->ys[k] == 0) continue;
    int pos = 0;
    while (O.ord[pos] != (int)k) ++pos;
    mp.pos2[na++] = pos;
  }
  const T *bcols[kMaxCols];   // the basis [s_ord.., b_ord..]
  for (int i = 0; i < r; ++i) {
    bcols[i] = col<T>(h->S, h->ld, O.ord[i]);
    bcols[r + i] = col<T>(h->B, h->ld, O.ord[i]);
  }
  double *G = h->dsc + h->lay.G, *W = h->dsc + h->lay.Wm, *gv = h->dsc + h->lay.g, *cx = h->dsc + h->lay.cx,
         *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef;
  if (!ctx->lds_attr_set) {  // up to 3*64*64 doubles of dynamic LDS (> the 64 KiB default cap); the attribute is per device
    MXLO_HIP(hipFuncSetAttribute((const void *)shifted_coef_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 96 * 1024));
    MXLO_HIP(hipFuncSetAttribute((const void *)shifted_gram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 112 * 1024));
    ctx->lds_attr_set = true;
  }
  if (!h->G_valid && nu > 0) {
    hipLaunchKernelGGL(shifted_gram_kernel, dim3(1), dim3(kBlock), sizeof(double) * (w2 * w2 + 2 * nu * w2),
                       ctx->stream, h->dsc + h->lay.SS, h->dsc + h->lay.YSf, h->dsc + h->lay.YY,
                       h->dsc + h->lay.Cm, mp, G, W);
    MXLO_LAUNCH_CHECK();
    h->G_valid = true;
  }
  const double g_inv = rT<T>(1.0 / h->scaling_factor);                 // :219
  const double x0 = rT<T>(1.0 / (g_inv + sigma));                      // :220
  if (nu > 0) {
    MXLO_TRY(panel_dots<T>(ctx, bcols, w2, b, n, dots));                // [S B]'b : the only reduction pass over n
    hipLaunchKernelGGL(shifted_gv_kernel, dim3(1), dim3(64), 0, ctx->stream, W, dots, gv, nu, w2);
    MXLO_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(shifted_coef_kernel, dim3(1), dim3(64), sizeof(double) * (2 * nu * nu + 2 * nu + 2),
                     ctx->stream, G, gv, cx, nu, x0, (int)(sizeof(T) == 4));
  MXLO_LAUNCH_CHECK();
  hipLaunchKernelGGL(shifted_back_kernel, dim3(1), dim3(64), 0, ctx->stream, W, cx, coef, nu, w2);
  MXLO_LAUNCH_CHECK();
  CombineArgs<T> A;
  A.ncol = w2;
  A.nfirst = 0;
  A.use_gamma = 0;
  A.gamma = 1;
  A.alpha = 1;
  A.beta = 0;
  A.coef = coef;
  for (int i = 0; i < w2; ++i) A.cols[i] = bcols[i];
  return launch_combine<T, CM_AXPYS>(ctx, x, b, (const T *)nullptr, A, n, 0);
}

template <typename T>
int32_t diag_t(mxlo_qn *h, T *d) {
  OrdArgs O;
  fill_ord(h, O, false);
  CombineArgs<T> A;
  A.nfirst = 0;
  A.use_gamma = h->scaling;
  A.gamma = h->scaling_factor;
  A.alpha = 1;
  A.beta = 0;
  if (h->kind == MXLO_QN_LBFGS_FWD) {
    MXLO_TRY(ensure_A<T>(h));
    A.ncol = 2 * O.na;
    A.coef = h->dsc + h->lay.coef;
    for (int i = 0; i < O.na; ++i) {
      A.cols[2 * i] = col<T>(h->B, h->ld, O.ord[i]);
      A.cols[2 * i + 1] = col<T>(h->A, h->ld, O.ord[i]);
    }
    return launch_combine<T, CM_DIAG_FWD>(h->ctx, d, (const T *)nullptr, (const T *)nullptr, A, h->n, 0);
  }
  // L-SR1: coef[i] = as[ord[i]]
  A.ncol = O.na;
  double *coef = h->dsc + h->lay.coef;
  A.coef = coef;
  for (int i = 0; i < O.na; ++i) A.cols[i] = col<T>(h->A, h->ld, O.ord[i]);
  if (O.na > 0) {   // one tiny launch instead of na device-to-device copies
    hipLaunchKernelGGL(gather_as_kernel, dim3(1), dim3(64), 0, h->ctx->stream, h->dsc + h->lay.as_, coef, O);
    MXLO_LAUNCH_CHECK();
  }
  return launch_combine<T, CM_DIAG_SR1>(h->ctx, d, (const T *)nullptr, (const T *)nullptr, A, h->n, 0);
}

#include "qn_big.h"

template <typename T>
int32_t panel_gemm_big(mxlo_ctx *ctx, const std::vector<const T *> &in, const std::vector<T *> &out, const double *C,
                       int64_t n) {
  const int nin = (int)in.size(), nout = (int)out.size();
  for (int o0 = 0; o0 < nout; o0 += kGemmOut) {
    const int no = std::min(kGemmOut, nout - o0);
    for (int i0 = 0; i0 < nin; i0 += kGemmIn) {
      const int ni = std::min(kGemmIn, nin - i0);
      PanelGemmArgs<T> G;
      G.nin = ni;
      G.nout = no;
      G.C = C + (int64_t)o0 * nin + i0;
      G.cstride = nin;
      G.accum = i0 > 0;
      for (int j = 0; j < ni; ++j) G.in[j] = in[i0 + j];
      for (int k = 0; k < no; ++k) G.out[k] = out[o0 + k];
      MXLO_TRY(launch_panel_gemm<T>(ctx, G, n));
    }
  }
  return MXLO_OK;
}

}  // namespace

// ---- live-handle registry: lets a captured hipGraph check that the quasi-Newton state it baked in (slot order,
// γ, which kernels run) is still current — common.h: qn_generation / graph staleness check in api_ctx.hip.
#include <mutex>
#include <unordered_set>
static std::mutex g_qn_mu;
static std::unordered_set<const mxlo_qn *> g_qn_live;
static int64_t g_qn_next_id = 1;   // process-unique handle id (under g_qn_mu)
static void qn_live_add(mxlo_qn *h) {
  std::lock_guard<std::mutex> lk(g_qn_mu);
  // The generation starts at (unique id << 32): a handle allocated later at the SAME address as a destroyed one can
  // never present the generation a graph captured from its predecessor (the staleness check compares pointer +
  // generation; state changes add 1, 2^32 of them per handle would be needed to reach the next id).
  h->generation = g_qn_next_id++ << 32;
  g_qn_live.insert(h);
}
static void qn_live_remove(const mxlo_qn *h) {
  std::lock_guard<std::mutex> lk(g_qn_mu);
  g_qn_live.erase(h);
}
namespace mxlo {
bool qn_generation(const mxlo_qn *h, int64_t *gen) {
  std::lock_guard<std::mutex> lk(g_qn_mu);
  if (!g_qn_live.count(h)) return false;
  *gen = h->generation;
  return true;
}
static void note_capture(mxlo_qn *h) {
  if (h->ctx->capturing) h->ctx->captured_qn.emplace_back(h, h->generation);
}
}  // namespace mxlo

// ==================================================================================== C ABI
MXLO_API int32_t mxlo_qn_create(mxlo_ctx *ctx, int32_t kind, int32_t dtype, int64_t n, int64_t mem,
                                int32_t scaling, int32_t damped, double sigma2, double sigma3,
                                mxlo_qn **out) {
  MXLO_REQUIRE(ctx && out, MXLO_EINVAL, "mxlo_qn_create: NULL argument");
  MXLO_DEVICE_GUARD(ctx);
  MXLO_REQUIRE(kind >= MXLO_QN_LBFGS_INV && kind <= MXLO_QN_LSR1, MXLO_EINVAL, "bad kind %d", kind);
  MXLO_REQUIRE(dtype == MXLO_F64 || dtype == MXLO_F32, MXLO_EINVAL, "bad dtype %d", dtype);
  MXLO_REQUIRE(n >= 0, MXLO_ESHAPE, "n < 0");
  if (mem < 1) mem = 1;  // max(mem, 1), src/lbfgs.jl:37
  // any mem, like the reference (src/lbfgs.jl:26-35): up to 64 (inverse) / 32 (forward, L-SR1) slots run the by-value
  // single-wave path, larger memories the device-resident path of qn_big.h
  const int mem_cap = kind == MXLO_QN_LBFGS_INV ? kMaxMem : kMaxMemFwd;
  MXLO_REQUIRE(mem <= kBigMaxMem, MXLO_EINVAL, "mem = %lld exceeds %d (the m x m Gram matrices alone would take %lld MB)",
               (long long)mem, kBigMaxMem, (long long)(13 * mem * mem * 8 >> 20));
  mxlo_qn *h = new mxlo_qn();
  h->ctx = ctx;
  h->kind = kind;
  h->dtype = dtype;
  h->n = n;
  h->mem = mem;
  const int64_t es = dtype == MXLO_F64 ? 8 : 4;
  const int64_t vec = 16 / es;
  h->push_mode = kind == MXLO_QN_LBFGS_FWD ? MXLO_PUSH_COMPACT : MXLO_PUSH_GRAM;
  h->big = mem > mem_cap;
  h->ld = ((n + vec - 1) / vec) * vec;
  if (h->ld == 0) h->ld = vec;
  h->scaling = scaling != 0;
  h->damped = damped != 0;
  h->sigma2 = sigma2;
  h->sigma3 = sigma3;
  h->mode = ctx->tune.lbfgs_inv_mode;
  h->ys.assign(mem, 0.0);
  h->age.assign(mem, 0);
  h->lay = DscLayout(mem, h->big);
  const size_t pbytes = (size_t)h->ld * mem * es;
  hipError_t e = hipSuccess;
  auto alloc = [&](void **p, size_t bytes) {
    if (e == hipSuccess) {
      e = hipMalloc(p, bytes);
      if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, ctx->stream);
    }
  };
  alloc(&h->S, pbytes);
  alloc(&h->Y, pbytes);
  if (kind == MXLO_QN_LSR1) alloc(&h->A, pbytes);   // forward L-BFGS: the a_k panel is allocated on first use (alloc_A)
  if (kind == MXLO_QN_LBFGS_FWD) alloc(&h->B, pbytes);
  alloc(&h->tmp, (size_t)h->ld * es);
  alloc(&h->tmp2, (size_t)h->ld * es);
  alloc((void **)&h->dsc, sizeof(double) * h->lay.total);
  if (h->big) alloc(&h->meta, meta_bytes(mem));
  if (e == hipSuccess && !getenv("MXLO_NO_PINNED_READBACK")) {   // optional: without it the decision scalars are copied to the
    void *hp = nullptr;                                          // caller's stack (pageable: slower; the env var is for A/B timing)
    void *dp = nullptr;
    if (hipHostMalloc(&hp, kPinnedTotal * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      memset(hp, 0, kPinnedTotal * sizeof(double));
      h->pinned = (double *)hp;
      if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) h->pinned_dev = (double *)dp;
      else (void)hipGetLastError();
    } else {
      (void)hipGetLastError();
    }
  }
  if (e != hipSuccess) {
    set_error("mxlo_qn_create: device allocation failed: %s", hipGetErrorString(e));
    mxlo_qn_destroy(h);
    return MXLO_ENOMEM;
  }
  if (h->big) {
    const int32_t st = sync_meta(h);
    if (st != MXLO_OK) {
      mxlo_qn_destroy(h);
      return st;
    }
  }
  qn_live_add(h);
  *out = h;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_qn_destroy(mxlo_qn *h) {
  if (!h) return MXLO_OK;
  MXLO_DEVICE_GUARD(h->ctx);
  qn_live_remove(h);
  (void)hipStreamSynchronize(h->ctx->stream);
  for (void *p : {h->S, h->Y, h->A, h->B, h->tmp, h->tmp2, (void *)h->dsc, h->meta})
    if (p) (void)hipFree(p);
  if (h->pinned) (void)hipHostFree(h->pinned);
  delete h;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_qn_set_mode(mxlo_qn *h, int32_t mode) {
  MXLO_REQUIRE(h, MXLO_EINVAL, "handle is NULL");
  MXLO_REQUIRE(mode == MXLO_INV_TWOPASS || mode == MXLO_INV_REFORDER, MXLO_EINVAL, "bad mode");
  if (h->mode != mode) ++h->generation;
  h->mode = mode;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_qn_set_push_mode(mxlo_qn *h, int32_t mode) {
  MXLO_REQUIRE(h, MXLO_EINVAL, "handle is NULL");
  MXLO_REQUIRE(mode == MXLO_PUSH_GRAM || mode == MXLO_PUSH_REFORDER || mode == MXLO_PUSH_COMPACT, MXLO_EINVAL,
               "bad push mode");
  if (mode == MXLO_PUSH_COMPACT && h->kind != MXLO_QN_LBFGS_FWD) mode = MXLO_PUSH_GRAM;   // forward L-BFGS only
  if (h->big) {   // device-resident path: forward L-BFGS stays compact, L-SR1 Gram-form; no reference-ordered rebuild
    MXLO_REQUIRE(mode != MXLO_PUSH_REFORDER, MXLO_ESTATE,
                 "the reference-ordered push! rebuild is a validation mode limited to mem <= %d", kMaxMemFwd);
    mode = h->kind == MXLO_QN_LBFGS_FWD ? MXLO_PUSH_COMPACT : MXLO_PUSH_GRAM;
  }
  if (h->push_mode != mode) ++h->generation;
  h->push_mode = mode;
  return MXLO_OK;
}

MXLO_API int32_t mxlo_qn_mul(mxlo_qn *h, void *res, const void *x, double alpha, double beta,
                             int32_t flags) {
  MXLO_REQUIRE(h && (h->n == 0 || (res && x)), MXLO_EINVAL, "mxlo_qn_mul: NULL argument");
  MXLO_DEVICE_GUARD(h->ctx);
  note_capture(h);
  if (h->dtype == MXLO_F64) return qn_mul_t<double>(h, (double *)res, (const double *)x, alpha, beta, flags);
  return qn_mul_t<float>(h, (float *)res, (const float *)x, alpha, beta, flags);
}

MXLO_API int32_t mxlo_qn_mul_shifted(mxlo_qn *h, void *res, const void *x, double alpha, double beta,
                                     double sigma, int32_t flags) {
  MXLO_REQUIRE(h && (h->n == 0 || (res && x)), MXLO_EINVAL, "mxlo_qn_mul_shifted: NULL argument");
  MXLO_DEVICE_GUARD(h->ctx);
  note_capture(h);
  // shifted_prod! (src/shifted_operators.jl:16-25): mul!(y, H, x, α, β); iszero(σ) || iszero(α) || axpy!(α*σ, x, y).
  // α*σ is formed in the callers' types (σ is a T; α a T or a Float64), then axpy! converts it to T.
  double c = 0.0;
  if (sigma != 0 && alpha != 0) {
    if (h->dtype == MXLO_F64) c = alpha * sigma;
    else if (flags & MXLO_ALPHA_F64) c = (double)(float)(alpha * (double)(float)sigma);
    else c = (double)((float)alpha * (float)sigma);
  }
  if (h->dtype == MXLO_F64)
    return qn_mul_t<double>(h, (double *)res, (const double *)x, alpha, beta, flags, c);
  return qn_mul_t<float>(h, (float *)res, (const float *)x, alpha, beta, flags, c);
}

MXLO_API int32_t mxlo_qn_push(mxlo_qn *h, const void *s, const void *y, int32_t *accepted) {
  MXLO_REQUIRE(h && accepted && (h->n == 0 || (s && y)), MXLO_EINVAL, "mxlo_qn_push: NULL argument");   // an empty row shard has no vectors
  MXLO_DEVICE_GUARD(h->ctx);
  if (h->kind == MXLO_QN_LSR1) {
    if (h->dtype == MXLO_F64) return lsr1_push<double>(h, (const double *)s, (const double *)y, accepted);
    return lsr1_push<float>(h, (const float *)s, (const float *)y, accepted);
  }
  MXLO_REQUIRE(!h->damped, MXLO_ESTATE,
               "damped operator: use mxlo_qn_push_damped_fwd / mxlo_qn_push_damped_inv");
  if (h->dtype == MXLO_F64) return lbfgs_push<double>(h, (const double *)s, (const double *)y, accepted);
  return lbfgs_push<float>(h, (const float *)s, (const float *)y, accepted);
}

MXLO_API int32_t mxlo_qn_push_damped_fwd(mxlo_qn *h, const void *s, const void *y, void *Bs,
                                         int32_t *accepted) {
  MXLO_REQUIRE(h && accepted && (h->n == 0 || (s && y && Bs)), MXLO_EINVAL, "NULL argument");
  MXLO_DEVICE_GUARD(h->ctx);
  MXLO_REQUIRE(h->damped, MXLO_ESTATE, "This push! should be used for damped operators");
  MXLO_REQUIRE(h->kind == MXLO_QN_LBFGS_FWD, MXLO_ESTATE,
               "This function be used for forward operators. Use push!(op, s, y, α, g, Bs) instead.");
  if (h->dtype == MXLO_F64)
    return lbfgs_push_damped<double>(h, (const double *)s, nullptr, (const double *)y, 0.0, nullptr,
                                     (double *)Bs, accepted);
  return lbfgs_push_damped<float>(h, (const float *)s, nullptr, (const float *)y, 0.0, nullptr,
                                  (float *)Bs, accepted);
}

MXLO_API int32_t mxlo_qn_push_damped_inv(mxlo_qn *h, const void *s, void *y, double alpha,
                                         const void *g, void *Bs, int32_t *accepted) {
  MXLO_REQUIRE(h && accepted && (h->n == 0 || (s && y && g && Bs)), MXLO_EINVAL, "NULL argument");
  MXLO_DEVICE_GUARD(h->ctx);
  MXLO_REQUIRE(h->damped, MXLO_ESTATE, "This push! should be used for damped operators");
  MXLO_REQUIRE(h->kind == MXLO_QN_LBFGS_INV, MXLO_ESTATE,
               "This function be used for inverse operators. Use push!(op, s, y, Bs) instead.");
  if (h->dtype == MXLO_F64)
    return lbfgs_push_damped<double>(h, (const double *)s, (double *)y, nullptr, alpha,
                                     (const double *)g, (double *)Bs, accepted);
  return lbfgs_push_damped<float>(h, (const float *)s, (float *)y, nullptr, alpha, (const float *)g,
                                  (float *)Bs, accepted);
}

MXLO_API int32_t mxlo_qn_solve_shifted(mxlo_qn *h, void *x, const void *b, double sigma) {
  MXLO_REQUIRE(h && (h->n == 0 || (x && b)), MXLO_EINVAL, "NULL argument");
  MXLO_DEVICE_GUARD(h->ctx);
  note_capture(h);
  MXLO_REQUIRE(h->kind == MXLO_QN_LBFGS_FWD, MXLO_ESTATE,
               "solve_shifted_system! is defined for forward L-BFGS operators");
  MXLO_REQUIRE(!(sigma < 0), MXLO_EDOMAIN, "σ must be nonnegative");
  if (h->big) {
    if (h->dtype == MXLO_F64) return solve_shifted_big<double>(h, (double *)x, (const double *)b, sigma);
    return solve_shifted_big<float>(h, (float *)x, (const float *)b, (double)(float)sigma);
  }
  if (h->dtype == MXLO_F64) return solve_shifted_t<double>(h, (double *)x, (const double *)b, sigma);
  return solve_shifted_t<float>(h, (float *)x, (const float *)b, (double)(float)sigma);
}

MXLO_API int32_t mxlo_qn_diag(mxlo_qn *h, void *d) {
  MXLO_REQUIRE(h && (h->n == 0 || d), MXLO_EINVAL, "NULL argument");
  MXLO_DEVICE_GUARD(h->ctx);
  note_capture(h);
  MXLO_REQUIRE(h->kind != MXLO_QN_LBFGS_INV, MXLO_ESTATE,
               "only the diagonal of a forward L-BFGS approximation is available");
  if (h->big) {
    if (h->dtype == MXLO_F64) return diag_big<double>(h, (double *)d);
    return diag_big<float>(h, (float *)d);
  }
  if (h->dtype == MXLO_F64) return diag_t<double>(h, (double *)d);
  return diag_t<float>(h, (float *)d);
}

MXLO_API int32_t mxlo_qn_reset(mxlo_qn *h) {
  MXLO_REQUIRE(h, MXLO_EINVAL, "handle is NULL");
  MXLO_DEVICE_GUARD(h->ctx);
  const size_t es = h->dtype == MXLO_F64 ? 8 : 4;
  const size_t pbytes = (size_t)h->ld * h->mem * es;
  for (void *p : {h->S, h->Y, h->A, h->B})
    if (p) MXLO_HIP(hipMemsetAsync(p, 0, pbytes, h->ctx->stream));
  MXLO_HIP(hipMemsetAsync(h->dsc, 0, sizeof(double) * h->lay.total, h->ctx->stream));
  h->ys.assign(h->mem, 0.0);
  h->age.assign(h->mem, 0);
  h->pushes = 0;
  h->scaling_factor = 1.0;
  h->insert0 = 0;
  h->G_valid = false;
  h->normA_valid = true;
  h->A_valid = true;
  h->gram_ok = true;
  ++h->generation;
  return sync_meta(h);
}

MXLO_API int32_t mxlo_qn_get_scalars(mxlo_qn *h, double scalars[5], double *ys, double *aux) {
  MXLO_REQUIRE(h && scalars, MXLO_EINVAL, "NULL argument");
  MXLO_DEVICE_GUARD(h->ctx);
  const int64_t mem = h->mem;
  std::vector<double> dev(mem, 0.0), nrm(mem, 0.0);
  double bound = 1.0;
  if (h->kind == MXLO_QN_LBFGS_INV) {
    MXLO_TRY(read_scalars(h, h->dsc + h->lay.alpha, dev.data(), (int)mem));
    if (h->scaling && h->scaling_factor != 0) bound = 1.0 / h->scaling_factor;
  } else if (h->kind == MXLO_QN_LBFGS_FWD) {
    // ‖B‖ <= ‖B0‖ + sum ‖b_i‖²  (src/lbfgs.jl:11, :224-234): norm_b² kept in misc[16+k]
    MXLO_TRY(read_scalars(h, h->dsc + h->lay.misc + 16, nrm.data(), (int)mem));
    bound = (h->scaling && h->scaling_factor != 0) ? 1.0 / h->scaling_factor : 1.0;
    for (int64_t k = 0; k < mem; ++k) {
      dev[k] = std::sqrt(nrm[k]);
      if (h->ys[k] != 0) bound += nrm[k];
    }
  } else {
    if (!h->normA_valid) {  // ||a_k||^2 for the opnorm bound, computed on demand after Gram-form pushes
      for (int64_t k = 0; k < mem; ++k) {
        if (h->ys[k] == 0) continue;
        if (h->dtype == MXLO_F64) {
          const double *c[1] = {col<double>(h->A, h->ld, k)};
          MXLO_TRY(panel_dots<double>(h->ctx, c, 1, c[0], h->n, h->dsc + h->lay.misc + 16 + k));
        } else {
          const float *c[1] = {col<float>(h->A, h->ld, k)};
          MXLO_TRY(panel_dots<float>(h->ctx, c, 1, c[0], h->n, h->dsc + h->lay.misc + 16 + k));
        }
      }
      h->normA_valid = true;
    }
    MXLO_TRY(read_scalars(h, h->dsc + h->lay.as_, dev.data(), (int)mem));
    MXLO_TRY(read_scalars(h, h->dsc + h->lay.misc + 16, nrm.data(), (int)mem));
    bound = 1.0;                                                          // src/lsr1.jl:156
    if (h->scaling && h->scaling_factor != 0) bound = 1.0 / std::fabs(h->scaling_factor);  // :159
    for (int64_t k = 0; k < mem; ++k)
      if (h->ys[k] != 0 && dev[k] != 0) bound += nrm[k] / std::fabs(dev[k]);             // :179
  }
  scalars[0] = (double)(h->insert0 + 1);
  scalars[1] = h->scaling_factor;
  scalars[2] = bound;
  scalars[3] = (double)mem;
  scalars[4] = (double)h->n;
  if (ys)
    for (int64_t k = 0; k < mem; ++k) ys[k] = h->ys[k];
  if (aux)
    for (int64_t k = 0; k < mem; ++k) aux[k] = dev[k];
  return MXLO_OK;
}

MXLO_API int32_t mxlo_qn_column(mxlo_qn *h, int32_t which, int64_t k, void **out) {
  MXLO_REQUIRE(h && out, MXLO_EINVAL, "NULL argument");
  MXLO_DEVICE_GUARD(h->ctx);
  MXLO_REQUIRE(k >= 0 && k < h->mem, MXLO_EINVAL, "slot out of range");
  void *p = which == 0 ? h->S : which == 1 ? h->Y : which == 2 ? h->A : which == 3 ? h->B : nullptr;
  MXLO_REQUIRE(p, MXLO_ESTATE, "panel %d not allocated for this operator kind", which);
  if (which == 2) {  // a compact push! left the a_k implicit
    if (h->big) {
      if (h->dtype == MXLO_F64) MXLO_TRY(ensure_A_big<double>(h));
      else MXLO_TRY(ensure_A_big<float>(h));
    } else if (h->dtype == MXLO_F64) MXLO_TRY(ensure_A<double>(h));
    else MXLO_TRY(ensure_A<float>(h));
  }
  *out = (char *)p + (size_t)k * h->ld * (h->dtype == MXLO_F64 ? 8 : 4);
  return MXLO_OK;
}
