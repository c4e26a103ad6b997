// qn_big.h — the quasi-Newton operators for memories beyond the single-wave coefficient kernels of qn.hip
// (mem > 64 inverse, > 32 forward L-BFGS / L-SR1). Included by qn.hip inside its anonymous namespace.
//
// The reference accepts any `mem` (src/lbfgs.jl:26-57, src/lsr1.jl:19-34). The fast path of qn.hip passes the slot
// order, ys[] and the column pointers BY VALUE in kernel arguments (<= 64 slots) and runs each m x m recurrence on
// one wave with a lane per coefficient. Here the same three-stage shape — panel dots, coefficients, one combine
// pass — works off DEVICE-RESIDENT metadata, so nothing is sized by a compile-time cap:
//   * `meta`: ys[], age[], the active slots oldest->newest and newest->oldest, the shifted-solve pair order; uploaded
//     by push!/reset! (which synchronise anyway), read by every later apply — an apply still issues no copy;
//   * coefficient kernels: one workgroup, thread-strided loops over the coefficients, block reductions, the m x m
//     work arrays in the handle's device scalar region;
//   * combine: ONE launch for any number of columns; column pointers are derived in the kernel from the panel base,
//     the leading dimension and the device order (a "recipe"), coefficients are read from device memory;
//   * panel dots / dual-x dots: the existing kernels, called in chunks of <= 128 / 64 columns.
// Elementwise statements keep the reference's association order exactly as in combine_kernel; reductions and the
// coefficient recurrences are the same algebra with a different (fixed) summation order — parity bars unchanged.

constexpr int kBigMaxMem = 4096;   // LDS of the coefficient kernels holds a few mem-sized double arrays

struct OrdDev {
  int na, mem;
  const int *ord;        // active slots in the order the dots / columns are issued
  const double *ys;      // by slot
  const long long *age;  // by slot
  double gamma;
  int use_gamma, is_f32;
};

// ---- meta buffer (device): [ys: mem doubles][age: mem int64][ord_old][ord_new][pos2: mem ints each]
inline size_t meta_bytes(int64_t mem) { return (size_t)mem * (8 + 8 + 4 + 4 + 4); }
inline const double *meta_ys(const mxlo_qn *h) { return (const double *)h->meta; }
inline const long long *meta_age(const mxlo_qn *h) { return (const long long *)((const char *)h->meta + 8 * h->mem); }
inline const int *meta_ord_old(const mxlo_qn *h) { return (const int *)((const char *)h->meta + 16 * h->mem); }
inline const int *meta_ord_new(const mxlo_qn *h) { return meta_ord_old(h) + h->mem; }
inline const int *meta_pos2(const mxlo_qn *h) { return meta_ord_old(h) + 2 * h->mem; }

inline int active_count(const mxlo_qn *h) {
  int na = 0;
  for (int64_t k = 0; k < h->mem; ++k) na += h->ys[k] != 0;
  return na;
}

// Upload the host mirrors that drive slot order. Called by every state change (create / accepted push! / reset!).
int32_t sync_meta(mxlo_qn *h) {
  if (!h->big) return MXLO_OK;
  const int64_t mem = h->mem;
  h->meta_host.assign(meta_bytes(mem), 0);
  double *ys = (double *)h->meta_host.data();
  long long *age = (long long *)(h->meta_host.data() + 8 * mem);
  int *oo = (int *)(h->meta_host.data() + 16 * mem), *on = oo + mem, *p2 = oo + 2 * mem;
  int na = 0;
  for (int64_t k = 0; k < mem; ++k) {
    ys[k] = h->ys[k];
    age[k] = h->age[k];
  }
  for (int64_t i = 0; i < mem; ++i) {
    const int64_t k = (h->insert0 + i) % mem;
    if (h->ys[k] != 0) oo[na++] = (int)k;
  }
  int nn = 0;
  for (int64_t i = 0; i < mem; ++i) {
    const int64_t k = ((h->insert0 - 1 - i) % mem + mem) % mem;
    if (h->ys[k] != 0) on[nn++] = (int)k;
  }
  // solve_shifted_system! walks k = mod(insert + j - 1, mem) + 1, j = 1..mem (src/utilities.jl:228): one slot after
  // the oldest-first order. pos2[i] = position in ord_old of the i-th pair of that walk.
  int np = 0;
  for (int64_t j = 1; j <= mem; ++j) {
    const int64_t k = (h->insert0 + j) % mem;
    if (h->ys[k] == 0) continue;
    int pos = 0;
    while (oo[pos] != (int)k) ++pos;
    p2[np++] = pos;
  }
  MXLO_HIP(hipMemcpyAsync(h->meta, h->meta_host.data(), h->meta_host.size(), hipMemcpyHostToDevice, h->ctx->stream));
  MXLO_HIP(hipStreamSynchronize(h->ctx->stream));   // the staging vector is reused by the next state change
  return MXLO_OK;
}

inline OrdDev ord_dev(const mxlo_qn *h, bool newest_first) {
  OrdDev O;
  O.na = active_count(h);
  O.mem = (int)h->mem;
  O.ord = newest_first ? meta_ord_new(h) : meta_ord_old(h);
  O.ys = meta_ys(h);
  O.age = meta_age(h);
  O.gamma = h->scaling_factor;
  O.use_gamma = h->scaling ? 1 : 0;
  O.is_f32 = h->dtype == MXLO_F32;
  return O;
}

// panel_dots over any number of columns (the reduction workspace holds kMaxRedCols columns per call)
template <typename T>
int32_t panel_dots_any(mxlo_ctx *ctx, const std::vector<const T *> &cols, const T *x, int64_t n, double *out) {
  for (size_t done = 0; done < cols.size(); done += kMaxRedCols) {
    const int nc = (int)std::min<size_t>(kMaxRedCols, cols.size() - done);
    MXLO_TRY(panel_dots<T>(ctx, cols.data() + done, nc, x, n, out + done));
  }
  return MXLO_OK;
}
template <typename T>
int32_t panel_dots2_any(mxlo_ctx *ctx, const std::vector<const T *> &cols, const T *x1, const T *x2, int64_t npad,
                        double *out1, double *out2) {
  for (size_t done = 0; done < cols.size(); done += kMaxRedCols / 2) {
    const int nc = (int)std::min<size_t>(kMaxRedCols / 2, cols.size() - done);
    MXLO_TRY(panel_dots2<T>(ctx, cols.data() + done, nc, x1, x2, npad, out1 + done, out2 + done));
  }
  return MXLO_OK;
}

// ---- block-wide helpers (one workgroup of kBlock threads) ---------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_allsum(v);
  __syncthreads();                       // red may still be read by the previous call's consumers
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- inverse two-loop in coefficient space (cf. inv_coef_kernel) ----------------------------------------------
__global__ void __launch_bounds__(kBlock)
inv_coef_big_kernel(const double *__restrict__ dots, double *__restrict__ coef, const double *__restrict__ SY,
                    const double *__restrict__ YS, const double *__restrict__ YY, double *__restrict__ alpha_out,
                    OrdDev O) {
  extern __shared__ double shb[];        // a[na], b[na], red[4]
  const int na = O.na, mem = O.mem, tid = threadIdx.x;
  double *a = shb, *b = shb + na, *red = shb + 2 * na;
  auto sy = [&](int i, int j) {          // s_i' y_j for slots i, j
    return O.age[j] >= O.age[i] ? SY[i + (int64_t)j * mem] : YS[j + (int64_t)i * mem];
  };
  auto yy = [&](int i, int j) { return O.age[j] >= O.age[i] ? YY[i + (int64_t)j * mem] : YY[j + (int64_t)i * mem]; };
  for (int t = tid; t < na; t += kBlock) a[t] = b[t] = 0.0;
  __syncthreads();
  for (int i = 0; i < na; ++i) {
    const int k = O.ord[i];
    double part = 0.0;
    for (int j = tid; j < i; j += kBlock) part += a[j] * sy(k, O.ord[j]);
    const double sq = dots[i] - block_sum(part, red);
    if (tid == 0) a[i] = rnd(sq / O.ys[k], O.is_f32);
    __syncthreads();
  }
  for (int t = tid; t < na; t += kBlock) alpha_out[O.ord[t]] = a[t];
  const double g = O.use_gamma ? O.gamma : 1.0;
  for (int i = na - 1; i >= 0; --i) {
    const int k = O.ord[i];
    double p1 = 0.0, p2 = 0.0;
    for (int j = tid; j < na; j += kBlock) {
      p1 += a[j] * yy(k, O.ord[j]);                    // sum_j alpha_j y_k'y_j
      if (j > i) p2 += b[j] * sy(O.ord[j], k);         // sum_{older j} beta_j s_j'y_k
    }
    const double yq = dots[na + i] - block_sum(p1, red);
    const double yr = g * yq + block_sum(p2, red);
    if (tid == 0) b[i] = rnd(a[i] - yr / O.ys[k], O.is_f32);
    __syncthreads();
  }
  for (int t = tid; t < na; t += kBlock) {
    coef[t] = a[t];                      // y columns, newest -> oldest
    coef[na + (na - 1 - t)] = b[t];      // s columns, oldest -> newest
  }
}

// ---- compact forward apply: w = -Cm'(Cm d) (+ d on the b half) (cf. cfwd_coef_kernel) ---------------------------
__global__ void __launch_bounds__(kBlock)
cfwd_coef_big_kernel(const double *__restrict__ dots, const double *__restrict__ Cm, double *__restrict__ coef, int r) {
  extern __shared__ double shb[];        // t[r]
  const int tid = threadIdx.x, w2 = 2 * r;
  for (int k = tid; k < r; k += kBlock) {
    double t = 0.0;
    for (int j = 0; j < w2; ++j) t += Cm[(int64_t)k * w2 + j] * dots[j];   // a_k'x = c_k'd
    shb[k] = t;
  }
  __syncthreads();
  for (int j = tid; j < w2; j += kBlock) {
    double w = j >= r ? dots[j] : 0.0;                                      // + b_j (b_j'x)
    for (int k = 0; k < r; ++k) w -= Cm[(int64_t)k * w2 + j] * shb[k];      // - a_k (a_k'x)
    coef[j] = w;
  }
}

// ---- forward push!, Gram form: coefficients of a_k on [s.., b..] (cf. afwd_coef_kernel; src/lbfgs.jl:236-250) ----
__global__ void __launch_bounds__(kBlock)
afwd_coef_big_kernel(const double *__restrict__ SS, const double *__restrict__ YSf, double *__restrict__ Cm, OrdDev O) {
  extern __shared__ double shb[];        // zk[2r], as[r], red[4]
  const int r = O.na, mem = O.mem, w = 2 * r, tid = threadIdx.x;
  double *zk = shb, *as = shb + w, *red = shb + w + r;
  for (int k = 0; k < r; ++k) {
    for (int j = tid; j < w; j += kBlock) {        // Z[j][k] = <basis_j, s_k>
      const int jj = j < r ? j : j - r;
      const double g = (j < r ? SS : YSf)[(int64_t)O.ord[jj] * mem + O.ord[k]];
      zk[j] = j < r ? g : g / sqrt(O.ys[O.ord[jj]]);
    }
    __syncthreads();
    for (int l = tid; l < k; l += kBlock) {        // dot(a_l, s_k): independent of the running a_k
      double s = 0.0;
      for (int j = 0; j < w; ++j) s += Cm[(int64_t)l * w + j] * zk[j];
      as[l] = s;
    }
    __syncthreads();
    double part = 0.0;
    for (int j = tid; j < w; j += kBlock) {
      double c = (j == k) ? 1.0 / O.gamma : 0.0;                       // a_k = s_k / γ                 (:239)
      for (int l = 0; l < k; ++l) {
        if (j == r + l) c += zk[r + l];                               // += dot(b_l, s_k) b_l           (:244)
        c -= as[l] * Cm[(int64_t)l * w + j];                          // -= dot(a_l, s_k) a_l           (:245)
      }
      Cm[(int64_t)k * w + j] = c;
      part += c * zk[j];
    }
    const double nn = block_sum(part, red);                           // dot(s_k, a_k)                  (:248)
    const double sc = 1.0 / sqrt(nn);
    for (int j = tid; j < w; j += kBlock) Cm[(int64_t)k * w + j] *= sc;
    __syncthreads();
  }
}

// ---- L-SR1 push!, Gram form (cf. asr1_coef_kernel; src/lsr1.jl:166-178): basis [y.., s..] -----------------------
__global__ void __launch_bounds__(kBlock)
asr1_coef_big_kernel(const double *__restrict__ SS, const double *__restrict__ YSf, double *__restrict__ Cm,
                     double *__restrict__ as_out, OrdDev O) {
  extern __shared__ double shb[];        // zk[2r], asl[r], as[r], red[4]
  const int r = O.na, mem = O.mem, w = 2 * r, tid = threadIdx.x;
  double *zk = shb, *asl = shb + w, *as = shb + w + r, *red = shb + w + 2 * r;
  for (int k = 0; k < r; ++k) {
    for (int j = tid; j < w; j += kBlock) {
      const int jj = j < r ? j : j - r;
      zk[j] = (j < r ? YSf : SS)[(int64_t)O.ord[jj] * mem + O.ord[k]];
    }
    __syncthreads();
    for (int l = tid; l < k; l += kBlock) {
      double s = 0.0;
      for (int j = 0; j < w; ++j) s += Cm[(int64_t)l * w + j] * zk[j];
      as[l] = s / asl[l];                                                    // dot(a_l,s_k)/as_l (:173)
    }
    __syncthreads();
    double part = 0.0;
    for (int j = tid; j < w; j += kBlock) {
      double c = (j == k) ? 1.0 : ((j == r + k) ? -1.0 / O.gamma : 0.0);     // y_k - s_k/γ       (:169)
      for (int l = 0; l < k; ++l) c -= as[l] * Cm[(int64_t)l * w + j];       //                   (:174)
      Cm[(int64_t)k * w + j] = c;
      part += c * zk[j];
    }
    const double ask = block_sum(part, red);                                 // as_k = a_k's_k    (:177)
    if (tid == 0) {
      asl[k] = ask;
      as_out[O.ord[k]] = ask;
    }
    __syncthreads();
  }
}

// grid-strided scalar helpers over device order (cf. lsr1_coef_kernel, div_as_kernel, copy_coef_kernel, gram_update_kernel)
__global__ void lsr1_coef_big_kernel(const double *__restrict__ dots, double *__restrict__ coef,
                                     const double *__restrict__ as_, OrdDev O, double alpha, int ct_f32) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O.na) return;
  const double d = rnd(dots[i], O.is_f32), as = rnd(as_[O.ord[i]], O.is_f32);
  if (ct_f32) coef[i] = (double)(((float)alpha * (float)d) / (float)as);
  else coef[i] = (alpha * d) / as;
}
__global__ void copy_coef_big_kernel(const double *__restrict__ dots, double *__restrict__ coef, int n, int is_f32) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) coef[i] = rnd(dots[i], is_f32);
}
__global__ void gather_as_big_kernel(const double *__restrict__ as_, double *__restrict__ coef, OrdDev O) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < O.na) coef[i] = as_[O.ord[i]];
}
__global__ void gram_update_big_kernel(double *__restrict__ SS, double *__restrict__ YSf, double *__restrict__ YY,
                                       const double *__restrict__ tmp, int mem, int ins) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= mem) return;
  SS[(int64_t)k * mem + ins] = tmp[k];
  SS[(int64_t)ins * mem + k] = tmp[k];
  YSf[(int64_t)k * mem + ins] = tmp[mem + k];       // y_k ' s_new
  YSf[(int64_t)ins * mem + k] = tmp[2 * mem + k];   // y_new ' s_k
  YY[(int64_t)k * mem + ins] = tmp[3 * mem + k];
  YY[(int64_t)ins * mem + k] = tmp[3 * mem + k];
}

// ---- one-launch combine over any number of columns --------------------------------------------------------------
enum ColKind { CK_INV = 0, CK_PAIR = 1, CK_SPLIT = 2, CK_ONE = 3 };
template <typename T>
struct BigCombine {
  const T *p0, *p1;      // panel bases
  int64_t ld;
  const int *ord;        // device order
  int na;                // active slots
  int kind;              // ColKind: how column c maps to (panel, slot)
  int ncol, nfirst, use_gamma;
  double gamma, alpha, beta;
  const double *coef;    // device coefficients, one per column (CM_AXPYS: coef[ncol] is the factor of x)
  double shift;
};
template <typename T>
__device__ __forceinline__ const T *big_col(const BigCombine<T> &A, int c) {
  // one in-range read of the device order per column, whatever the recipe
  const int k = A.kind, na = A.na;
  const bool second = k == CK_PAIR ? (c & 1) != 0 : (k == CK_ONE ? false : c >= na);
  const int idx = k == CK_PAIR ? (c >> 1) : (k == CK_ONE ? c : (c < na ? c : (k == CK_INV ? 2 * na - 1 - c : c - na)));
  const T *base = second ? A.p1 : A.p0;
  return base + (int64_t)A.ord[idx] * A.ld;
}

template <typename T, typename CA, typename CB, int MODE, bool BETA0, int VEC, bool NT>
__global__ void __launch_bounds__(kBlock)
combine_big_kernel(T *__restrict__ res, const T *__restrict__ x, BigCombine<T> A, int64_t nvec) {
  using V = typename VecOf<T, VEC>::type;
  const T g = (T)A.gamma;
  const CA al = (CA)A.alpha;
  const CB be = (CB)A.beta;
  const int ncol = A.ncol;
  const T c0x = MODE == CM_AXPYS ? (T)A.coef[ncol] : T(0);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kBlock) {
    T q[VEC];
    V xv;
    if constexpr (MODE == CM_DIAG_FWD || MODE == CM_DIAG_SR1) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) q[e] = A.use_gamma ? (T)1 / g : (T)1;
    } else {
      xv = ldg<NT>(reinterpret_cast<const V *>(x + i * VEC));
      V rv;
      if constexpr (MODE == CM_LSR1 && !BETA0) rv = ldg<NT>(reinterpret_cast<const V *>(res + i * VEC));
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const T xe = vget<T, VEC>(xv, e);
        if constexpr (MODE == CM_FWD || MODE == CM_CFWD) q[e] = A.use_gamma ? xe / g : xe;
        else if constexpr (MODE == CM_INV) q[e] = xe;
        else if constexpr (MODE == CM_LSR1)
          q[e] = fin_ab<T, CA, CB, BETA0>((al * (CA)xe) / (CA)g, be, BETA0 ? T(0) : vget<T, VEC>(rv, e));
        else if constexpr (MODE == CM_AXPYS) q[e] = c0x * xe;
      }
    }
    constexpr int UB = 4;
    for (int cb = 0; cb < ncol; cb += UB) {
      const int nb = ncol - cb < UB ? ncol - cb : UB;
      V cv[UB];
      double cf[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (u < nb) {
          cv[u] = ldg<NT>(reinterpret_cast<const V *>(big_col(A, cb + u) + i * VEC));
          cf[u] = (MODE == CM_DIAG_FWD) ? 0.0 : A.coef[cb + u];
        }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (u >= nb) break;
        const int c = cb + u;
        if constexpr (MODE == CM_FWD || MODE == CM_DIAG_FWD) {
          if (u & 1) continue;             // pairs (b, a) handled on the even member (batches are pair-aligned)
          const T cbv = (T)cf[u], cav = (T)cf[u + 1];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const T b = vget<T, VEC>(cv[u], e), a = vget<T, VEC>(cv[u + 1], e);
            if constexpr (MODE == CM_FWD) q[e] = q[e] + ((cbv * b) - (cav * a));      // lbfgs.jl:194
            else q[e] = q[e] + ((b * b) - (a * a));                                   // lbfgs.jl:391
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
        } else if constexpr (MODE == CM_LSR1) {
          const CA cc = (CA)cf[u];
#pragma unroll
          for (int e = 0; e < VEC; ++e) q[e] = (T)((CA)q[e] + (cc * (CA)vget<T, VEC>(cv[u], e)));   // lsr1.jl:103
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
    }
    if constexpr (MODE == CM_INV) {
      if (ncol == 0 && A.use_gamma) {      // no pairs: `q .*= scaling_factor` still executes (γ == 1 after reset!)
#pragma unroll
        for (int e = 0; e < VEC; ++e) q[e] = q[e] * g;
      }
    }
    V out;
    if constexpr (MODE == CM_FWD || MODE == CM_INV || MODE == CM_CFWD) {
      V rv;
      if constexpr (!BETA0) rv = ldg<NT>(reinterpret_cast<const V *>(res + i * VEC));
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        vset<T, VEC>(out, e, fin_ab<T, CA, CB, BETA0>(al * (CA)q[e], be, BETA0 ? T(0) : vget<T, VEC>(rv, e)));
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) vset<T, VEC>(out, e, q[e]);
    }
    if constexpr (MODE == CM_FWD || MODE == CM_INV || MODE == CM_LSR1 || MODE == CM_CFWD) {
      if (A.shift != 0.0) {
        const T sh = (T)A.shift;
#pragma unroll
        for (int e = 0; e < VEC; ++e) vset<T, VEC>(out, e, vget<T, VEC>(out, e) + (sh * vget<T, VEC>(xv, e)));
      }
    }
    stg<NT>(reinterpret_cast<V *>(res + i * VEC), out);
  }
}

template <typename T, int MODE>
int32_t launch_combine_big_part(mxlo_ctx *ctx, T *res, const T *x, const BigCombine<T> &A, int64_t n, int32_t flags,
                                bool vec) {
  if (n <= 0) return MXLO_OK;
  constexpr int VECF = Vec16<T>::N;
  auto go = [&]<typename CA, typename CB, bool B0, int VEC>() -> int32_t {
    const int64_t nvec = n / VEC;
    const int grid = grid_for(ctx, nvec, kBlock, 0);
    const bool nt = (int64_t)sizeof(T) * n * (A.ncol + 2) >= ctx->tune.nt_min_bytes;
    if (nt)
      hipLaunchKernelGGL((combine_big_kernel<T, CA, CB, MODE, B0, VEC, true>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         res, x, A, nvec);
    else
      hipLaunchKernelGGL((combine_big_kernel<T, CA, CB, MODE, B0, VEC, false>), dim3(grid), dim3(kBlock), 0, ctx->stream,
                         res, x, A, nvec);
    MXLO_LAUNCH_CHECK();
    return MXLO_OK;
  };
  constexpr bool uses_ab = (MODE == CM_FWD || MODE == CM_INV || MODE == CM_LSR1 || MODE == CM_CFWD);
  if constexpr (!uses_ab) {
    return vec ? go.template operator()<T, T, true, VECF>() : go.template operator()<T, T, true, 1>();
  } else {
    return dispatch_ab<T>(A.beta == 0 ? 0.0 : 1.0, flags, [&]<typename CA, typename CB, bool B0>() -> int32_t {
      return vec ? go.template operator()<CA, CB, B0, VECF>() : go.template operator()<CA, CB, B0, 1>();
    });
  }
}

// panel columns are 16-byte aligned (ld is a multiple of the vector width); the vector path also needs res / x aligned
template <typename T, int MODE>
int32_t launch_combine_big(mxlo_ctx *ctx, T *res, const T *x, const BigCombine<T> &A, int64_t n, int32_t flags) {
  if (n <= 0) return MXLO_OK;
  constexpr int VECF = Vec16<T>::N;
  const bool vec = (((uintptr_t)res & 15u) == 0) && (!x || ((uintptr_t)x & 15u) == 0) && n >= VECF;
  if (!vec) return launch_combine_big_part<T, MODE>(ctx, res, x, A, n, flags, false);
  const int64_t nbody = (n / VECF) * VECF;
  MXLO_TRY((launch_combine_big_part<T, MODE>(ctx, res, x, A, nbody, flags, true)));
  if (nbody < n) {
    BigCombine<T> At = A;
    At.p0 = A.p0 ? A.p0 + nbody : A.p0;
    At.p1 = A.p1 ? A.p1 + nbody : A.p1;
    return launch_combine_big_part<T, MODE>(ctx, res + nbody, x ? x + nbody : x, At, n - nbody, flags, false);
  }
  return MXLO_OK;
}

// ---- applies ----------------------------------------------------------------------------------------------------
template <typename T>
int32_t inv_mul_big(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags, double shift) {
  mxlo_ctx *ctx = h->ctx;
  const OrdDev O = ord_dev(h, /*newest_first=*/true);
  const int na = O.na;
  double *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef;
  if (na > 0) {
    const std::vector<int> oh = ord_host(h, true);
    std::vector<const T *> cols(2 * na);
    for (int i = 0; i < na; ++i) {
      cols[i] = col<T>(h->S, h->ld, oh[i]);
      cols[na + i] = col<T>(h->Y, h->ld, oh[i]);
    }
    MXLO_TRY(panel_dots_any<T>(ctx, cols, x, h->n, dots));
    hipLaunchKernelGGL(inv_coef_big_kernel, dim3(1), dim3(kBlock), sizeof(double) * (2 * na + 8), ctx->stream, dots, coef,
                       h->dsc + h->lay.SY, h->dsc + h->lay.YS, h->dsc + h->lay.YY, h->dsc + h->lay.alpha, O);
    MXLO_LAUNCH_CHECK();
  }
  BigCombine<T> A{(const T *)h->Y, (const T *)h->S, h->ld, O.ord, na, CK_INV, 2 * na, na, h->scaling, h->scaling_factor,
                  alpha, beta, coef, shift};
  return launch_combine_big<T, CM_INV>(ctx, res, x, A, h->n, flags);
}

template <typename T>
int32_t ensure_A_big(mxlo_qn *h);

template <typename T>
int32_t fwd_mul_big(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags, double shift) {
  mxlo_ctx *ctx = h->ctx;
  const OrdDev O = ord_dev(h, false);
  const int na = O.na;
  double *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef;
  const std::vector<int> oh = ord_host(h, false);
  std::vector<const T *> cols(2 * na);
  if (na > 0 && !h->A_valid) {   // compact form: x/γ + [S B]·w
    for (int i = 0; i < na; ++i) {
      cols[i] = col<T>(h->S, h->ld, oh[i]);
      cols[na + i] = col<T>(h->B, h->ld, oh[i]);
    }
    MXLO_TRY(panel_dots_any<T>(ctx, cols, x, h->n, dots));
    hipLaunchKernelGGL(cfwd_coef_big_kernel, dim3(1), dim3(kBlock), sizeof(double) * (na + 8), ctx->stream, dots,
                       h->dsc + h->lay.Cm, coef, na);
    MXLO_LAUNCH_CHECK();
    BigCombine<T> A{(const T *)h->S, (const T *)h->B, h->ld, O.ord, na, CK_SPLIT, 2 * na, 0, h->scaling,
                    h->scaling_factor, alpha, beta, coef, shift};
    return launch_combine_big<T, CM_CFWD>(ctx, res, x, A, h->n, flags);
  }
  if (na > 0) {
    for (int i = 0; i < na; ++i) {  // pair order (b_k, a_k): coef = (bx, ax)
      cols[2 * i] = col<T>(h->B, h->ld, oh[i]);
      cols[2 * i + 1] = col<T>(h->A, h->ld, oh[i]);
    }
    MXLO_TRY(panel_dots_any<T>(ctx, cols, x, h->n, dots));
    hipLaunchKernelGGL(copy_coef_big_kernel, dim3((2 * na + 255) / 256), dim3(256), 0, ctx->stream, dots, coef, 2 * na,
                       (int)(h->dtype == MXLO_F32));
    MXLO_LAUNCH_CHECK();
  }
  BigCombine<T> A{(const T *)h->B, (const T *)h->A, h->ld, O.ord, na, CK_PAIR, 2 * na, 0, h->scaling, h->scaling_factor,
                  alpha, beta, coef, shift};
  return launch_combine_big<T, CM_FWD>(ctx, res, x, A, h->n, flags);
}

template <typename T>
int32_t lsr1_mul_big(mxlo_qn *h, T *res, const T *x, double alpha, double beta, int32_t flags, double shift) {
  mxlo_ctx *ctx = h->ctx;
  const OrdDev O = ord_dev(h, false);
  const int na = O.na;
  double *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef;
  if (na > 0) {
    const std::vector<int> oh = ord_host(h, false);
    std::vector<const T *> cols(na);
    for (int i = 0; i < na; ++i) cols[i] = col<T>(h->A, h->ld, oh[i]);
    MXLO_TRY(panel_dots_any<T>(ctx, cols, x, h->n, dots));
    const int ct_f32 = alpha_is_f64(sizeof(T), flags) ? 0 : 1;
    hipLaunchKernelGGL(lsr1_coef_big_kernel, dim3((na + 255) / 256), dim3(256), 0, ctx->stream, dots, coef,
                       h->dsc + h->lay.as_, O, alpha, ct_f32);
    MXLO_LAUNCH_CHECK();
  }
  BigCombine<T> A{(const T *)h->A, (const T *)h->A, h->ld, O.ord, na, CK_ONE, na, 0, 1, h->scaling_factor, alpha, beta, coef,
                  shift};   // p1 unused for CK_ONE (kept non-null: the column select may be evaluated for both panels)
  return launch_combine_big<T, CM_LSR1>(ctx, res, x, A, h->n, flags);
}

// ---- Gram bookkeeping of push! ---------------------------------------------------------------------------------------
template <typename T>
int32_t gram_update_slot_big(mxlo_qn *h, int64_t slot) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n, mem = h->mem;
  double *tmp = h->dsc + h->lay.gtmp;
  std::vector<const T *> cs(mem), cy(mem);
  for (int64_t k = 0; k < mem; ++k) {
    cs[k] = col<T>(h->S, h->ld, k);
    cy[k] = col<T>(h->Y, h->ld, k);
  }
  constexpr int VECP = Vec16<T>::N;
  const int64_t npad = (n + VECP - 1) / VECP * VECP;
  const T *ss = col<T>(h->S, h->ld, slot), *yy = col<T>(h->Y, h->ld, slot);
  MXLO_TRY(panel_dots2_any<T>(ctx, cs, ss, yy, npad, tmp, tmp + 2 * mem));            // S's, S'y
  MXLO_TRY(panel_dots2_any<T>(ctx, cy, ss, yy, npad, tmp + mem, tmp + 3 * mem));      // Y's, Y'y
  hipLaunchKernelGGL(gram_update_big_kernel, dim3((unsigned)((mem + 255) / 256)), dim3(256), 0, ctx->stream,
                     h->dsc + h->lay.SS, h->dsc + h->lay.YSf, h->dsc + h->lay.YY, tmp, (int)mem, (int)slot);
  MXLO_LAUNCH_CHECK();
  return MXLO_OK;
}

// A = [in panels]·C, out_k = sum_j C[k][j] in_j: outputs in chunks of kGemmOut, inputs in chunks of kGemmIn with the
// partial sums carried through the output columns (f64 accumulators inside a chunk).
template <typename T>
int32_t panel_gemm_big(mxlo_ctx *ctx, const std::vector<const T *> &in, const std::vector<T *> &out, const double *C,
                       int64_t n);

// forward push!: rebuild Cm with the freshly written slot LAST (called before insert0 advances)
template <typename T>
int32_t fwd_rebuild_gram_big(mxlo_qn *h, int64_t ins) {
  mxlo_ctx *ctx = h->ctx;
  MXLO_TRY(gram_update_slot_big<T>(h, ins));
  // the order "slots (ins+1 .. ins+mem) mod mem" is oldest->newest AFTER insert0 advances: upload it now
  const int64_t keep = h->insert0;
  h->insert0 = (ins + 1) % h->mem;
  MXLO_TRY(sync_meta(h));
  OrdDev O = ord_dev(h, false);
  h->insert0 = keep;
  O.gamma = h->scaling_factor;   // (:239) divides unconditionally; γ == 1 without scaling
  const int r = O.na;
  hipLaunchKernelGGL(afwd_coef_big_kernel, dim3(1), dim3(kBlock), sizeof(double) * (3 * r + 8), ctx->stream,
                     h->dsc + h->lay.SS, h->dsc + h->lay.YSf, h->dsc + h->lay.Cm, O);
  MXLO_LAUNCH_CHECK();
  h->A_valid = false;            // big memories are always compact: a_k = [S B]·c_k stays implicit
  return MXLO_OK;
}

template <typename T>
int32_t ensure_A_big(mxlo_qn *h) {
  if (h->kind != MXLO_QN_LBFGS_FWD) return MXLO_OK;
  MXLO_TRY(alloc_A(h));
  if (h->A_valid) return MXLO_OK;
  const std::vector<int> oh = ord_host(h, false);
  const int r = (int)oh.size();
  std::vector<const T *> in(2 * r);
  std::vector<T *> out(r);
  for (int j = 0; j < r; ++j) {
    in[j] = col<T>(h->S, h->ld, oh[j]);
    in[r + j] = col<T>(h->B, h->ld, oh[j]);
    out[j] = col<T>(h->A, h->ld, oh[j]);
  }
  MXLO_TRY(panel_gemm_big<T>(h->ctx, in, out, h->dsc + h->lay.Cm, h->n));
  h->A_valid = true;
  return MXLO_OK;
}

// ---- solve_shifted_system! in coefficient space, any number of pairs (cf. shifted_*_kernel) ------------------------
struct ShiftDev {
  int r, mem;
  const int *ord;    // active slots oldest -> newest
  const int *pos2;   // position in ord of the i-th pair of the reference's solve order
  const double *ys;
};

__global__ void __launch_bounds__(kBlock)
shifted_gram_big_kernel(const double *__restrict__ SS, const double *__restrict__ YSf, const double *__restrict__ YY,
                        const double *__restrict__ Cm, ShiftDev mp, double *__restrict__ G, double *__restrict__ W,
                        double *__restrict__ M, double *__restrict__ Tm) {
  const int r = mp.r, w2 = 2 * r, nu = 2 * r, mem = mp.mem, tid = threadIdx.x;
  for (int idx = tid; idx < w2 * w2; idx += kBlock) {
    const int p = idx / w2, q = idx % w2;
    const int sp = mp.ord[p < r ? p : p - r], sq = mp.ord[q < r ? q : q - r];
    double v;
    if (p < r && q < r) v = SS[(int64_t)sp * mem + sq];
    else if (p < r) v = YSf[(int64_t)sq * mem + sp] / sqrt(mp.ys[sq]);
    else if (q < r) v = YSf[(int64_t)sp * mem + sq] / sqrt(mp.ys[sp]);
    else v = YY[(int64_t)sp * mem + sq] / (sqrt(mp.ys[sp]) * sqrt(mp.ys[sq]));
    M[idx] = v;
  }
  for (int idx = tid; idx < nu * w2; idx += kBlock) {
    const int t = idx / w2, p = idx % w2, P = mp.pos2[t >> 1];
    W[idx] = (t & 1) ? (p == r + P ? 1.0 : 0.0) : Cm[(int64_t)P * w2 + p];   // even: a_k (sign +1), odd: b_k (-1)
  }
  __threadfence_block();
  __syncthreads();
  for (int idx = tid; idx < nu * w2; idx += kBlock) {
    const int t = idx / w2, q = idx % w2;
    double acc = 0.0;
    for (int p = 0; p < w2; ++p) acc = fma(W[t * w2 + p], M[p * w2 + q], acc);
    Tm[idx] = acc;
  }
  __threadfence_block();
  __syncthreads();
  for (int idx = tid; idx < nu * nu; idx += kBlock) {
    const int t = idx / nu, u = idx % nu;
    double acc = 0.0;
    for (int q = 0; q < w2; ++q) acc = fma(Tm[t * w2 + q], W[u * w2 + q], acc);
    G[idx] = acc;
  }
}

__global__ void shifted_gv_big_kernel(const double *__restrict__ W, const double *__restrict__ d, double *__restrict__ gv,
                                      int nu, int w2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nu) return;
  double acc = 0.0;
  for (int p = 0; p < w2; ++p) acc = fma(W[(int64_t)t * w2 + p], d[p], acc);
  gv[t] = acc;
}

__global__ void shifted_back_big_kernel(const double *__restrict__ W, const double *__restrict__ cx,
                                        double *__restrict__ coef, int nu, int w2, double x0) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < w2) {
    double acc = 0.0;
    for (int t = 0; t < nu; ++t) acc = fma(W[(int64_t)t * w2 + p], cx[t], acc);
    coef[p] = acc;
  }
  if (p == 0) coef[w2] = x0;           // the factor of b travels right past the columns
}

// the recursion of src/utilities.jl:226-246 on coefficient vectors (P is nu x nu in device memory)
__global__ void __launch_bounds__(kBlock)
shifted_coef_big_kernel(const double *__restrict__ G, const double *__restrict__ gvec, double *__restrict__ cx,
                        double *__restrict__ P, int nu, double x0, int is_f32) {
  extern __shared__ double shb[];        // v[nu], c2[nu], red[4]
  double *v = shb, *c2 = shb + nu, *red = shb + 2 * nu;
  const int tid = threadIdx.x;
  for (int j = tid; j < nu; j += kBlock) cx[j] = 0.0;
  for (int i = 0; i < nu; ++i) {
    const int sign_i = (i & 1) ? -1 : 1;
    for (int t = tid; t < i; t += kBlock) {           // c2[t] = (sign_t v_t) * dot(p_t, u_i)      (:235-237)
      double c0 = 0.0;
      for (int j = 0; j <= t; ++j) c0 += P[(int64_t)t * nu + j] * G[(int64_t)j * nu + i];
      c2[t] = (((t & 1) ? -1.0 : 1.0) * v[t]) * c0;
    }
    __syncthreads();
    double up = 0.0, pb = 0.0;
    for (int j = tid; j <= i; j += kBlock) {          // p_i = x0 u_i + sum_t c2[t] p_t               (:231,:238)
      double pij = (j == i) ? x0 : 0.0;
      for (int t = j; t < i; ++t) pij += c2[t] * P[(int64_t)t * nu + j];
      P[(int64_t)i * nu + j] = pij;
      up += pij * G[(int64_t)i * nu + j];             // dot(u_i, p_i)
      pb += pij * gvec[j];                            // p_i' b
    }
    up = block_sum(up, red);
    pb = block_sum(pb, red);
    const double vi = rnd(1.0 / (1.0 - sign_i * up), is_f32);                                          // (:242)
    if (tid == 0) v[i] = vi;
    for (int j = tid; j <= i; j += kBlock) cx[j] += ((sign_i * vi) * pb) * P[(int64_t)i * nu + j];     // (:243-244)
    __threadfence_block();
    __syncthreads();
  }
}

template <typename T>
int32_t solve_shifted_big(mxlo_qn *h, T *x, const T *b, double sigma) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n;
  const OrdDev O = ord_dev(h, false);
  const int r = O.na, nu = 2 * r, w2 = 2 * r;
  MXLO_REQUIRE(h->gram_ok, MXLO_ESTATE, "solve_shifted_system!: Gram matrices not maintained");
  double *G = h->dsc + h->lay.G, *W = h->dsc + h->lay.Wm, *gv = h->dsc + h->lay.g, *cx = h->dsc + h->lay.cx,
         *dots = h->dsc + h->lay.dots, *coef = h->dsc + h->lay.coef;
  double *M = h->dsc + h->lay.bigM, *Tm = h->dsc + h->lay.bigT, *P = h->dsc + h->lay.bigP;
  const ShiftDev mp{r, (int)h->mem, meta_ord_old(h), meta_pos2(h), meta_ys(h)};
  if (!h->G_valid && nu > 0) {
    hipLaunchKernelGGL(shifted_gram_big_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, h->dsc + h->lay.SS,
                       h->dsc + h->lay.YSf, h->dsc + h->lay.YY, h->dsc + h->lay.Cm, mp, G, W, M, Tm);
    MXLO_LAUNCH_CHECK();
    h->G_valid = true;
  }
  const double g_inv = rT<T>(1.0 / h->scaling_factor);                 // :219
  const double x0 = rT<T>(1.0 / (g_inv + sigma));                      // :220
  if (nu > 0) {
    const std::vector<int> oh = ord_host(h, false);
    std::vector<const T *> bcols(w2);
    for (int i = 0; i < r; ++i) {
      bcols[i] = col<T>(h->S, h->ld, oh[i]);
      bcols[r + i] = col<T>(h->B, h->ld, oh[i]);
    }
    MXLO_TRY(panel_dots_any<T>(ctx, bcols, b, n, dots));              // [S B]'b : the only reduction pass over n
    hipLaunchKernelGGL(shifted_gv_big_kernel, dim3((nu + 255) / 256), dim3(256), 0, ctx->stream, W, dots, gv, nu, w2);
    MXLO_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(shifted_coef_big_kernel, dim3(1), dim3(kBlock), sizeof(double) * (2 * nu + 8), ctx->stream, G, gv,
                     cx, P, nu, x0, (int)(sizeof(T) == 4));
  MXLO_LAUNCH_CHECK();
  hipLaunchKernelGGL(shifted_back_big_kernel, dim3((w2 + 256) / 256), dim3(256), 0, ctx->stream, W, cx, coef, nu, w2, x0);
  MXLO_LAUNCH_CHECK();
  BigCombine<T> A{(const T *)h->S, (const T *)h->B, h->ld, O.ord, r, CK_SPLIT, w2, 0, 0, 1.0, 1.0, 0.0, coef, 0.0};
  return launch_combine_big<T, CM_AXPYS>(ctx, x, b, A, n, 0);
}

template <typename T>
int32_t diag_big(mxlo_qn *h, T *d) {
  const OrdDev O = ord_dev(h, false);
  double *coef = h->dsc + h->lay.coef;
  if (h->kind == MXLO_QN_LBFGS_FWD) {
    MXLO_TRY(ensure_A_big<T>(h));
    BigCombine<T> A{(const T *)h->B, (const T *)h->A, h->ld, O.ord, O.na, CK_PAIR, 2 * O.na, 0, h->scaling,
                    h->scaling_factor, 1.0, 0.0, coef, 0.0};
    return launch_combine_big<T, CM_DIAG_FWD>(h->ctx, d, (const T *)nullptr, A, h->n, 0);
  }
  if (O.na > 0) {
    hipLaunchKernelGGL(gather_as_big_kernel, dim3((O.na + 255) / 256), dim3(256), 0, h->ctx->stream, h->dsc + h->lay.as_,
                       coef, O);
    MXLO_LAUNCH_CHECK();
  }
  BigCombine<T> A{(const T *)h->A, (const T *)h->A, h->ld, O.ord, O.na, CK_ONE, O.na, 0, h->scaling, h->scaling_factor, 1.0,
                  0.0, coef, 0.0};
  return launch_combine_big<T, CM_DIAG_SR1>(h->ctx, d, (const T *)nullptr, A, h->n, 0);
}

// ---- push! ---------------------------------------------------------------------------------------------------------
// push_common! (src/lbfgs.jl:210-255) for a big memory: Gram bookkeeping with chunked dots; the forward operator is
// always compact (the a_k panel stays implicit, cf. MXLO_PUSH_COMPACT).
template <typename T>
int32_t lbfgs_push_common_big(mxlo_qn *h, const T *s, const T *y, double ys, double yy) {
  mxlo_ctx *ctx = h->ctx;
  const int64_t n = h->n, ins = h->insert0, mem = h->mem;
  T *si = col<T>(h->S, h->ld, ins), *yi = col<T>(h->Y, h->ld, ins);
  MXLO_HIP(hipMemcpyAsync(si, s, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));  // :220
  MXLO_HIP(hipMemcpyAsync(yi, y, sizeof(T) * n, hipMemcpyDeviceToDevice, ctx->stream));  // :221
  h->ys[ins] = ys;                                                                       // :222
  h->age[ins] = ++h->pushes;
  if (h->scaling) h->scaling_factor = rT<T>(ys / yy);                                    // :225
  h->G_valid = false;
  constexpr int VECP = Vec16<T>::N;
  const int64_t npad = (n + VECP - 1) / VECP * VECP;
  if (h->kind == MXLO_QN_LBFGS_INV) {
    std::vector<const T *> cs(mem), cy(mem);
    for (int64_t k = 0; k < mem; ++k) {
      cs[k] = col<T>(h->S, h->ld, k);
      cy[k] = col<T>(h->Y, h->ld, k);
    }
    MXLO_TRY(panel_dots_any<T>(ctx, cs, yi, n, h->dsc + h->lay.SY + ins * mem));                   // S'y_new
    return panel_dots2_any<T>(ctx, cy, yi, si, npad, h->dsc + h->lay.YY + ins * mem, h->dsc + h->lay.YS + ins * mem);
  }
  T *bi = col<T>(h->B, h->ld, ins);
  const T sq = sizeof(T) == 8 ? (T)std::sqrt(ys) : (T)sqrtf((float)ys);
  MXLO_TRY((launch_map<T, 1, false, false>(ctx, bi, yi, (const T *)nullptr, n, DivConstOp<T>{sq})));   // :232
  {
    const T *cols[1] = {bi};
    MXLO_TRY(panel_dots<T>(ctx, cols, 1, bi, n, h->dsc + h->lay.misc + 16 + ins));                 // norm_b^2
  }
  return fwd_rebuild_gram_big<T>(h, ins);
}

// L-SR1 rank-one terms after an accepted push (src/lsr1.jl:166-181), Gram form; insert0 has already advanced
template <typename T>
int32_t lsr1_rebuild_big(mxlo_qn *h, int64_t ins) {
  mxlo_ctx *ctx = h->ctx;
  MXLO_TRY(sync_meta(h));
  MXLO_TRY(gram_update_slot_big<T>(h, ins));
  const OrdDev O = ord_dev(h, false);
  const int r = O.na;
  hipLaunchKernelGGL(asr1_coef_big_kernel, dim3(1), dim3(kBlock), sizeof(double) * (4 * r + 8), ctx->stream,
                     h->dsc + h->lay.SS, h->dsc + h->lay.YSf, h->dsc + h->lay.Cm, h->dsc + h->lay.as_, O);
  MXLO_LAUNCH_CHECK();
  const std::vector<int> oh = ord_host(h, false);
  std::vector<const T *> in(2 * r);
  std::vector<T *> out(r);
  for (int j = 0; j < r; ++j) {
    in[j] = col<T>(h->Y, h->ld, oh[j]);
    in[r + j] = col<T>(h->S, h->ld, oh[j]);
    out[j] = col<T>(h->A, h->ld, oh[j]);
  }
  h->normA_valid = false;
  return panel_gemm_big<T>(ctx, in, out, h->dsc + h->lay.Cm, h->n);
}
