/* lo_oracle_impl.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the LinearOperators.jl v2.14.2 `mul!` hot path, one C
 * function per reference closure, every statement in the reference's own
 * evaluation order (n-ary `*` folds left, no FMA: build with
 * -ffp-contract=off). Included twice by lo_oracle.c with
 *     T   = double / float      (element type of the Julia vectors)
 *     SUF = f64 / f32
 * Citations are `src/<file>.jl:<line>` of the reference.
 *
 * Mixed precision (Julia does not convert caller scalars to T): with T == float,
 * ORC_ALPHA_F64 / ORC_BETA_F64 say which caller scalars are Float64; the α-term
 * of a statement is evaluated in promote_type(typeof(α), T), the β-term in
 * promote_type(typeof(β), T), their sum in the wider type, rounded once on store
 * (`mul!(res32, op32, v32, 2.0, 3.0)`: both flags; `α::Float32, β::Float64`: the
 * product (α*d)*v is rounded to Float32 before the Float64 addition).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this file. The product (libmxlo.so) never links or calls it.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* Scalar-type selection. Julia evaluates `α .* d .* v .+ β .* res` with α and β in the
 * types the CALLER passed: the α-term in CA = promote_type(typeof(α), T), the β-term in
 * CB = promote_type(typeof(β), T), and their sum in promote_type(CA, CB) — which is what C's
 * usual arithmetic conversions do to a (CA) + (CB) expression. BODY(CA, CB) is instantiated for
 * the combination the flags select (ORC_ALPHA_F64 / ORC_BETA_F64; both always double when T is). */
#define WITH_CT(BODY)                                                                            \
  do {                                                                                           \
    const int ad_ = sizeof(T) == 8 || (flags & ORC_ALPHA_F64);                                   \
    const int bd_ = sizeof(T) == 8 || (flags & ORC_BETA_F64);                                    \
    if (ad_ && bd_) {                                                                            \
      BODY(double, double);                                                                      \
    } else if (ad_) {                                                                            \
      BODY(double, float);                                                                       \
    } else if (bd_) {                                                                            \
      BODY(float, double);                                                                       \
    } else {                                                                                     \
      BODY(float, float);                                                                        \
    }                                                                                            \
  } while (0)

/* LinearAlgebra.dot on real vectors (OpenBLAS ddot/sdot in the reference; the
 * summation order of BLAS is implementation-defined, so any fixed order is an
 * equally valid restatement; results are pinned to tolerance only). Eight
 * independent partial sums mimic the SIMD kernels' shape and let gcc vectorise. */
static T FN(orc_dot)(const T *a, const T *b, int64_t n) {
  T p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int u = 0; u < 8; ++u) p[u] += a[i + u] * b[i + u];
  T s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  for (; i < n; ++i) s += a[i] * b[i];
  return s;
}

T FN(orc_dot_public)(const T *a, const T *b, int64_t n) { return FN(orc_dot)(a, b, n); }

/* LinearAlgebra.norm (2-norm) = sqrt(dot(x,x)) up to BLAS nrm2 scaling tricks. */
static T FN(orc_norm)(const T *a, int64_t n) {
  T s = FN(orc_dot)(a, a, n);
  return sizeof(T) == 8 ? (T)sqrt((double)s) : (T)sqrtf((float)s);
}

/* ---- mulSquareOpDiagonal! / mulOpDiagonal! — src/special-operators.jl:125-131,144-151 ---- */
void FN(orc_diag_mul)(T *res, const T *d, const T *v, int64_t n_min, int64_t nrow, double alpha,
                      double beta, int32_t flags) {
  const int64_t ds = (flags & ORC_D_SCALAR) ? 0 : 1; /* 1-element d broadcasts */
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                                      \
    if (beta == 0) { /* :126-127  res .= α .* d .* v   => (α*d)*v */                            \
      for (int64_t i = 0; i < n_min; ++i) res[i] = (T)((a * (CA)d[i * ds]) * (CA)v[i]);          \
    } else { /* :128-129  res .= α .* d .* v .+ β .* res */                                     \
      for (int64_t i = 0; i < n_min; ++i)                                                        \
        res[i] = (T)(((a * (CA)d[i * ds]) * (CA)v[i]) + (b * (CB)res[i]));                       \
    }                                                                                            \
  }
  WITH_CT(BODY);
#undef BODY
  for (int64_t i = n_min; i < nrow; ++i) res[i] = 0; /* :150  res[n_min+1:end] .= 0 */
}

/* ---- mulOpEye! — src/special-operators.jl:36-44 ---- */
void FN(orc_eye_mul)(T *res, const T *v, int64_t n_min, int64_t nrow, double alpha, double beta,
                     int32_t flags) {
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                                      \
    if (beta == 0) { /* :38-39 */                                                                \
      for (int64_t i = 0; i < n_min; ++i) res[i] = (T)(a * (CA)v[i]);                            \
      for (int64_t i = n_min; i < nrow; ++i) res[i] = 0;                                         \
    } else { /* :41-42 — the tail receives β itself (reference quirk) unless the               \
                generic-axpby flag asks for β*res */                                             \
      for (int64_t i = 0; i < n_min; ++i) res[i] = (T)((a * (CA)v[i]) + (b * (CB)res[i]));       \
      if (flags & ORC_TAIL_BETA)                                                                 \
        for (int64_t i = n_min; i < nrow; ++i) res[i] = (T)b;                                    \
      else                                                                                       \
        for (int64_t i = n_min; i < nrow; ++i) res[i] = (T)(b * (CB)res[i]);                     \
    }                                                                                            \
  }
  WITH_CT(BODY);
#undef BODY
}

/* ---- mulOpZeros! — src/special-operators.jl:102-108 ---- */
void FN(orc_zeros_mul)(T *res, int64_t nrow, double beta, int32_t flags) {
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CB b = (CB)beta;                                                                       \
    if (beta == 0)                                                                               \
      for (int64_t i = 0; i < nrow; ++i) res[i] = 0; /* :104 */                                  \
    else                                                                                         \
      for (int64_t i = 0; i < nrow; ++i) res[i] = (T)((CB)res[i] * b); /* :106 res .*= β */      \
  }
  WITH_CT(BODY);
#undef BODY
}

/* Base.sum on a Vector: pairwise with 1024-element leaves (base/reduce.jl
 * mapreduce_impl); the leaf loop is @simd in Julia (reassociable), so the value
 * is pinned to tolerance only. */
static T FN(orc_sum_pairwise)(const T *v, int64_t lo, int64_t hi) {
  if (hi - lo <= 1024) {
    T s = 0;
    for (int64_t i = lo; i < hi; ++i) s += v[i];
    return s;
  }
  int64_t mid = lo + (hi - lo) / 2;
  return FN(orc_sum_pairwise)(v, lo, mid) + FN(orc_sum_pairwise)(v, mid, hi);
}

/* ---- mulOpOnes! — src/special-operators.jl:79-85 ---- */
void FN(orc_ones_mul)(T *res, int64_t nrow, const T *v, int64_t ncol, double alpha, double beta,
                      int32_t flags) {
  const T sv = FN(orc_sum_pairwise)(v, 0, ncol);
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                                      \
    const CA as = a * (CA)sv; /* (α * sum(v)) */                                                 \
    if (beta == 0)                                                                               \
      for (int64_t i = 0; i < nrow; ++i) res[i] = (T)as;                                         \
    else                                                                                         \
      for (int64_t i = 0; i < nrow; ++i) res[i] = (T)(as + (b * (CB)res[i]));                    \
  }
  WITH_CT(BODY);
#undef BODY
}

/* ---- prod3! `res .*= α` — src/operations.jl:13-15 ---- */
void FN(orc_scale)(T *res, int64_t n, double alpha, int32_t flags) {
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha;                                                                      \
    for (int64_t i = 0; i < n; ++i) res[i] = (T)((CA)res[i] * a);                                \
  }
  WITH_CT(BODY);
#undef BODY
}

/* ---- mulHouseholder! — src/linalg.jl:77-83 ---- */
void FN(orc_householder_mul)(T *res, const T *h, const T *v, int64_t n, double alpha,
                             double beta, int32_t flags) {
  const T c = (T)2 * FN(orc_dot)(h, v, n); /* 2 * dot(h, v) : scalar, type T */
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                                      \
    if (beta == 0) { /* :79  res .= α .* (v .- 2 * dot(h, v) .* h) */                           \
      for (int64_t i = 0; i < n; ++i) res[i] = (T)(a * (CA)(T)(v[i] - (c * h[i])));              \
    } else { /* :81 */                                                                           \
      for (int64_t i = 0; i < n; ++i)                                                            \
        res[i] = (T)((a * (CA)(T)(v[i] - (c * h[i]))) + (b * (CB)res[i]));                       \
    }                                                                                            \
  }
  WITH_CT(BODY);
#undef BODY
}

/* ---- mulHermitian! — src/linalg.jl:97-103 with L = tril(A,-1) (:112) ----
 * res .= α .* (d .* v .+ L * v .+ (v' * L)')[:] (.+ β .* res)
 * `L*v` and `v'*L` are BLAS gemv in the reference (order unpinned). A is the
 * original column-major n x n matrix; only i > j entries are used. t1,t2 are
 * caller scratch of length n. */
void FN(orc_hermitian_mul)(T *res, const T *d, const T *A, int64_t lda, const T *v, int64_t n,
                           double alpha, double beta, int32_t flags, T *t1, T *t2) {
  for (int64_t i = 0; i < n; ++i) t1[i] = 0;
  for (int64_t j = 0; j < n; ++j) { /* L*v, column sweep (gemv 'N') */
    const T vj = v[j];
    const T *col = A + j * lda;
    for (int64_t i = j + 1; i < n; ++i) t1[i] += col[i] * vj;
  }
  for (int64_t j = 0; j < n; ++j) { /* (v'*L)' : t2[j] = sum_{i>j} L[i,j]*v[i] (gemv 'T') */
    const T *col = A + j * lda;
    T s = 0;
    for (int64_t i = j + 1; i < n; ++i) s += col[i] * v[i];
    t2[j] = s;
  }
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                                      \
    for (int64_t i = 0; i < n; ++i) {                                                            \
      const T inner = ((d[i] * v[i]) + t1[i]) + t2[i]; /* d.*v .+ L*v .+ (v'L)' */               \
      if (beta == 0)                                                                             \
        res[i] = (T)(a * (CA)inner);                                                             \
      else                                                                                       \
        res[i] = (T)((a * (CA)inner) + (b * (CB)res[i]));                                        \
    }                                                                                            \
  }
  WITH_CT(BODY);
#undef BODY
}

/* ---- mulRestrict! / multRestrict! — src/special-operators.jl:167-174 ----
 * α, β ignored by the reference. idx is 1-based as Julia stores it. */
void FN(orc_restrict)(T *res, const T *v, const int64_t *idx, int64_t nidx) {
  for (int64_t k = 0; k < nidx; ++k) res[k] = v[idx[k] - 1]; /* res .= view(v, I) */
}
void FN(orc_extend)(T *res, int64_t nres, const T *u, const int64_t *idx, int64_t nidx) {
  for (int64_t i = 0; i < nres; ++i) res[i] = 0;             /* res .= 0     */
  for (int64_t k = 0; k < nidx; ++k) res[idx[k] - 1] = u[k]; /* res[I] = u : last write wins */
}

/* ---- dense LinearOperator(M) — src/constructors.jl:19-29 (LinearAlgebra.mul!) ---- */
void FN(orc_gemv)(T *res, const T *M, int64_t m, int64_t n, int64_t ld, const T *v,
                  double alpha, double beta, int32_t trans, int32_t flags, T *tmp) {
  /* tmp (length of res) = M*v or M'*v, then res = α*tmp + β*res (gemv semantics) */
  const int64_t nr = trans ? n : m;
  if (!trans) {
    for (int64_t i = 0; i < m; ++i) tmp[i] = 0;
    for (int64_t j = 0; j < n; ++j) {
      const T vj = v[j];
      for (int64_t i = 0; i < m; ++i) tmp[i] += M[i + j * ld] * vj;
    }
  } else {
    for (int64_t j = 0; j < n; ++j) {
      T s = 0;
      for (int64_t i = 0; i < m; ++i) s += M[i + j * ld] * v[i];
      tmp[j] = s;
    }
  }
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                                      \
    for (int64_t i = 0; i < nr; ++i)                                                             \
      res[i] = (beta == 0) ? (T)(a * (CA)tmp[i]) : (T)((a * (CA)tmp[i]) + (b * (CB)res[i]));     \
  }
  WITH_CT(BODY);
#undef BODY
}

/* ---- sparse LinearOperator(M::SparseMatrixCSC) — src/constructors.jl:19-29 calls LinearAlgebra.mul!(res, M, v, α, β) ----
 * For a SparseMatrixCSC that method lives in the SparseArrays stdlib, which is NOT part of /root/reference (a Julia
 * standard library, version = the Julia the package runs on, >= 1.10 by Project.toml). Its published algorithm
 * (stdlib/SparseArrays/src/linalg.jl), restated:
 *   _spmatmul!(C, A, B, α, β):        β != 1 && (β == 0 ? fill!(C, 0) : rmul!(C, β))      [_rmul_or_fill!]
 *                                     for col = 1:n:  αxj = B[col] * α
 *                                       for k in nzrange(A, col):  C[rowval[k]] += nzval[k] * αxj
 *   _At_or_Ac_mul_B!(tfun, C, A, B, α, β):   the same β step, then
 *                                     for col = 1:n:  tmp = zero(eltype(C))
 *                                       for k in nzrange(A, col):  tmp += tfun(nzval[k]) * B[rowval[k]]
 *                                       C[col] += tmp * α
 * PARITY UNPINNED by golden vectors: the reference's tests only compare such operators with dense matrices to
 * sqrt(eps) (test/test_linop.jl:743-756, test/test_kron.jl:3-36); tests/test_oracle_kat.py pins this restatement the same
 * way (against an independent dense product). colptr / rowval are 1-based as Julia stores them. trans: 0 = A*v, 1 = Aᵀ*v
 * (= A'*v for the real element types instantiated here). */
void FN(orc_csc_mul)(T *res, const int64_t *colptr, const int64_t *rowval, const T *nzval, int64_t m, int64_t n,
                     const T *v, double alpha, double beta, int32_t trans, int32_t flags) {
  const int64_t nr = trans ? n : m;
#define BODY(CA, CB)                                                                             \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                               \
    if (b != (CB)1) {                                                                            \
      if (b == (CB)0) { for (int64_t i = 0; i < nr; ++i) res[i] = (T)0; }                        \
      else { for (int64_t i = 0; i < nr; ++i) res[i] = (T)((CB)res[i] * b); }                    \
    }                                                                                            \
    if (!trans) {                                                                                \
      for (int64_t col = 0; col < n; ++col) {                                                    \
        const CA axj = (CA)v[col] * a;                                                           \
        for (int64_t k = colptr[col] - 1; k < colptr[col + 1] - 1; ++k)                          \
          res[rowval[k] - 1] = (T)((CA)res[rowval[k] - 1] + (CA)nzval[k] * axj);                 \
      }                                                                                          \
    } else {                                                                                     \
      for (int64_t col = 0; col < n; ++col) {                                                    \
        T tmp = (T)0;                                                                            \
        for (int64_t k = colptr[col] - 1; k < colptr[col + 1] - 1; ++k)                          \
          tmp = (T)(tmp + nzval[k] * v[rowval[k] - 1]);                                          \
        res[col] = (T)((CA)res[col] + (CA)tmp * a);                                              \
      }                                                                                          \
    }                                                                                            \
  }
  WITH_CT(BODY);
#undef BODY
}

/* ---- kron(A,B) prod!/tprod! — src/kron.jl:14-31 ----
 * prod!:  X = reshape(x, q, n); res .= α .* Matrix(B * X * transpose(A))[:] (.+ β .* res)
 * `B * X * transpose(A)` is a composite operator (src/operations.jl:131-156,160)
 * and Matrix(·) applies it to the m unit vectors (src/abstract.jl:282-292):
 *   column i = B * (X * (transpose(A) * e_i)) = B * (X * A[i, :]).
 * tprod!: X = reshape(x, p, m); column i = transpose(B) * (X * (A * e_i)) = Bᵀ (X A[:, i]).
 * work holds the intermediate (max(q,p) elements) and the result matrix. */
void FN(orc_kron_mul)(T *res, const T *A, int64_t m, int64_t n, int64_t lda, const T *B,
                      int64_t p, int64_t q, int64_t ldb, const T *x, double alpha, double beta,
                      int32_t trans, int32_t flags, T *work) {
  const int64_t ncols_out = trans ? n : m;  /* columns of the materialised matrix */
  const int64_t nrows_out = trans ? q : p;  /* its rows */
  const int64_t xr = trans ? p : q;         /* rows of X */
  const int64_t xc = trans ? m : n;         /* cols of X */
  T *u = work;                              /* X * w           (xr) */
  T *R = work + xr;                         /* result matrix   (nrows_out * ncols_out) */
  for (int64_t i = 0; i < ncols_out; ++i) {
    /* w = transpose(A) * e_i = A[i, :]  (prod!)  |  A * e_i = A[:, i]  (tprod!) */
    for (int64_t r = 0; r < xr; ++r) u[r] = 0;
    for (int64_t j = 0; j < xc; ++j) {
      const T wj = trans ? A[j + i * lda] : A[i + j * lda];
      for (int64_t r = 0; r < xr; ++r) u[r] += x[r + j * xr] * wj;
    }
    T *col = R + i * nrows_out;
    if (!trans) { /* B * u */
      for (int64_t r = 0; r < p; ++r) col[r] = 0;
      for (int64_t j = 0; j < q; ++j) {
        const T uj = u[j];
        for (int64_t r = 0; r < p; ++r) col[r] += B[r + j * ldb] * uj;
      }
    } else { /* transpose(B) * u */
      for (int64_t j = 0; j < q; ++j) {
        T s = 0;
        for (int64_t r = 0; r < p; ++r) s += B[r + j * ldb] * u[r];
        col[j] = s;
      }
    }
  }
  const int64_t nout = nrows_out * ncols_out;
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                                      \
    for (int64_t i = 0; i < nout; ++i)                                                           \
      res[i] = (beta == 0) ? (T)(a * (CA)R[i]) : (T)((a * (CA)R[i]) + (b * (CB)res[i]));         \
  }
  WITH_CT(BODY);
#undef BODY
}

/* ======================================================================== */
/*  L-BFGS — src/lbfgs.jl                                                    */
/* ======================================================================== */

/* LBFGSData — src/lbfgs.jl:4-24. Vector{Vector{T}} fields are stored as
 * mem x n row-major arrays (slot k at base + k*n); `insert` is 1-BASED as in
 * Julia so the index arithmetic below can be copied verbatim. */
typedef struct {
  int64_t n, mem;
  int32_t scaling, damped, inverse, pad_;
  T scaling_factor, sigma2, sigma3, opnorm_upper_bound;
  T *s, *y, *ys, *alpha, *a, *b, *norm_b;
  int64_t insert;
  T *Ax;        /* n           */
  T *shifted_p; /* n x 2mem column-major (may be NULL if solve is never called) */
  T *shifted_v; /* 2mem        */
  T *shifted_u; /* n           */
} FN(orc_lbfgs);

static inline int64_t FN(jmod)(int64_t a, int64_t m) { /* Julia mod: result in [0,m) */
  int64_t r = a % m;
  return r < 0 ? r + m : r;
}

/* InverseLBFGSOperator lbfgs_multiply — src/lbfgs.jl:117-154 */
void FN(orc_lbfgs_inv_mul)(FN(orc_lbfgs) * data, T *res, const T *x, double alpha_m,
                           double beta_m, int32_t flags) {
  const int64_t n = data->n, mem = data->mem;
  T *q = data->Ax;                              /* :127 */
  for (int64_t j = 0; j < n; ++j) q[j] = x[j];  /* :128 q .= x */
  for (int64_t i = 1; i <= mem; ++i) {          /* :130 */
    const int64_t k = FN(jmod)(data->insert - i - 1, mem) + 1; /* :131 */
    if (data->ys[k - 1] != 0) {                 /* :132 */
      const T *sk = data->s + (k - 1) * n, *yk = data->y + (k - 1) * n;
      const T ak = FN(orc_dot)(sk, q, n) / data->ys[k - 1]; /* :133 */
      data->alpha[k - 1] = ak;                  /* :134 */
      for (int64_t j = 0; j < n; ++j) q[j] = q[j] - (ak * yk[j]); /* :135 q .-= αk .* y[k] */
    }
  }
  if (data->scaling)                            /* :139 q .*= scaling_factor */
    for (int64_t j = 0; j < n; ++j) q[j] = q[j] * data->scaling_factor;
  for (int64_t i = 1; i <= mem; ++i) {          /* :141 */
    const int64_t k = FN(jmod)(data->insert + i - 2, mem) + 1; /* :142 */
    if (data->ys[k - 1] != 0) {                 /* :143 */
      const T *sk = data->s + (k - 1) * n, *yk = data->y + (k - 1) * n;
      const T ak = data->alpha[k - 1];          /* :144 */
      const T bt = ak - (FN(orc_dot)(yk, q, n) / data->ys[k - 1]); /* :145 */
      for (int64_t j = 0; j < n; ++j) q[j] = q[j] + (bt * sk[j]);  /* :146 q .+= β .* s[k] */
    }
  }
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha_m; const CB b = (CB)beta_m;                                                  \
    if (beta_m == 0) /* :150 res .= αm .* q */                                                   \
      for (int64_t j = 0; j < n; ++j) res[j] = (T)(a * (CA)q[j]);                                \
    else /* :152 */                                                                              \
      for (int64_t j = 0; j < n; ++j) res[j] = (T)((a * (CA)q[j]) + (b * (CB)res[j]));           \
  }
  WITH_CT(BODY);
#undef BODY
}

/* LBFGSOperator lbfgs_multiply (forward) — src/lbfgs.jl:173-202 */
void FN(orc_lbfgs_fwd_mul)(FN(orc_lbfgs) * data, T *res, const T *x, double alpha, double beta,
                           int32_t flags) {
  const int64_t n = data->n, mem = data->mem;
  T *q = data->Ax;                              /* :183 */
  for (int64_t j = 0; j < n; ++j) q[j] = x[j];  /* :184 */
  if (data->scaling)                            /* :186 q ./= scaling_factor */
    for (int64_t j = 0; j < n; ++j) q[j] = q[j] / data->scaling_factor;
  for (int64_t i = 1; i <= mem; ++i) {          /* :189 */
    const int64_t k = FN(jmod)(data->insert + i - 2, mem) + 1; /* :190 */
    if (data->ys[k - 1] != 0) {                 /* :191 */
      const T *ak = data->a + (k - 1) * n, *bk = data->b + (k - 1) * n;
      const T ax = FN(orc_dot)(ak, x, n);       /* :192 */
      const T bx = FN(orc_dot)(bk, x, n);       /* :193 */
      for (int64_t j = 0; j < n; ++j)           /* :194 q .+= bx .* b[k] .- ax .* a[k] */
        q[j] = q[j] + ((bx * bk[j]) - (ax * ak[j]));
    }
  }
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                                      \
    if (beta == 0) /* :198 */                                                                    \
      for (int64_t j = 0; j < n; ++j) res[j] = (T)(a * (CA)q[j]);                                \
    else /* :200 */                                                                              \
      for (int64_t j = 0; j < n; ++j) res[j] = (T)((a * (CA)q[j]) + (b * (CB)res[j]));           \
  }
  WITH_CT(BODY);
#undef BODY
}

static inline T FN(orc_sqrt)(T v) { return sizeof(T) == 8 ? (T)sqrt((double)v) : (T)sqrtf((float)v); }
static inline T FN(orc_eps)(void) { return sizeof(T) == 8 ? (T)2.220446049250313e-16 : (T)1.1920929e-07f; }

/* push_common! — src/lbfgs.jl:210-255 */
void FN(orc_lbfgs_push_common)(FN(orc_lbfgs) * data, const T *s, const T *y, T ys) {
  const int64_t n = data->n, mem = data->mem;
  const int64_t insert = data->insert;                        /* :218 */
  T *si = data->s + (insert - 1) * n, *yi = data->y + (insert - 1) * n;
  for (int64_t j = 0; j < n; ++j) si[j] = s[j];               /* :220 */
  for (int64_t j = 0; j < n; ++j) yi[j] = y[j];               /* :221 */
  data->ys[insert - 1] = ys;                                  /* :222 */
  if (data->scaling) {                                        /* :223 */
    if (data->scaling_factor != 0) data->opnorm_upper_bound -= (T)1 / data->scaling_factor; /* :224 */
    data->scaling_factor = ys / FN(orc_dot)(y, y, n);         /* :225 */
    if (data->scaling_factor != 0) data->opnorm_upper_bound += (T)1 / data->scaling_factor; /* :226 */
  }
  if (!data->inverse) {                                       /* :230 */
    T *bi = data->b + (insert - 1) * n;
    data->opnorm_upper_bound -= data->norm_b[insert - 1] * data->norm_b[insert - 1]; /* :231 */
    const T sq = FN(orc_sqrt)(ys);
    for (int64_t j = 0; j < n; ++j) bi[j] = y[j] / sq;        /* :232 b[insert] .= y ./ sqrt(ys) */
    data->norm_b[insert - 1] = FN(orc_norm)(bi, n);           /* :233 */
    data->opnorm_upper_bound += data->norm_b[insert - 1] * data->norm_b[insert - 1]; /* :234 */
    for (int64_t i = 1; i <= mem; ++i) {                      /* :236 */
      const int64_t k = FN(jmod)(insert + i - 1, mem) + 1;    /* :237 */
      if (data->ys[k - 1] != 0) {                             /* :238 */
        T *ak = data->a + (k - 1) * n;
        const T *sk = data->s + (k - 1) * n;
        for (int64_t j = 0; j < n; ++j) ak[j] = sk[j] / data->scaling_factor; /* :239 */
        for (int64_t jj = 1; jj <= i - 1; ++jj) {             /* :241 */
          const int64_t l = FN(jmod)(insert + jj - 1, mem) + 1; /* :242 */
          if (data->ys[l - 1] != 0) {                         /* :243 */
            const T *bl = data->b + (l - 1) * n, *al = data->a + (l - 1) * n;
            const T c1 = FN(orc_dot)(bl, sk, n);
            for (int64_t j = 0; j < n; ++j) ak[j] = ak[j] + (c1 * bl[j]); /* :244 */
            const T c2 = FN(orc_dot)(al, sk, n);
            for (int64_t j = 0; j < n; ++j) ak[j] = ak[j] - (c2 * al[j]); /* :245 */
          }
        }
        const T dn = FN(orc_sqrt)(FN(orc_dot)(sk, ak, n));
        for (int64_t j = 0; j < n; ++j) ak[j] = ak[j] / dn;   /* :248 */
      }
    }
  }
  data->insert = FN(jmod)(insert, mem) + 1;                   /* :253 */
}

/* push!(op, s, y) — src/lbfgs.jl:269-287 (undamped). Returns 1 if accepted. */
int32_t FN(orc_lbfgs_push)(FN(orc_lbfgs) * data, const T *s, const T *y) {
  const T ys = FN(orc_dot)(y, s, data->n);  /* :277 */
  if (ys <= FN(orc_eps)()) return 0;        /* :281-284 */
  FN(orc_lbfgs_push_common)(data, s, y, ys);
  return 1;
}

/* push!(op, s, y, Bs) forward damped — src/lbfgs.jl:289-323. ytmp: scratch n. */
int32_t FN(orc_lbfgs_push_damped_fwd)(FN(orc_lbfgs) * data, const T *s, const T *y, T *Bs,
                                      T *ytmp) {
  const int64_t n = data->n;
  T ys = FN(orc_dot)(y, s, n);                                 /* :300 */
  const T s2 = data->sigma2, s3 = data->sigma3;
  FN(orc_lbfgs_fwd_mul)(data, Bs, s, 1.0, 0.0, 0);             /* :305 mul!(Bs, op, s, one, zero) */
  const T sBs = FN(orc_dot)(s, Bs, n);                         /* :306 */
  int damp = 0;
  T theta = 0;
  if (ys < ((T)1 - s2) * sBs) {                                /* :308 */
    theta = s2 * sBs / (sBs - ys);
    damp = 1;
  } else if (ys > ((T)1 + s3) * sBs) {                         /* :311 */
    theta = s3 * sBs / (ys - sBs);
    damp = 1;
  }
  const T *yy = y;
  if (damp) {                                                  /* :315-318 (y is rebound, not mutated) */
    for (int64_t j = 0; j < n; ++j) ytmp[j] = (theta * y[j]) + (((T)1 - theta) * Bs[j]);
    ys = (theta * ys) + (((T)1 - theta) * sBs);
    yy = ytmp;
  }
  FN(orc_lbfgs_push_common)(data, s, yy, ys);                  /* :320 — note: no ys<=eps rejection here */
  return 1;
}

/* push!(op, s, y, α, g, Bs) inverse damped — src/lbfgs.jl:325-357. Mutates y (:351). */
int32_t FN(orc_lbfgs_push_damped_inv)(FN(orc_lbfgs) * data, const T *s, T *y, T alpha,
                                      const T *g, T *Bs) {
  const int64_t n = data->n;
  T ys = FN(orc_dot)(y, s, n);                                 /* :336 */
  const T s2 = data->sigma2, s3 = data->sigma3;
  for (int64_t j = 0; j < n; ++j) Bs[j] = (-alpha) * g[j];     /* :341 Bs .= -α .* g */
  const T sBs = FN(orc_dot)(s, Bs, n);                         /* :342 */
  int damp = 0;
  T theta = 0;
  if (ys < ((T)1 - s2) * sBs) {
    theta = s2 * sBs / (sBs - ys);
    damp = 1;
  } else if (ys > ((T)1 + s3) * sBs) {
    theta = s3 * sBs / (ys - sBs);
    damp = 1;
  }
  if (damp) {                                                  /* :350-353 */
    for (int64_t j = 0; j < n; ++j) y[j] = (theta * y[j]) + (((T)1 - theta) * Bs[j]);
    ys = (theta * ys) + (((T)1 - theta) * sBs);
  }
  FN(orc_lbfgs_push_common)(data, s, y, ys);
  return 1;
}

/* diag!(op, d) — src/lbfgs.jl:379-395 (forward only) */
void FN(orc_lbfgs_diag)(FN(orc_lbfgs) * data, T *d) {
  const int64_t n = data->n, mem = data->mem;
  for (int64_t j = 0; j < n; ++j) d[j] = 1;                    /* :385 */
  if (data->scaling)
    for (int64_t j = 0; j < n; ++j) d[j] = d[j] / data->scaling_factor; /* :386 */
  for (int64_t i = 1; i <= mem; ++i) {                         /* :388 */
    const int64_t k = FN(jmod)(data->insert + i - 2, mem) + 1;
    if (data->ys[k - 1] != 0) {
      const T *ak = data->a + (k - 1) * n, *bk = data->b + (k - 1) * n;
      for (int64_t j = 0; j < n; ++j) d[j] = d[j] + ((bk[j] * bk[j]) - (ak[j] * ak[j])); /* :391 */
    }
  }
}

/* reset!(data, inverse) — src/lbfgs.jl:401-415 */
void FN(orc_lbfgs_reset)(FN(orc_lbfgs) * data) {
  const int64_t n = data->n, mem = data->mem;
  for (int64_t j = 0; j < mem * n; ++j) data->s[j] = 0, data->y[j] = 0;
  if (!data->inverse)
    for (int64_t j = 0; j < mem * n; ++j) data->a[j] = 0, data->b[j] = 0;
  for (int64_t k = 0; k < mem; ++k) data->ys[k] = 0;
  if (data->inverse)
    for (int64_t k = 0; k < mem; ++k) data->alpha[k] = 0;
  data->scaling_factor = 1;
  data->insert = 1;
}

/* solve_shifted_system!(x, B, b, σ) — src/utilities.jl:207-248.
 * Returns 0, or 1 if σ < 0 (reference throws ArgumentError, :213-215). */
int32_t FN(orc_solve_shifted)(FN(orc_lbfgs) * data, T *x, const T *b, T sigma) {
  if (sigma < 0) return 1;
  const int64_t n = data->n, mem = data->mem;
  const int64_t insert = data->insert;                         /* :217 */
  const T g_inv = (T)1 / data->scaling_factor;                 /* :219 */
  const T x0 = (T)1 / (g_inv + sigma);                         /* :220 */
  for (int64_t j = 0; j < n; ++j) x[j] = x0 * b[j];            /* :221 */
  const int64_t max_i = 2 * mem;                               /* :223 */
  int sign_i = 1;
  T *u = data->shifted_u;
  for (int64_t i = 1; i <= max_i; ++i) {                       /* :226 */
    const int64_t jj = (i + 1) / 2;                            /* :227 */
    const int64_t k = FN(jmod)(insert + jj - 1, mem) + 1;      /* :228 */
    const T *src = (sign_i == -1) ? data->b + (k - 1) * n : data->a + (k - 1) * n; /* :229 */
    for (int64_t j = 0; j < n; ++j) u[j] = src[j];
    T *pi = data->shifted_p + (i - 1) * n;
    for (int64_t j = 0; j < n; ++j) pi[j] = x0 * u[j];         /* :231 */
    int sign_t = 1;
    for (int64_t t = 1; t <= i - 1; ++t) {                     /* :234 */
      const T *pt = data->shifted_p + (t - 1) * n;
      const T c0 = FN(orc_dot)(pt, u, n);                      /* :235 */
      const T c1 = (T)sign_t * data->shifted_v[t - 1];         /* :236 */
      const T c2 = c1 * c0;                                    /* :237 */
      for (int64_t j = 0; j < n; ++j) pi[j] = pi[j] + (c2 * pt[j]); /* :238 */
      sign_t = -sign_t;
    }
    data->shifted_v[i - 1] = (T)1 / ((T)1 - ((T)sign_i * FN(orc_dot)(u, pi, n))); /* :242 */
    const T coef = ((T)sign_i * data->shifted_v[i - 1]) * FN(orc_dot)(pi, b, n);  /* :243-244 */
    for (int64_t j = 0; j < n; ++j) x[j] = x[j] + (coef * pi[j]);
    sign_i = -sign_i;                                          /* :245 */
  }
  return 0;
}

/* ======================================================================== */
/*  L-SR1 — src/lsr1.jl                                                      */
/* ======================================================================== */
typedef struct {
  int64_t n, mem;
  int32_t scaling, pad_;
  T scaling_factor, opnorm_upper_bound;
  T *s, *y, *ys, *a, *as;
  int64_t insert;
  T *Ax, *tmp;
} FN(orc_lsr1);

/* lsr1_multiply — src/lsr1.jl:89-107 */
void FN(orc_lsr1_mul)(FN(orc_lsr1) * data, T *q, const T *x, double alpha, double beta,
                      int32_t flags) {
  const int64_t n = data->n, mem = data->mem;
#define BODY(CA, CB)                                                                                 \
  {                                                                                              \
    const CA a = (CA)alpha; const CB b = (CB)beta;                                                      \
    const CA g = (CA)data->scaling_factor;                                                       \
    if (beta == 0) /* :93 q .= α .* x ./ scaling_factor  => (α*x)/γ */                           \
      for (int64_t j = 0; j < n; ++j) q[j] = (T)((a * (CA)x[j]) / g);                            \
    else /* :95 */                                                                               \
      for (int64_t j = 0; j < n; ++j) q[j] = (T)(((a * (CA)x[j]) / g) + (b * (CB)q[j]));         \
    for (int64_t i = 1; i <= mem; ++i) { /* :98 */                                               \
      const int64_t k = FN(jmod)(data->insert + i - 2, mem) + 1; /* :99 */                       \
      if (data->ys[k - 1] != 0) { /* :100 */                                                     \
        const T *ak = data->a + (k - 1) * n;                                                     \
        const CA ax = (a * (CA)FN(orc_dot)(ak, x, n)) / (CA)data->as[k - 1]; /* :101 */          \
        for (int64_t j = 0; j < n; ++j) q[j] = (T)((CA)q[j] + (ax * (CA)ak[j])); /* :103 */      \
      }                                                                                          \
    }                                                                                            \
  }
  WITH_CT(BODY);
#undef BODY
}

static inline T FN(orc_abs)(T v) { return v < 0 ? -v : v; }

/* push!(op::LSR1Operator, s, y) — src/lsr1.jl:119-184. Returns 1 if accepted. */
int32_t FN(orc_lsr1_push)(FN(orc_lsr1) * data, const T *s, const T *y) {
  const int64_t n = data->n, mem = data->mem;
  T *ymBs = data->tmp;                                         /* :123 */
  for (int64_t j = 0; j < n; ++j) ymBs[j] = y[j];              /* :124 */
  /* :125 mul!(ymBs, op, s, -1, 1): α=-1, β=1 are Int literals; Int*T promotes to T */
  FN(orc_lsr1_mul)(data, ymBs, s, -1.0, 1.0, 0);
  const T ys = FN(orc_dot)(y, s, n);                           /* :126 */
  const T sNorm = FN(orc_norm)(s, n);                          /* :127 */
  const T yy = FN(orc_dot)(y, y, n);                           /* :128 */
  const T eps = FN(orc_eps)();                                 /* :130 */
  const int well_defined =
      FN(orc_abs)(FN(orc_dot)(ymBs, s, n)) >= eps + eps * FN(orc_norm)(ymBs, n) * sNorm; /* :131 */
  int sufficient_curvature = 1, scaling_condition = 1;
  if (data->scaling) {                                         /* :135 */
    const T yNorm = FN(orc_sqrt)(yy);                          /* :136 */
    sufficient_curvature = FN(orc_abs)(ys) >= eps * yNorm * sNorm; /* :137 */
    if (sufficient_curvature) {
      const T sf = ys / yy;                                    /* :139 */
      for (int64_t j = 0; j < n; ++j) data->tmp[j] = y[j] - (s[j] / sf); /* :140 */
      scaling_condition = FN(orc_norm)(data->tmp, n) >= eps * yNorm * sNorm; /* :141 */
    }
  }
  if (!(well_defined && sufficient_curvature && scaling_condition)) return 0; /* :145-149 */
  T *si = data->s + (data->insert - 1) * n, *yi = data->y + (data->insert - 1) * n;
  for (int64_t j = 0; j < n; ++j) si[j] = s[j];                /* :151 */
  for (int64_t j = 0; j < n; ++j) yi[j] = y[j];                /* :152 */
  data->ys[data->insert - 1] = ys;                             /* :153 */
  data->opnorm_upper_bound = 1;                                /* :156 */
  if (data->scaling) {                                         /* :157 */
    data->scaling_factor = ys / yy;                            /* :158 */
    if (data->scaling_factor != 0)
      data->opnorm_upper_bound = (T)1 / FN(orc_abs)(data->scaling_factor); /* :159 */
  }
  data->insert = FN(jmod)(data->insert, mem) + 1;              /* :163 */
  for (int64_t i = 1; i <= mem; ++i) {                         /* :166 */
    const int64_t k = FN(jmod)(data->insert + i - 2, mem) + 1; /* :167 */
    if (data->ys[k - 1] != 0) {
      T *ak = data->a + (k - 1) * n;
      const T *sk = data->s + (k - 1) * n, *yk = data->y + (k - 1) * n;
      for (int64_t j = 0; j < n; ++j) ak[j] = yk[j] - (sk[j] / data->scaling_factor); /* :169 */
      for (int64_t jj = 1; jj <= i - 1; ++jj) {                /* :170 */
        const int64_t l = FN(jmod)(data->insert + jj - 2, mem) + 1; /* :171 */
        if (data->ys[l - 1] != 0) {
          const T *al = data->a + (l - 1) * n;
          const T as = FN(orc_dot)(al, sk, n) / data->as[l - 1]; /* :173 */
          for (int64_t j = 0; j < n; ++j) ak[j] = ak[j] - (as * al[j]); /* :174 */
        }
      }
      data->as[k - 1] = FN(orc_dot)(ak, sk, n);                /* :177 */
      if (data->as[k - 1] != 0) {                              /* :179 */
        const T na = FN(orc_norm)(ak, n);
        data->opnorm_upper_bound += (na * na) / FN(orc_abs)(data->as[k - 1]);
      }
    }
  }
  return 1;
}

/* diag!(op::LSR1Operator, d) — src/lsr1.jl:196-211 */
void FN(orc_lsr1_diag)(FN(orc_lsr1) * data, T *d) {
  const int64_t n = data->n, mem = data->mem;
  for (int64_t j = 0; j < n; ++j) d[j] = 1;
  if (data->scaling)
    for (int64_t j = 0; j < n; ++j) d[j] = d[j] / data->scaling_factor;
  for (int64_t i = 1; i <= mem; ++i) {
    const int64_t k = FN(jmod)(data->insert + i - 2, mem) + 1;
    if (data->ys[k - 1] != 0) {
      const T *ak = data->a + (k - 1) * n;
      const T as = data->as[k - 1];
      for (int64_t j = 0; j < n; ++j) d[j] = d[j] + ((ak[j] * ak[j]) / as); /* :206 */
    }
  }
}

/* reset!(data::LSR1Data) — src/lsr1.jl:217-228 */
void FN(orc_lsr1_reset)(FN(orc_lsr1) * data) {
  const int64_t n = data->n, mem = data->mem;
  for (int64_t j = 0; j < mem * n; ++j) data->s[j] = 0, data->y[j] = 0, data->a[j] = 0;
  for (int64_t k = 0; k < mem; ++k) data->ys[k] = 0, data->as[k] = 0;
  data->scaling_factor = 1;
  data->insert = 1;
}

#undef FN
#undef CAT
#undef CAT_
#undef WITH_CT
