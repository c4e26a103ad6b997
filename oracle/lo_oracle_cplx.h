/* lo_oracle_cplx.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Complex{R} restatement of the elementwise leaves and opHouseholder of LinearOperators.jl v2.14.2, included by
 * lo_oracle.c with R = double / float (SUF = c64 / c32). Vectors are interleaved (re, im) arrays exactly like
 * Julia's Vector{Complex{R}}. Arithmetic follows base/complex.jl component by component, no FMA
 * (-ffp-contract=off):
 *     *(z::Complex, w::Complex) = Complex(zr*wr - zi*wi, zr*wi + zi*wr)
 *     *(x::Real,    w::Complex) = Complex(x*wr, x*wi)          +,- : componentwise
 * Caller scalars arrive as (re, im) doubles; ORC_ALPHA_REAL / ORC_BETA_REAL mark Real scalars,
 * ORC_ALPHA_F64 / ORC_BETA_F64 their width next to ComplexF32 data (promotion as in lo_oracle_impl.h).
 */
#define CCAT_(a, b) a##_##b
#define CCAT(a, b) CCAT_(a, b)
#define CFN(name) CCAT(name, SUF)

/* BODY(RA, RB): component types of the alpha- and beta-terms */
#define WITH_RAB(BODY)                                                                           \
  do {                                                                                           \
    const int ad_ = sizeof(R) == 8 || (flags & ORC_ALPHA_F64);                                   \
    const int bd_ = sizeof(R) == 8 || (flags & ORC_BETA_F64);                                    \
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

/* t = s * w for a caller scalar s = (sr, si) that is Real (s_real) or Complex, in type RT */
#define SMUL(RT, sr, si, s_real, wr, wi, tr, ti)                                                 \
  do {                                                                                           \
    if (s_real) {                                                                                \
      tr = (sr) * (RT)(wr);                                                                      \
      ti = (sr) * (RT)(wi);                                                                      \
    } else {                                                                                     \
      tr = ((sr) * (RT)(wr)) - ((si) * (RT)(wi));                                                \
      ti = ((sr) * (RT)(wi)) + ((si) * (RT)(wr));                                                \
    }                                                                                            \
  } while (0)

/* mulSquareOpDiagonal! / mulOpDiagonal! — src/special-operators.jl:125-131,144-151; ORC_CONJ_D: ctprod! uses
 * conj.(d) (:139-141) */
void CFN(orc_diag_mul)(R *res, const R *d, const R *v, int64_t n_min, int64_t nrow, double are, double aim,
                       double bre, double bim, int32_t flags) {
  const int a_real = (flags & ORC_ALPHA_REAL) != 0, b_real = (flags & ORC_BETA_REAL) != 0;
  const int cj = (flags & ORC_CONJ_D) != 0, b0 = (bre == 0 && (b_real || bim == 0));
#define BODY(RA, RB)                                                                             \
  {                                                                                              \
    const RA ar = (RA)are, ai = (RA)aim;                                                         \
    const RB br = (RB)bre, bi = (RB)bim;                                                         \
    for (int64_t i = 0; i < n_min; ++i) {                                                        \
      const R dr = d[2 * i], di = cj ? -d[2 * i + 1] : d[2 * i + 1];                             \
      RA tr, ti;                                                                                 \
      SMUL(RA, ar, ai, a_real, dr, di, tr, ti); /* α*d */                                        \
      const RA vr = (RA)v[2 * i], vi = (RA)v[2 * i + 1];                                         \
      const RA ur = (tr * vr) - (ti * vi), ui = (tr * vi) + (ti * vr); /* (α*d)*v */             \
      if (b0) { /* :126-127 */                                                                   \
        res[2 * i] = (R)ur;                                                                      \
        res[2 * i + 1] = (R)ui;                                                                  \
      } else { /* :128-129 */                                                                    \
        RB wr, wi;                                                                               \
        SMUL(RB, br, bi, b_real, res[2 * i], res[2 * i + 1], wr, wi); /* β*res */                \
        res[2 * i] = (R)(ur + wr); /* usual arithmetic conversions = promote_type */             \
        res[2 * i + 1] = (R)(ui + wi);                                                           \
      }                                                                                          \
    }                                                                                            \
  }
  WITH_RAB(BODY);
#undef BODY
  for (int64_t i = 2 * n_min; i < 2 * nrow; ++i) res[i] = 0; /* :150 */
}

/* mulOpEye! — src/special-operators.jl:36-44 (ORC_TAIL_BETA: the tail receives β itself; else β*res) */
void CFN(orc_eye_mul)(R *res, const R *v, int64_t n_min, int64_t nrow, double are, double aim, double bre,
                      double bim, int32_t flags) {
  const int a_real = (flags & ORC_ALPHA_REAL) != 0, b_real = (flags & ORC_BETA_REAL) != 0;
  const int b0 = (bre == 0 && (b_real || bim == 0));
#define BODY(RA, RB)                                                                             \
  {                                                                                              \
    const RA ar = (RA)are, ai = (RA)aim;                                                         \
    const RB br = (RB)bre, bi = b_real ? (RB)0 : (RB)bim;                                        \
    for (int64_t i = 0; i < n_min; ++i) {                                                        \
      RA tr, ti;                                                                                 \
      SMUL(RA, ar, ai, a_real, v[2 * i], v[2 * i + 1], tr, ti);                                  \
      if (b0) {                                                                                  \
        res[2 * i] = (R)tr;                                                                      \
        res[2 * i + 1] = (R)ti;                                                                  \
      } else {                                                                                   \
        RB wr, wi;                                                                               \
        SMUL(RB, br, bi, b_real, res[2 * i], res[2 * i + 1], wr, wi);                            \
        res[2 * i] = (R)(tr + wr);                                                               \
        res[2 * i + 1] = (R)(ti + wi);                                                           \
      }                                                                                          \
    }                                                                                            \
    for (int64_t i = n_min; i < nrow; ++i) {                                                     \
      if (b0) {                                                                                  \
        res[2 * i] = res[2 * i + 1] = 0;                                                         \
      } else if (flags & ORC_TAIL_BETA) {                                                        \
        res[2 * i] = (R)br;                                                                      \
        res[2 * i + 1] = (R)bi;                                                                  \
      } else { /* res*β, Complex*Complex with z = res */                                         \
        const RB rr = (RB)res[2 * i], ri = (RB)res[2 * i + 1];                                   \
        if (b_real) {                                                                            \
          res[2 * i] = (R)(rr * br);                                                             \
          res[2 * i + 1] = (R)(ri * br);                                                         \
        } else {                                                                                 \
          res[2 * i] = (R)((rr * br) - (ri * bi));                                               \
          res[2 * i + 1] = (R)((rr * bi) + (ri * br));                                           \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  }
  WITH_RAB(BODY);
#undef BODY
}

/* res .*= s — mulOpZeros! with β (src/special-operators.jl:106) and prod3!'s res .*= α (src/operations.jl:14);
 * f64s: the scalar is a Float64 / ComplexF64 */
void CFN(orc_scale)(R *res, int64_t n, double sre, double sim, int32_t s_real, int32_t f64s) {
#define SC(RS)                                                                                   \
  {                                                                                              \
    const RS sr = (RS)sre, si = (RS)sim;                                                         \
    for (int64_t i = 0; i < n; ++i) {                                                            \
      const RS rr = (RS)res[2 * i], ri = (RS)res[2 * i + 1];                                     \
      if (s_real) {                                                                              \
        res[2 * i] = (R)(rr * sr);                                                               \
        res[2 * i + 1] = (R)(ri * sr);                                                           \
      } else {                                                                                   \
        res[2 * i] = (R)((rr * sr) - (ri * si));                                                 \
        res[2 * i + 1] = (R)((rr * si) + (ri * sr));                                             \
      }                                                                                          \
    }                                                                                            \
  }
  if (sizeof(R) == 8 || f64s) SC(double) else SC(float)
#undef SC
}

/* LinearAlgebra.dot(h, v) = sum conj(h_i) v_i (BLAS zdotc/cdotc in the reference: order unpinned) */
void CFN(orc_dotc)(const R *h, const R *v, int64_t n, R *out) {
  R sr = 0, si = 0;
  for (int64_t i = 0; i < n; ++i) {
    sr += (h[2 * i] * v[2 * i]) + (h[2 * i + 1] * v[2 * i + 1]);
    si += (h[2 * i] * v[2 * i + 1]) - (h[2 * i + 1] * v[2 * i]);
  }
  out[0] = sr;
  out[1] = si;
}

/* mulHouseholder! — src/linalg.jl:77-83: res .= α .* (v .- 2 * dot(h, v) .* h) (.+ β .* res) */
void CFN(orc_householder_mul)(R *res, const R *h, const R *v, int64_t n, double are, double aim, double bre,
                              double bim, int32_t flags) {
  const int a_real = (flags & ORC_ALPHA_REAL) != 0, b_real = (flags & ORC_BETA_REAL) != 0;
  const int b0 = (bre == 0 && (b_real || bim == 0));
  R dt[2];
  CFN(orc_dotc)(h, v, n, dt);
  const R cr = (R)2 * dt[0], ci = (R)2 * dt[1]; /* 2 * dot: Int * Complex */
#define BODY(RA, RB)                                                                             \
  {                                                                                              \
    const RA ar = (RA)are, ai = (RA)aim;                                                         \
    const RB br = (RB)bre, bi = (RB)bim;                                                         \
    for (int64_t i = 0; i < n; ++i) {                                                            \
      const R hr = h[2 * i], hi = h[2 * i + 1];                                                  \
      const R pr = (cr * hr) - (ci * hi), pi = (cr * hi) + (ci * hr); /* c .* h */               \
      const R ir = v[2 * i] - pr, ii = v[2 * i + 1] - pi;                                        \
      RA tr, ti;                                                                                 \
      SMUL(RA, ar, ai, a_real, ir, ii, tr, ti);                                                  \
      if (b0) {                                                                                  \
        res[2 * i] = (R)tr;                                                                      \
        res[2 * i + 1] = (R)ti;                                                                  \
      } else {                                                                                   \
        RB wr, wi;                                                                               \
        SMUL(RB, br, bi, b_real, res[2 * i], res[2 * i + 1], wr, wi);                            \
        res[2 * i] = (R)(tr + wr);                                                               \
        res[2 * i + 1] = (R)(ti + wi);                                                           \
      }                                                                                          \
    }                                                                                            \
  }
  WITH_RAB(BODY);
#undef BODY
}

/* res = α*t (+ β*res) for a length-nr complex vector t (shared epilogue of the dense leaves below) */
static void CFN(orc_cplx_epilogue)(R *res, const R *t, int64_t nr, double are, double aim, double bre, double bim,
                                   int32_t flags) {
  const int a_real = (flags & ORC_ALPHA_REAL) != 0, b_real = (flags & ORC_BETA_REAL) != 0;
  const int b0 = (bre == 0 && (b_real || bim == 0));
#define BODY(RA, RB)                                                                             \
  {                                                                                              \
    const RA ar = (RA)are, ai = (RA)aim;                                                         \
    const RB br = (RB)bre, bi = (RB)bim;                                                         \
    for (int64_t i = 0; i < nr; ++i) {                                                           \
      RA tr, ti;                                                                                 \
      SMUL(RA, ar, ai, a_real, t[2 * i], t[2 * i + 1], tr, ti);                                  \
      if (b0) {                                                                                  \
        res[2 * i] = (R)tr;                                                                      \
        res[2 * i + 1] = (R)ti;                                                                  \
      } else {                                                                                   \
        RB wr, wi;                                                                               \
        SMUL(RB, br, bi, b_real, res[2 * i], res[2 * i + 1], wr, wi);                            \
        res[2 * i] = (R)(tr + wr);                                                               \
        res[2 * i + 1] = (R)(ti + wi);                                                           \
      }                                                                                          \
    }                                                                                            \
  }
  WITH_RAB(BODY);
#undef BODY
}

/* dense LinearOperator(M) on Complex{R} — src/constructors.jl:19-29: prod! = mul!(res, M, v, α, β), tprod! with
 * transpose(M), ctprod! with adjoint(M). mode 0: M*v, 1: transpose(M)*v, 2: M'*v, 3: conj(M)*v (what a row-major
 * alias of M' needs). tmp: 2*length(res) scalars. Plain loops in R (BLAS order is unspecified: tolerance-pinned). */
void CFN(orc_gemv)(R *res, const R *M, int64_t m, int64_t n, int64_t ld, const R *v, double are, double aim,
                   double bre, double bim, int32_t mode, int32_t flags, R *tmp) {
  const int trans = (mode == 1 || mode == 2), cj = (mode == 2 || mode == 3);
  const int64_t nr = trans ? n : m;
  for (int64_t i = 0; i < 2 * nr; ++i) tmp[i] = 0;
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i) {
      const R ar_ = M[2 * (i + j * ld)], ai_ = cj ? -M[2 * (i + j * ld) + 1] : M[2 * (i + j * ld) + 1];
      const R *x = trans ? v + 2 * i : v + 2 * j;
      R *o = trans ? tmp + 2 * j : tmp + 2 * i;
      o[0] += (ar_ * x[0]) - (ai_ * x[1]);
      o[1] += (ar_ * x[1]) + (ai_ * x[0]);
    }
  CFN(orc_cplx_epilogue)(res, tmp, nr, are, aim, bre, bim, flags);
}

/* sparse LinearOperator(M::SparseMatrixCSC{Complex{R}}) — src/constructors.jl:19-29 -> the SparseArrays stdlib's mul!
 * (see lo_oracle_impl.h: orc_csc_mul for the loops and the pinning note). mode 0: A*v (`_spmatmul!`), 1: transpose(A)*v,
 * 2: A'*v (`_At_or_Ac_mul_B!` with tfun = transpose / adjoint). colptr / rowval 1-based. */
void CFN(orc_csc_mul)(R *res, const int64_t *colptr, const int64_t *rowval, const R *nzval, int64_t m, int64_t n, const R *v,
                      double are, double aim, double bre, double bim, int32_t mode, int32_t flags) {
  const int a_real = (flags & ORC_ALPHA_REAL) != 0, b_real = (flags & ORC_BETA_REAL) != 0;
  const int64_t nr = mode ? n : m;
  const int b_one = (bre == 1 && (b_real || bim == 0)), b0 = (bre == 0 && (b_real || bim == 0));
#define BODY(RA, RB)                                                                             \
  {                                                                                              \
    const RA ar = (RA)are, ai = (RA)aim;                                                         \
    const RB br = (RB)bre, bi = (RB)bim;                                                         \
    if (!b_one) {                                        /* _rmul_or_fill!(C, β) */              \
      for (int64_t i = 0; i < nr; ++i) {                                                         \
        if (b0) { res[2 * i] = 0; res[2 * i + 1] = 0; }                                          \
        else { RB tr, ti; SMUL(RB, br, bi, b_real, res[2 * i], res[2 * i + 1], tr, ti); res[2 * i] = (R)tr; res[2 * i + 1] = (R)ti; } \
      }                                                                                          \
    }                                                                                            \
    for (int64_t col = 0; col < n; ++col) {                                                      \
      if (mode == 0) {                                                                           \
        RA xr, xi;                                       /* αxj = B[col] * α */                  \
        SMUL(RA, ar, ai, a_real, v[2 * col], v[2 * col + 1], xr, xi);                            \
        for (int64_t k = colptr[col] - 1; k < colptr[col + 1] - 1; ++k) {                        \
          const RA zr = (RA)nzval[2 * k], zi = (RA)nzval[2 * k + 1];                             \
          R *o = res + 2 * (rowval[k] - 1);                                                      \
          o[0] = (R)((RA)o[0] + ((zr * xr) - (zi * xi)));                                        \
          o[1] = (R)((RA)o[1] + ((zr * xi) + (zi * xr)));                                        \
        }                                                                                        \
      } else {                                                                                   \
        R tr = 0, ti = 0;                                /* tmp += tfun(nzv[k]) * B[rv[k]] */    \
        for (int64_t k = colptr[col] - 1; k < colptr[col + 1] - 1; ++k) {                        \
          const R zr = nzval[2 * k], zi = mode == 2 ? -nzval[2 * k + 1] : nzval[2 * k + 1];      \
          const R *x = v + 2 * (rowval[k] - 1);                                                  \
          tr = tr + ((zr * x[0]) - (zi * x[1]));                                                 \
          ti = ti + ((zr * x[1]) + (zi * x[0]));                                                 \
        }                                                                                        \
        RA ur, ui;                                       /* C[col] += tmp * α */                 \
        SMUL(RA, ar, ai, a_real, tr, ti, ur, ui);                                                \
        res[2 * col] = (R)((RA)res[2 * col] + ur);                                               \
        res[2 * col + 1] = (R)((RA)res[2 * col + 1] + ui);                                       \
      }                                                                                          \
    }                                                                                            \
  }
  WITH_RAB(BODY);
#undef BODY
}

/* mulHermitian! — src/linalg.jl:97-103 with L = tril(A, -1) (:111): res .= α .* (d .* v .+ L*v .+ (v'*L)') (.+ β .* res).
 * (v'*L)'[j] = sum_{i>j} conj(L[i,j]) * v[i]. d is Real (d_real: n scalars, the reference test passes real.(diag(A)))
 * or Complex (2n scalars). t1, t2: 2n scalars each. */
void CFN(orc_hermitian_mul)(R *res, const R *d, int32_t d_real, const R *A, int64_t lda, const R *v, int64_t n,
                            double are, double aim, double bre, double bim, int32_t flags, R *t1, R *t2) {
  for (int64_t i = 0; i < 2 * n; ++i) t1[i] = t2[i] = 0;
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = j + 1; i < n; ++i) {
      const R lr = A[2 * (i + j * lda)], li = A[2 * (i + j * lda) + 1];
      t1[2 * i] += (lr * v[2 * j]) - (li * v[2 * j + 1]);          /* L*v */
      t1[2 * i + 1] += (lr * v[2 * j + 1]) + (li * v[2 * j]);
      t2[2 * j] += (lr * v[2 * i]) + (li * v[2 * i + 1]);          /* conj(L[i,j]) * v[i] */
      t2[2 * j + 1] += (lr * v[2 * i + 1]) - (li * v[2 * i]);
    }
  for (int64_t i = 0; i < n; ++i) { /* inner = (d.*v .+ L*v) .+ (v'*L)' in R, reusing t1 */
    R pr, pi;
    if (d_real) {
      pr = d[i] * v[2 * i];
      pi = d[i] * v[2 * i + 1];
    } else {
      pr = (d[2 * i] * v[2 * i]) - (d[2 * i + 1] * v[2 * i + 1]);
      pi = (d[2 * i] * v[2 * i + 1]) + (d[2 * i + 1] * v[2 * i]);
    }
    t1[2 * i] = (pr + t1[2 * i]) + t2[2 * i];
    t1[2 * i + 1] = (pi + t1[2 * i + 1]) + t2[2 * i + 1];
  }
  CFN(orc_cplx_epilogue)(res, t1, n, are, aim, bre, bim, flags);
}

/* kron(A, B) on Complex{R} — src/kron.jl:10-40: (A ⊗ B) x = vec(B X transpose(A)); tprod! (mode 1) on the transposed
 * factors, ctprod! (mode 2) on the conjugate-transposed factors. Column i of the result matrix is opB * (X * w_i) with
 * w_i = row / column i of A (conjugated for mode 2), exactly the reference's `m` single-vector products (:17-18).
 * A: m x n (lda), B: p x q (ldb), complex interleaved. work: 2*(xr + nrows_out*ncols_out) scalars. */
void CFN(orc_kron_mul)(R *res, const R *A, int64_t m, int64_t n, int64_t lda, const R *B, int64_t p, int64_t q,
                       int64_t ldb, const R *x, double are, double aim, double bre, double bim, int32_t mode,
                       int32_t flags, R *work) {
  const int trans = mode != 0, cj = mode == 2;
  const int64_t ncols_out = trans ? n : m, nrows_out = trans ? q : p;
  const int64_t xr = trans ? p : q, xc = trans ? m : n;
  R *u = work, *Rm = work + 2 * xr;
  for (int64_t i = 0; i < ncols_out; ++i) {
    for (int64_t r = 0; r < 2 * xr; ++r) u[r] = 0;
    for (int64_t j = 0; j < xc; ++j) {
      const R *w = trans ? A + 2 * (j + i * lda) : A + 2 * (i + j * lda);
      const R wr = w[0], wi = cj ? -w[1] : w[1];
      for (int64_t r = 0; r < xr; ++r) {
        const R xr_ = x[2 * (r + j * xr)], xi_ = x[2 * (r + j * xr) + 1];
        u[2 * r] += (xr_ * wr) - (xi_ * wi);
        u[2 * r + 1] += (xr_ * wi) + (xi_ * wr);
      }
    }
    R *col = Rm + 2 * i * nrows_out;
    for (int64_t r = 0; r < 2 * nrows_out; ++r) col[r] = 0;
    for (int64_t j = 0; j < q; ++j)
      for (int64_t r = 0; r < p; ++r) {
        const R br_ = B[2 * (r + j * ldb)], bi_ = cj ? -B[2 * (r + j * ldb) + 1] : B[2 * (r + j * ldb) + 1];
        const R *uu = trans ? u + 2 * r : u + 2 * j;
        R *o = trans ? col + 2 * j : col + 2 * r;
        o[0] += (br_ * uu[0]) - (bi_ * uu[1]);
        o[1] += (br_ * uu[1]) + (bi_ * uu[0]);
      }
  }
  CFN(orc_cplx_epilogue)(res, Rm, nrows_out * ncols_out, are, aim, bre, bim, flags);
}

#undef SMUL
#undef WITH_RAB
#undef CFN
#undef CCAT
#undef CCAT_
