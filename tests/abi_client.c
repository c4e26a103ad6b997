/* abi_client.c — a plain-C client of include/mxlo.h (no Python, no torch): what a foreign-language glue
 * (Julia `ccall`) does. Built and run by tests/test_gpu_abi_client.py on the GPU box:
 *   gcc -std=c99 tests/abi_client.c -Iinclude -Llinearoperators.jl_amd/csrc -lmxlo -lm
 * Checks opDiagonal (bit-exact), opHouseholder (1e-12), restriction (bit-exact), an InverseLBFGS
 * push!/mul! round against an in-file statement-by-statement two-loop, and error codes. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mxlo.h"

#define CK(call)                                                                          \
  do {                                                                                    \
    int32_t st__ = (call);                                                                \
    if (st__ != MXLO_OK) {                                                                \
      printf("FAIL %s -> %d (%s)\n", #call, st__, mxlo_last_error());                     \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

static double urand(unsigned long long *s) { /* splitmix64 -> [0,1) */
  unsigned long long z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

static void *dev_from(mxlo_ctx *ctx, const void *h, size_t bytes) {
  void *p = NULL;
  if (mxlo_malloc(ctx, (int64_t)bytes, &p) != MXLO_OK) return NULL;
  if (h && mxlo_memcpy_h2d(ctx, p, h, (int64_t)bytes) != MXLO_OK) return NULL;
  return p;
}

int main(void) {
  const int64_t n = 100003;
  unsigned long long seed = 42;
  mxlo_ctx *ctx = NULL;
  CK(mxlo_ctx_create(0, NULL, &ctx));
  printf("%s\n", mxlo_version());

  double *d = malloc(n * 8), *v = malloc(n * 8), *r0 = malloc(n * 8), *out = malloc(n * 8), *ref = malloc(n * 8);
  double nrm = 0;
  for (int64_t i = 0; i < n; ++i) {
    d[i] = urand(&seed) - 0.5;
    v[i] = 2 * urand(&seed) - 1;
    r0[i] = 2 * urand(&seed) - 1;
    nrm += d[i] * d[i];
  }
  nrm = sqrt(nrm);
  void *dd = dev_from(ctx, d, n * 8), *dv = dev_from(ctx, v, n * 8), *dr = dev_from(ctx, r0, n * 8);
  if (!dd || !dv || !dr) return 1;

  /* opDiagonal 5-arg mul!: bit-exact vs (alpha*d)*v + beta*res */
  const double alpha = 2.0 / 3.0, beta = -0.3;
  CK(mxlo_diag_mul(ctx, MXLO_F64, dr, dd, dv, n, n, alpha, beta, 0));
  CK(mxlo_memcpy_d2h(ctx, out, dr, n * 8));
  for (int64_t i = 0; i < n; ++i) {
    volatile double t1 = alpha * d[i];
    volatile double t2 = t1 * v[i];
    volatile double t3 = beta * r0[i];
    ref[i] = t2 + t3;
  }
  if (memcmp(out, ref, n * 8) != 0) { printf("FAIL diag not bit-exact\n"); return 1; }

  /* opHouseholder: h = d/||d|| */
  for (int64_t i = 0; i < n; ++i) d[i] /= nrm;
  CK(mxlo_memcpy_h2d(ctx, dd, d, n * 8));
  CK(mxlo_householder_mul(ctx, MXLO_F64, dr, dd, dv, n, 1.0, 0.0, 0));
  CK(mxlo_memcpy_d2h(ctx, out, dr, n * 8));
  double dot = 0, err = 0, rn = 0;
  for (int64_t i = 0; i < n; ++i) dot += d[i] * v[i];
  for (int64_t i = 0; i < n; ++i) {
    const double want = v[i] - (2 * dot) * d[i];
    err += (out[i] - want) * (out[i] - want);
    rn += want * want;
  }
  if (sqrt(err / rn) > 1e-12) { printf("FAIL householder %.3e\n", sqrt(err / rn)); return 1; }

  /* opRestriction 1:2:n (range form) and its adjoint */
  const int64_t len = (n + 1) / 2;
  void *dw = dev_from(ctx, NULL, len * 8);
  CK(mxlo_gather_range(ctx, 8, dw, dv, n, 1, 2, len));
  CK(mxlo_memcpy_d2h(ctx, out, dw, len * 8));
  for (int64_t k = 0; k < len; ++k)
    if (out[k] != v[2 * k]) { printf("FAIL gather_range\n"); return 1; }
  CK(mxlo_scatter_zero_range(ctx, 8, dr, n, dw, 1, 2, len));
  CK(mxlo_memcpy_d2h(ctx, out, dr, n * 8));
  for (int64_t i = 0; i < n; ++i)
    if (out[i] != ((i % 2 == 0) ? v[i] : 0.0)) { printf("FAIL scatter_zero_range\n"); return 1; }

  /* InverseLBFGSOperator: push 7 pairs into mem = 5, then d = -(H*g) in both evaluation modes */
  enum { MEM = 5, NP = 7 };
  const int64_t m = 4099;
  mxlo_qn *H = NULL;
  CK(mxlo_qn_create(ctx, MXLO_QN_LBFGS_INV, MXLO_F64, m, MEM, 1, 0, 0.99, 10.0, &H));
  static double S[NP][4099], Y[NP][4099], g[4099], q[4099], al[NP];
  void *ds = dev_from(ctx, NULL, m * 8), *dy = dev_from(ctx, NULL, m * 8), *dg = dev_from(ctx, NULL, m * 8),
       *dres = dev_from(ctx, NULL, m * 8);
  double ys[NP], gamma = 1.0;
  for (int p = 0; p < NP; ++p) {
    ys[p] = 0;
    double yy = 0;
    for (int64_t i = 0; i < m; ++i) {
      S[p][i] = 2 * urand(&seed) - 1;
      Y[p][i] = S[p][i] * (0.5 + 1.5 * urand(&seed));
      ys[p] += Y[p][i] * S[p][i];
      yy += Y[p][i] * Y[p][i];
    }
    gamma = ys[p] / yy;
    CK(mxlo_memcpy_h2d(ctx, ds, S[p], m * 8));
    CK(mxlo_memcpy_h2d(ctx, dy, Y[p], m * 8));
    int32_t acc = 0;
    CK(mxlo_qn_push(H, ds, dy, &acc));
    if (!acc) { printf("FAIL push rejected\n"); return 1; }
  }
  double sc[5];
  CK(mxlo_qn_get_scalars(H, sc, NULL, NULL));
  if ((int)sc[0] != NP % MEM + 1 || fabs(sc[1] - gamma) > 1e-12 * gamma) { printf("FAIL scalars\n"); return 1; }
  for (int64_t i = 0; i < m; ++i) g[i] = 2 * urand(&seed) - 1;
  CK(mxlo_memcpy_h2d(ctx, dg, g, m * 8));
  /* reference two-loop (src/lbfgs.jl:127-153) over the last MEM pairs, newest first */
  memcpy(q, g, m * 8);
  for (int p = NP - 1; p >= NP - MEM; --p) {
    double sq = 0;
    for (int64_t i = 0; i < m; ++i) sq += S[p][i] * q[i];
    al[p] = sq / ys[p];
    for (int64_t i = 0; i < m; ++i) q[i] -= al[p] * Y[p][i];
  }
  for (int64_t i = 0; i < m; ++i) q[i] *= gamma;
  for (int p = NP - MEM; p < NP; ++p) {
    double yq = 0;
    for (int64_t i = 0; i < m; ++i) yq += Y[p][i] * q[i];
    const double b = al[p] - yq / ys[p];
    for (int64_t i = 0; i < m; ++i) q[i] += b * S[p][i];
  }
  for (int mode = 0; mode < 2; ++mode) {
    CK(mxlo_qn_set_mode(H, mode));
    CK(mxlo_qn_mul(H, dres, dg, -1.0, 0.0, 0));
    CK(mxlo_memcpy_d2h(ctx, out, dres, m * 8));
    err = rn = 0;
    for (int64_t i = 0; i < m; ++i) {
      err += (out[i] + q[i]) * (out[i] + q[i]);
      rn += q[i] * q[i];
    }
    if (sqrt(err / rn) > 1e-10) { printf("FAIL lbfgs mode %d: %.3e\n", mode, sqrt(err / rn)); return 1; }
  }
  /* ShiftedOperator(H, sigma) apply fused into the combine pass == mul! followed by axpy!(alpha*sigma, x, y), bit for bit */
  {
    const double sigma = 0.37, al2 = 2.0, be2 = -3.0;
    void *da = dev_from(ctx, g, m * 8), *db = dev_from(ctx, g, m * 8);   /* both start from res0 = g */
    static double oa[4099], ob[4099];
    CK(mxlo_qn_mul_shifted(H, da, dg, al2, be2, sigma, 0));
    CK(mxlo_qn_mul(H, db, dg, al2, be2, 0));
    CK(mxlo_eye_mul(ctx, MXLO_F64, db, dg, m, m, al2 * sigma, 1.0, 0));
    CK(mxlo_memcpy_d2h(ctx, oa, da, m * 8));
    CK(mxlo_memcpy_d2h(ctx, ob, db, m * 8));
    if (memcmp(oa, ob, m * 8) != 0) { printf("FAIL shifted apply not bit-identical to mul + axpy\n"); return 1; }
    CK(mxlo_free(ctx, da)); CK(mxlo_free(ctx, db));
  }
  /* hipGraph: record {Householder apply; inverse L-BFGS apply accumulating into it} once, replay on new data */
  {
    void *stream = NULL;
    mxlo_graph *gr = NULL;
    static double o1[4099], o2[4099], hh[4099];
    double hn = 0;
    for (int64_t i = 0; i < m; ++i) { hh[i] = 2 * urand(&seed) - 1; hn += hh[i] * hh[i]; }
    for (int64_t i = 0; i < m; ++i) hh[i] /= sqrt(hn);
    void *dh = dev_from(ctx, hh, m * 8), *dacc = dev_from(ctx, NULL, m * 8);
    if (mxlo_graph_begin(ctx) != MXLO_ESTATE) { printf("FAIL default stream must not be capturable\n"); return 1; }
    CK(mxlo_ctx_sync(ctx));
    CK(mxlo_ctx_create_stream(ctx, &stream));
    CK(mxlo_householder_mul(ctx, MXLO_F64, dacc, dh, dg, m, 1.0, 0.0, 0));      /* warm-up */
    CK(mxlo_graph_begin(ctx));
    CK(mxlo_householder_mul(ctx, MXLO_F64, dacc, dh, dg, m, 1.0, 0.0, 0));
    CK(mxlo_qn_mul(H, dacc, dg, 0.5, 1.0, 0));
    CK(mxlo_graph_end(ctx, &gr));
    for (int64_t i = 0; i < m; ++i) g[i] = 2 * urand(&seed) - 1;               /* new contents, same buffer */
    CK(mxlo_memcpy_h2d(ctx, dg, g, m * 8));
    CK(mxlo_graph_launch(gr));
    CK(mxlo_memcpy_d2h(ctx, o1, dacc, m * 8));
    CK(mxlo_householder_mul(ctx, MXLO_F64, dacc, dh, dg, m, 1.0, 0.0, 0));
    CK(mxlo_qn_mul(H, dacc, dg, 0.5, 1.0, 0));
    CK(mxlo_memcpy_d2h(ctx, o2, dacc, m * 8));
    if (memcmp(o1, o2, m * 8) != 0) { printf("FAIL graph replay differs from the eager calls\n"); return 1; }
    CK(mxlo_graph_destroy(gr));
    CK(mxlo_free(ctx, dh)); CK(mxlo_free(ctx, dacc));
  }
  /* diagonal quasi-Newton push! (DiagonalPSB, src/DiagonalHessianApproximation.jl:45-64) vs the statements in C */
  {
    static double d0[4099], dref[4099], dout[4099];
    double s2 = 0, s4 = 0, sy = 0, s2d = 0;
    for (int64_t i = 0; i < m; ++i) d0[i] = 0.5 + urand(&seed);
    for (int64_t i = 0; i < m; ++i) {
      const double si = S[0][i], q2 = si * si;
      s2 += q2; s4 += q2 * q2; sy += si * Y[0][i]; s2d += q2 * d0[i];
    }
    const double sn = sqrt(s2), sn2 = sn * sn, trA2 = s4 / (sn2 * sn2), qq = ((sy / sn2) - (s2d / sn2)) / trA2, c = qq / sn2;
    for (int64_t i = 0; i < m; ++i) dref[i] = d0[i] + c * (S[0][i] * S[0][i]);
    void *ddq = dev_from(ctx, d0, m * 8);
    int32_t st = -1;
    CK(mxlo_memcpy_h2d(ctx, ds, S[0], m * 8));
    CK(mxlo_memcpy_h2d(ctx, dy, Y[0], m * 8));
    CK(mxlo_diagqn_push(ctx, MXLO_F64, MXLO_DQN_PSB, ddq, ds, dy, m, &st));
    CK(mxlo_memcpy_d2h(ctx, dout, ddq, m * 8));
    err = rn = 0;
    for (int64_t i = 0; i < m; ++i) { err += (dout[i] - dref[i]) * (dout[i] - dref[i]); rn += dref[i] * dref[i]; }
    if (st != 0 || sqrt(err / rn) > 1e-12) { printf("FAIL diagqn push %d %.3e\n", st, sqrt(err / rn)); return 1; }
    CK(mxlo_fill(ctx, MXLO_F64, ds, m, 0.0));
    CK(mxlo_diagqn_push(ctx, MXLO_F64, MXLO_DQN_PSB, ddq, ds, dy, m, &st));
    if (st != 1) { printf("FAIL s == 0 must be reported\n"); return 1; }
    CK(mxlo_free(ctx, ddq));
  }
  /* sparse LinearOperator(M::SparseMatrixCSC): Julia's three arrays as stored (1-based), A*x and A'*x against the
   * SparseArrays loops written out in C (src/constructors.jl:19-29 -> mul!(res, M, v, α, β)); then the same matrix as a
   * block of the ONE-launch block-diagonal operator next to a diagonal block */
  {
    enum { SM = 3001, SN = 2500, PER = 5 };
    static int64_t cp[SN + 1], rv[SN * PER];
    static double nz[SN * PER], xs[SN], us[SM], ref[SM], reft[SN], got[SM > SN ? SM : SN];
    for (int64_t j = 0; j <= SN; ++j) cp[j] = 1 + j * PER;
    for (int64_t j = 0; j < SN; ++j)
      for (int k = 0; k < PER; ++k) {
        rv[j * PER + k] = 1 + (int64_t)((j * 7 + k * 601) % SM);      /* distinct rows within a column (601 * k mod 3001) */
        nz[j * PER + k] = 2 * urand(&seed) - 1;
      }
    for (int64_t j = 0; j < SN; ++j) xs[j] = 2 * urand(&seed) - 1;
    for (int64_t i = 0; i < SM; ++i) { us[i] = 2 * urand(&seed) - 1; ref[i] = 0.25 * i; }
    for (int64_t j = 0; j < SN; ++j) reft[j] = 0;
    /* _spmatmul!: C .*= beta; C[rowval[k]] += nzval[k] * (x[col] * alpha)     (alpha = 2, beta = -3) */
    for (int64_t i = 0; i < SM; ++i) ref[i] *= -3.0;
    for (int64_t j = 0; j < SN; ++j) {
      const double axj = xs[j] * 2.0;
      for (int64_t k = cp[j] - 1; k < cp[j + 1] - 1; ++k) ref[rv[k] - 1] += nz[k] * axj;
    }
    for (int64_t j = 0; j < SN; ++j) {                                 /* _At_or_Ac_mul_B!, beta = 0 */
      double tmp = 0;
      for (int64_t k = cp[j] - 1; k < cp[j + 1] - 1; ++k) tmp += nz[k] * us[rv[k] - 1];
      reft[j] += tmp * 1.0;
    }
    void *dcp = dev_from(ctx, cp, sizeof cp), *drv = dev_from(ctx, rv, sizeof rv), *dnz = dev_from(ctx, nz, sizeof nz);
    void *dxs = dev_from(ctx, xs, sizeof xs), *dus = dev_from(ctx, us, sizeof us), *dout = dev_from(ctx, NULL, sizeof got);
    mxlo_csc *A = NULL;
    CK(mxlo_csc_create(ctx, MXLO_F64, SM, SN, (const int64_t *)dcp, (const int64_t *)drv, dnz, 1, &A));
    int64_t info[8];
    CK(mxlo_csc_info(A, info));
    if (info[0] != SM || info[1] != SN || info[2] != SN * PER || info[5] != 0) { printf("FAIL csc info\n"); return 1; }
    for (int64_t i = 0; i < SM; ++i) got[i] = 0.25 * i;
    CK(mxlo_memcpy_h2d(ctx, dout, got, SM * 8));
    CK(mxlo_csc_mul(A, dout, dxs, 2.0, -3.0, MXLO_OP_N, 0));
    CK(mxlo_memcpy_d2h(ctx, got, dout, SM * 8));
    err = rn = 0;
    for (int64_t i = 0; i < SM; ++i) { err += (got[i] - ref[i]) * (got[i] - ref[i]); rn += ref[i] * ref[i]; }
    if (sqrt(err / rn) > 1e-14) { printf("FAIL csc A*x %.3e\n", sqrt(err / rn)); return 1; }
    CK(mxlo_csc_mul(A, dout, dus, 1.0, 0.0, MXLO_OP_T, 0));
    CK(mxlo_memcpy_d2h(ctx, got, dout, SN * 8));
    err = rn = 0;
    for (int64_t j = 0; j < SN; ++j) { err += (got[j] - reft[j]) * (got[j] - reft[j]); rn += reft[j] * reft[j]; }
    if (sqrt(err / rn) > 1e-14) { printf("FAIL csc A'*x %.3e\n", sqrt(err / rn)); return 1; }
    /* BlockDiagonalOperator(opDiagonal(d[0:100]), A): one launch, the sparse block at row / column offset 100 */
    mxlo_block_desc bl[2];
    memset(bl, 0, sizeof bl);
    bl[0].kind = MXLO_BLK_DIAG; bl[0].m = bl[0].n = 100; bl[0].data = dd;
    bl[1].kind = MXLO_BLK_CSC; bl[1].row_off = 100; bl[1].col_off = 100; bl[1].m = SM; bl[1].n = SN; bl[1].data = A;
    mxlo_blockdiag *bd = NULL;
    CK(mxlo_blockdiag_create(ctx, MXLO_F64, bl, 2, &bd));
    static double xin[100 + SN], yout[100 + SM];
    for (int64_t i = 0; i < 100; ++i) xin[i] = 1.0;
    memcpy(xin + 100, xs, sizeof xs);
    void *dxin = dev_from(ctx, xin, sizeof xin), *dyo = dev_from(ctx, NULL, sizeof yout);
    CK(mxlo_blockdiag_mul(bd, dyo, dxin, 2.0, 0.0, MXLO_OP_N, 0));
    CK(mxlo_memcpy_d2h(ctx, yout, dyo, sizeof yout));
    CK(mxlo_csc_mul(A, dout, dxs, 2.0, 0.0, MXLO_OP_N, 0));
    CK(mxlo_memcpy_d2h(ctx, got, dout, SM * 8));
    if (memcmp(yout + 100, got, SM * 8) != 0) { printf("FAIL sparse block of the fused operator differs from the leaf\n"); return 1; }
    for (int64_t i = 0; i < 100; ++i)
      if (yout[i] != (2.0 * d[i]) * 1.0) { printf("FAIL diagonal block next to the sparse block\n"); return 1; }
    /* a bad structure is an error code naming the position, never a crash */
    cp[3] = cp[2] - 1;
    CK(mxlo_memcpy_h2d(ctx, dcp, cp, sizeof cp));
    mxlo_csc *B = NULL;
    if (mxlo_csc_create(ctx, MXLO_F64, SM, SN, (const int64_t *)dcp, (const int64_t *)drv, dnz, 1, &B) != MXLO_EINVAL) {
      printf("FAIL a decreasing colptr must be MXLO_EINVAL\n"); return 1;
    }
    CK(mxlo_blockdiag_destroy(bd));
    CK(mxlo_csc_destroy(A));
    CK(mxlo_free(ctx, dcp)); CK(mxlo_free(ctx, drv)); CK(mxlo_free(ctx, dnz)); CK(mxlo_free(ctx, dxs)); CK(mxlo_free(ctx, dus));
    CK(mxlo_free(ctx, dout)); CK(mxlo_free(ctx, dxin)); CK(mxlo_free(ctx, dyo));
  }
  /* error conventions: codes, never exceptions */
  if (mxlo_qn_solve_shifted(H, dres, dg, 0.1) != MXLO_ESTATE) { printf("FAIL expected ESTATE\n"); return 1; }
  if (mxlo_diag_mul(ctx, MXLO_F64, dr, dd, dv, n + 1, n, 1.0, 0.0, 0) != MXLO_ESHAPE) { printf("FAIL expected ESHAPE\n"); return 1; }
  CK(mxlo_qn_destroy(H));
  CK(mxlo_ctx_sync(ctx));
  CK(mxlo_free(ctx, dd)); CK(mxlo_free(ctx, dv)); CK(mxlo_free(ctx, dr)); CK(mxlo_free(ctx, dw));
  CK(mxlo_free(ctx, ds)); CK(mxlo_free(ctx, dy)); CK(mxlo_free(ctx, dg)); CK(mxlo_free(ctx, dres));
  CK(mxlo_ctx_destroy(ctx));
  printf("ABI CLIENT OK\n");
  return 0;
}
