/* All-core timing variant of the mulHouseholder! restatement (src/linalg.jl:77-83) — TEST/BENCH
 * INFRASTRUCTURE ONLY, used solely by bench.py's cpu_baseline leg to report an upper bound for the host
 * (SURVEY §8d "all-core OpenMP mode"). The reference itself is single-threaded here (Julia broadcast
 * does not thread; OpenBLAS ddot threads only above its own cut-off), so the 1-thread oracle in
 * lo_oracle.c is the like-for-like baseline and this file is the generous one.
 * Same per-element arithmetic as orc_householder_mul_f64; the dot is a chunked OpenMP reduction
 * (summation order differs from the 1-thread oracle — timing only, never used as a checker). */
#include <omp.h>
#include <stdint.h>

static inline uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* first-touch initialisation by the threads that will stream the pages later: x[i] = lo + (hi-lo)*u01 */
void orc_mt_fill_f64(double *x, int64_t n, uint64_t seed, double lo, double hi, int32_t threads) {
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int64_t i = 0; i < n; ++i)
    x[i] = lo + (hi - lo) * ((double)(splitmix(seed + (uint64_t)i) >> 11) * (1.0 / 9007199254740992.0));
}

void orc_mt_householder_mul_f64(double *res, const double *h, const double *v, int64_t n, double alpha,
                                double beta, int32_t threads) {
  double dot = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : dot) num_threads(threads)
  for (int64_t i = 0; i < n; ++i) dot += h[i] * v[i];
  const double c = 2.0 * dot;
  if (beta == 0) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; ++i) res[i] = alpha * (v[i] - (c * h[i]));
  } else {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; ++i) res[i] = (alpha * (v[i] - (c * h[i]))) + (beta * res[i]);
  }
}

int32_t orc_mt_max_threads(void) { return (int32_t)omp_get_max_threads(); }

/* ---- all-core variants of the OTHER two figures of the metric string (bench.py extras.*.cpu.all_cores) ----------------
 * Same statement sequence as the 1-thread restatements in lo_oracle_impl.h; every statement is one OpenMP loop (dots are
 * chunked reductions). Timing only. */

/* lbfgs_multiply, inverse operator (src/lbfgs.jl:117-154): q .= x; for the pairs newest -> oldest: αk = dot(s_k, q)/ys_k;
 * q .-= αk .* y_k; q .*= γ; for the pairs oldest -> newest: β = αk - dot(y_k, q)/ys_k; q .+= β .* s_k; res .= αm .* q (.+ βm .* res).
 * S, Y: n x mem column-major panels (leading dimension ld), `insert` 1-based as in the reference. */
void orc_mt_lbfgs_inv_mul_f64(double *res, const double *S, const double *Y, int64_t ld, const double *ys, double *alpha_k,
                              int32_t mem, int32_t insert, int32_t scaling, double gamma, const double *x, double *q, int64_t n,
                              double am, double bm, int32_t threads) {
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int64_t i = 0; i < n; ++i) q[i] = x[i];
  for (int32_t i = 1; i <= mem; ++i) {
    const int32_t k = (((insert - i - 1) % mem) + mem) % mem;          /* 0-based slot of mod(insert - i - 1, mem) + 1 */
    if (ys[k] != 0) {
      const double *sk = S + (int64_t)k * ld, *yk = Y + (int64_t)k * ld;
      double d = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : d) num_threads(threads)
      for (int64_t j = 0; j < n; ++j) d += sk[j] * q[j];
      const double ak = d / ys[k];
      alpha_k[k] = ak;
#pragma omp parallel for schedule(static) num_threads(threads)
      for (int64_t j = 0; j < n; ++j) q[j] -= ak * yk[j];
    }
  }
  if (scaling) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t j = 0; j < n; ++j) q[j] *= gamma;
  }
  for (int32_t i = 1; i <= mem; ++i) {
    const int32_t k = (((insert + i - 2) % mem) + mem) % mem;
    if (ys[k] != 0) {
      const double *sk = S + (int64_t)k * ld, *yk = Y + (int64_t)k * ld;
      double d = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : d) num_threads(threads)
      for (int64_t j = 0; j < n; ++j) d += yk[j] * q[j];
      const double b = alpha_k[k] - d / ys[k];
#pragma omp parallel for schedule(static) num_threads(threads)
      for (int64_t j = 0; j < n; ++j) q[j] += b * sk[j];
    }
  }
  if (bm == 0) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t j = 0; j < n; ++j) res[j] = am * q[j];
  } else {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t j = 0; j < n; ++j) res[j] = (am * q[j]) + (bm * res[j]);
  }
}

/* kron(A, B) prod! in the reference's literal form (src/kron.jl:17-18 through Matrix(B * X * transpose(A)),
 * src/abstract.jl:282-292): column i of the result = B * (X * A[i, :]) — the m columns are independent, one per thread at a
 * time (each thread owns an intermediate of q doubles in `work`, threads * q long). A: m x n, B: p x q, column-major. */
void orc_mt_kron_mul_f64(double *res, const double *A, int64_t m, int64_t n, const double *B, int64_t p, int64_t q,
                         const double *x, double alpha, double beta, double *work, int32_t threads) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
  for (int64_t i = 0; i < m; ++i) {
    double *u = work + (int64_t)omp_get_thread_num() * q;
    for (int64_t r = 0; r < q; ++r) u[r] = 0;
    for (int64_t j = 0; j < n; ++j) {
      const double wj = A[i + j * m];
      const double *xj = x + j * q;
      for (int64_t r = 0; r < q; ++r) u[r] += xj[r] * wj;
    }
    double *col = res + i * p;
    if (beta == 0) {
      for (int64_t r = 0; r < p; ++r) col[r] = 0;
    } else {
      for (int64_t r = 0; r < p; ++r) col[r] = (beta / alpha) * col[r];   /* timing variant: α(Σ + (β/α) res) */
    }
    for (int64_t j = 0; j < q; ++j) {
      const double uj = u[j];
      const double *bj = B + j * p;
      for (int64_t r = 0; r < p; ++r) col[r] += bj[r] * uj;
    }
    for (int64_t r = 0; r < p; ++r) col[r] *= alpha;
  }
}
