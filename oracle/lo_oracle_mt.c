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
