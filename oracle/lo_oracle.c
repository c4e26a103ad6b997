/* lo_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU oracle for the LinearOperators.jl v2.14.2 `mul!` hot path: instantiates
 * lo_oracle_impl.h (a statement-by-statement C restatement of the reference
 * closures, each function citing the src/<file>.jl:<line> it follows) for
 * Float64 and Float32.
 *
 * PINNING: Julia is not available in this image, so the reference cannot be
 * executed here (no oracle/_ref). The oracle is pinned against the
 * known-answer cases held by the reference's OWN tests (SURVEY.md §8c;
 * tests/golden/kat_reference_tests.json, checked in tests/test_oracle_kat.py)
 * and against independent dense-matrix constructions the way the reference's
 * tests do. Results of global reductions (dot) are pinned to tolerance only —
 * exactly as in the reference, whose dot is BLAS with unspecified order.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: Julia never contracts
 * a*b+c into an FMA without @fastmath/muladd).
 *
 * Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_ALPHA_F64 0x1
#define ORC_BETA_F64 0x8
#define ORC_SCALARS_F64 (ORC_ALPHA_F64 | ORC_BETA_F64)
#define ORC_D_SCALAR 0x2
#define ORC_TAIL_BETA 0x4
#define ORC_CONJ_D 0x10
#define ORC_ALPHA_REAL 0x20
#define ORC_BETA_REAL 0x40

#define T double
#define SUF f64
#include "lo_oracle_impl.h"
#undef T
#undef SUF

#define T float
#define SUF f32
#include "lo_oracle_impl.h"
#undef T
#undef SUF

/* Complex{Float64} / Complex{Float32} elementwise leaves + opHouseholder */
#define R double
#define SUF c64
#include "lo_oracle_cplx.h"
#undef R
#undef SUF
#define R float
#define SUF c32
#include "lo_oracle_cplx.h"
#undef R
#undef SUF

/* Generic byte-wise gather/scatter for elem sizes other than 4/8 (bit-exact data movement). */
void orc_restrict_bytes(char *res, const char *v, const int64_t *idx, int64_t nidx, int64_t es) {
  for (int64_t k = 0; k < nidx; ++k) memcpy(res + k * es, v + (idx[k] - 1) * es, (size_t)es);
}
void orc_extend_bytes(char *res, int64_t nres, const char *u, const int64_t *idx, int64_t nidx,
                      int64_t es) {
  memset(res, 0, (size_t)(nres * es));
  for (int64_t k = 0; k < nidx; ++k) memcpy(res + (idx[k] - 1) * es, u + k * es, (size_t)es);
}

const char *orc_version(void) { return "lo_oracle 0.1 (restates LinearOperators.jl v2.14.2)"; }
