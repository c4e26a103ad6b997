// complex_scalars.h — how caller scalars next to ComplexF64 / ComplexF32 data are carried and applied (shared by
// complex.hip and sparse.hip). Julia evaluates `α .* w .+ β .* r` component by component, the α-term in
// promote_type(typeof(α), T), the β-term in promote_type(typeof(β), T); a REAL scalar multiplies componentwise (which
// differs from Complex(x, 0) in signed zeros and non-finite values): MXLO_ALPHA_REAL / MXLO_BETA_REAL.
#pragma once
#include "common.h"
#include "stream_kernels.h"

namespace mxlo {

template <typename R>
using C = cx<R>;

// t (complex, component type RA) (+ b*r): the β-term in RB, the sum in the wider type, one rounding to R per component
template <typename R, typename RA, typename RB, bool BETA0>
__device__ __forceinline__ C<R> cfin(RA tr, RA ti, RB br, RB bi, bool b_real, C<R> r) {
  if constexpr (BETA0) return C<R>((R)tr, (R)ti);
  else {
    using P = Wider<RA, RB>;
    const RB rr = (RB)r.re, ri = (RB)r.im;
    RB ur, ui;
    if (b_real) {
      ur = br * rr;
      ui = br * ri;
    } else {
      ur = (br * rr) - (bi * ri);
      ui = (br * ri) + (bi * rr);
    }
    return C<R>((R)((P)tr + (P)ur), (R)((P)ti + (P)ui));
  }
}

template <typename RA>
struct Sc {   // a caller scalar: complex, or real (a_real) multiplying componentwise
  RA re, im;
  bool real;
  template <typename R>
  __device__ __forceinline__ void mul(C<R> w, RA &tr, RA &ti) const {   // scalar * w
    const RA wr = (RA)w.re, wi = (RA)w.im;
    if (real) {
      tr = re * wr;
      ti = re * wi;
    } else {
      tr = (re * wr) - (im * wi);
      ti = (re * wi) + (im * wr);
    }
  }
};

struct ScalArgs {
  double are, aim, bre, bim;
  bool a_real, b_real, a64, b64;
};

inline ScalArgs scal_args(size_t comp, double are, double aim, double bre, double bim, int32_t flags) {
  ScalArgs s;
  s.a64 = comp == 8 || (flags & MXLO_ALPHA_F64);
  s.b64 = comp == 8 || (flags & MXLO_BETA_F64);
  s.a_real = (flags & MXLO_ALPHA_REAL) != 0;
  s.b_real = (flags & MXLO_BETA_REAL) != 0;
  auto rnd = [](double x, bool f64) { return f64 ? x : (double)(float)x; };
  s.are = rnd(are, s.a64);
  s.aim = s.a_real ? 0.0 : rnd(aim, s.a64);
  s.bre = rnd(bre, s.b64);
  s.bim = s.b_real ? 0.0 : rnd(bim, s.b64);
  return s;
}

// calls f.template operator()<RA, RB, BETA0>() for the component types of alpha / beta
template <typename R, typename F>
int32_t dispatch_c(const ScalArgs &s, F &&f) {
  const bool b0 = s.bre == 0 && s.bim == 0;   // β == zero(T2)
  if constexpr (sizeof(R) == 8) {
    return b0 ? f.template operator()<double, double, true>() : f.template operator()<double, double, false>();
  } else {
    if (b0) return s.a64 ? f.template operator()<double, double, true>() : f.template operator()<float, float, true>();
    if (s.a64) return s.b64 ? f.template operator()<double, double, false>() : f.template operator()<double, float, false>();
    return s.b64 ? f.template operator()<float, double, false>() : f.template operator()<float, float, false>();
  }
}

}  // namespace mxlo
