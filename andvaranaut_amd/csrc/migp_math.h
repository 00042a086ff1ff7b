// Device math shared by the covariance assembly (assemble.hip) and the kernels that differentiate it (grad_predict.hip):
// K and dK/dtheta are built from the SAME exp / sqrt sequences, so they are bit-consistent with each other.
#pragma once
#include <hip/hip_runtime.h>

namespace migp {

// exp(x) for x <= 0 (any magnitude, -inf included): Cody-Waite reduction by ln 2 in two pieces, degree-13 Taylor polynomial
// on |r| <= ln2 / 2 (remainder 4e-18 relative), ldexp.  Max error 1.0 ulp against libm over [-745, 0]; arguments below
// -746 (exp underflows to 0 from -745.13 on) are clamped there, so -inf -- r2 = inf from an underflowing length scale --
// gives 0 like libm, not NaN (the unclamped reduction computed inf - inf).  That holds for the RBF family only: the Matern /
// Exponential argument goes through sqrt_pos first, and sqrt_pos(inf) is NaN.  fmax also turns a NaN argument into -746:
// NaN does NOT propagate through this function, which is why the host rejects non-finite theta (factor_internal) and
// non-finite X / y (MiGP.__init__, MiGP.update_data) before anything reaches the device.
__device__ __forceinline__ double exp_nonpos(double x) {
  x = __builtin_fmax(x, -746.0);
  const double k = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(-k, 0.6931471803691238, x);
  r = __builtin_fma(-k, 1.9082149292705877e-10, r);
  double p = 1.6059043836821613e-10;                       // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);          // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);         // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);         // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06);        // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);          // 1/8!
  p = __builtin_fma(p, r, 0.0001984126984126984);         // 1/7!
  p = __builtin_fma(p, r, 0.001388888888888889);          // 1/6!
  p = __builtin_fma(p, r, 0.008333333333333333);          // 1/5!
  p = __builtin_fma(p, r, 0.041666666666666664);          // 1/4!
  p = __builtin_fma(p, r, 0.16666666666666666);           // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)k);                // k >= -1077: 2^-1077 p rounds to 0
}

// sqrt(t) for normal t > 0: v_rsq_f64 seed (measured 2^-24.2 relative, tools/probe_rcp.hip), ONE Goldschmidt step
// (g, h to ~2^-48) and one Newton correction g + (t - g^2) h, whose error is the PRODUCT of the two (2^-96) plus the
// final rounding: <= 1 ulp against libm (tools/probe_math.hip).  Round 2 ran two Goldschmidt steps (three more FMAs).
__device__ __forceinline__ double sqrt_pos(double t) {
  const double y = __builtin_amdgcn_rsq(t);
  double g = t * y, h = 0.5 * y;
  const double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  const double dd = __builtin_fma(-g, g, t);
  return __builtin_fma(dd, h, g);
}

}  // namespace migp
