// m3t_exact_math.h -- the two transcendental spots of the constraint code besides the logarithm:
//   Eigen::AngleAxisf(Matrix3f)::angle() = 2 atan2(|q.vec|, |q.w|)          (constraint.cpp:177,219; soft_constraint.cpp:276,309)
//   xcotx(x) = x / tan(x) with its two tanf() range tests                    (common.h:73-77)
// glibc's atan2f / tanf / tan and ocml's differ in the last bit, and the KKT systems of closed chains amplify that to
// 2e-5 within a few frames.  Here both are taken THROUGH f64 -- "the f32 nearest to the f64 value", as m3t_log.h does for
// the logarithm -- by ONE implementation made of IEEE + - x / only (no libm, no fma, -ffp-contract=off on both sides),
// included by the kernels (m3t_links.hip) and by the oracle (oracle/m3t_oracle.cpp): the same operations in the same
// order on both sides, hence the same bits.  That it is also the right function is checked on the host against glibc
// (tests/cpp/exact_math_check.cpp): xcotx for every float in [0, fl(pi/2)], atan2 over 10^8 pairs.
//
// Domains are those the callers can produce: atan2 with y >= 0, x >= 0 (a norm and an absolute value), tan / xcotx with
// 0 <= x <= 2 (half of an angle in [0, fl(pi)]); anything else, NaN included, returns NaN.
#pragma once
#include <stdint.h>
#include <string.h>

#include "m3t_exact_math_table.h"

#if defined(__HIPCC__)
#define M3T_XM_FN __device__ inline
#define M3T_XM_TABLE __device__ const
#else
#define M3T_XM_FN static inline
#define M3T_XM_TABLE static const
#endif

M3T_XM_TABLE uint64_t m3t_xm_atan_table[17] = M3T_XM_ATAN_TABLE_INIT;  // atan(i / 16)
M3T_XM_TABLE uint64_t m3t_xm_tan_table[27] = M3T_XM_TAN_TABLE_INIT;    // tan(i / 32)

M3T_XM_FN double m3t_xm_bits(uint64_t b) {
  double d;
  memcpy(&d, &b, sizeof d);
  return d;
}
M3T_XM_FN double m3t_xm_nan() { return m3t_xm_bits(0x7ff8000000000000ull); }

// atan2(y, x) for y >= 0, x >= 0 in f64, a few 2^-53 from the true value: quotient of the smaller by the larger, the
// nearest table point c = i / 16, atan(t) = atan(c) + atan((t - c) / (1 + t c)), odd polynomial to r^13 on |r| <= 1/32
M3T_XM_FN double m3t_atan2_pos(double y, double x) {
  if (!(y >= 0.0) || !(x >= 0.0)) return m3t_xm_nan();
  if (y == 0.0) return 0.0;  // (x = 0 as well: atan2(0, 0) = 0)
  const bool swap = y > x;
  const double num = swap ? x : y, den = swap ? y : x;
  const double pio2_hi = m3t_xm_bits(M3T_XM_PIO2_HI_BITS), pio2_lo = m3t_xm_bits(M3T_XM_PIO2_LO_BITS);
  if (den > 1.0e300) {  // infinity (the callers pass floats): the quotient below would be NaN or 0
    if (num > 1.0e300) return 0.5 * pio2_hi;
    return swap ? pio2_hi : 0.0;
  }
  const double t = num / den;  // [0, 1]
  const int i = (int)(t * 16.0 + 0.5);
  const double c = (double)i * 0.0625;
  const double r = (t - c) / (1.0 + t * c);
  const double r2 = r * r;
  double p = 1.0 / 13.0;
  p = p * r2 - 1.0 / 11.0;
  p = p * r2 + 1.0 / 9.0;
  p = p * r2 - 1.0 / 7.0;
  p = p * r2 + 1.0 / 5.0;
  p = p * r2 - 1.0 / 3.0;
  const double a = m3t_xm_bits(m3t_xm_atan_table[i]) + (r + (p * r2) * r);
  return swap ? (pio2_hi - a) + pio2_lo : a;
}
// Eigen's angle(): 2 * atan2f(n, |w|), the atan2 rounded to f32 through f64
M3T_XM_FN float m3t_atan2f_pos(float y, float x) { return (float)m3t_atan2_pos((double)y, (double)x); }

// tan(r) for |r| <= 0.85: nearest table point a = i / 32, tan(a + b) = (tan a + tan b) / (1 - tan a tan b), odd
// polynomial to b^11 on |b| <= 1/64
M3T_XM_FN double m3t_xm_tan_reduced(double r) {
  const bool neg = r < 0.0;
  if (neg) r = -r;
  const int i = (int)(r * 32.0 + 0.5);  // 0 .. 27 -> the callers keep r <= 0.8: i <= 26
  const double b = r - (double)i * 0.03125;  // exact
  const double b2 = b * b;
  double p = 1382.0 / 155925.0;
  p = p * b2 + 62.0 / 2835.0;
  p = p * b2 + 17.0 / 315.0;
  p = p * b2 + 2.0 / 15.0;
  p = p * b2 + 1.0 / 3.0;
  const double tb = b + (p * b2) * b;
  const double ta = m3t_xm_bits(m3t_xm_tan_table[i]);
  const double t = (ta + tb) / (1.0 - ta * tb);
  return neg ? -t : t;
}
// tan(x) in f64 for 0 <= x <= 2 (x = pi/2 cannot happen: not a double)
M3T_XM_FN double m3t_tan(double x) {
  if (!(x >= 0.0 && x <= 2.0)) return m3t_xm_nan();
  if (x <= 0.8) return m3t_xm_tan_reduced(x);
  const double r = (m3t_xm_bits(M3T_XM_PIO2_HI_BITS) - x) + m3t_xm_bits(M3T_XM_PIO2_LO_BITS);  // in [-0.43, 0.78]
  return 1.0 / m3t_xm_tan_reduced(r);
}
// common.h:73-77: tanf(x) <= FLT_MIN -> 1, tanf(x) >= FLT_MAX -> 0, else x / tan(x) (the quotient in double, as the
// unqualified tan() of the reference takes it); tanf = the f32 nearest to the f64 value
M3T_XM_FN float m3t_xcotx(float x) {
  const double t = m3t_tan((double)x);
  const float tf = (float)t;
  if (tf <= 1.17549435e-38f) return 1.0f;
  if (tf >= 3.40282347e+38f) return 0.0f;
  return (float)((double)x / t);
}
