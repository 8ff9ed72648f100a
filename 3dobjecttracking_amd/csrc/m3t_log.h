// m3t_log.h -- float(std::log(double(x))) for the distribution values of RegionModality::CalculateGradientAndHessian
// (region_modality.cpp:520-523), without the ~75 dependent f64 instructions of a general-purpose double logarithm.
//
// The parity contract of this spot is "the f32 nearest to log(x)", taken through f64 on both sides
// because glibc's logf and ocml's logf differ in the last bit.  Any f64 value within a few 2^-53 of log(x) rounds to
// the same f32 unless it lies right next to a rounding boundary.  m3t_log_fast computes log(x) in f64 to about 2^-44
// (64-entry table, log1p polynomial to r^6 on |r| <= 2^-7, the scheme of table-driven logf implementations with a
// longer table and polynomial), rounds it to f32 and CHECKS that the value moved down and up by 2^-42 rounds to
// the same f32; if not (about one call in 10^5), or if x is not a normal number in (0, 1], the caller takes the
// general double logarithm.  tests/cpp/log_check.cpp compares it with float(std::log(double(x))) for every one of
// the 1 056 964 609 floats in [FLT_MIN, 1] on the host; the device executes the same IEEE operations (explicit fma,
// -ffp-contract=off everywhere else).
#pragma once
#include <stdint.h>
#include <string.h>

#include "m3t_log_table.h"

#if defined(__HIPCC__)
#define M3T_LOG_FN __device__ __forceinline__
#else
#define M3T_LOG_FN static inline
#endif

#define M3T_LOG_TABLE_DOUBLES (2 << M3T_LOG_TABLE_BITS)

M3T_LOG_FN double m3t_log_bits_to_double(uint64_t b) {
  double d;
  memcpy(&d, &b, sizeof d);
  return d;
}

// table: M3T_LOG_TABLE_DOUBLES doubles {1 / c_i, log c_i} (m3t_log_table.h).  Returns false when the caller has to
// fall back to the general double logarithm.
template <typename TablePtr>
M3T_LOG_FN bool m3t_log_fast(float x, TablePtr table, float* out) {
  uint32_t ix;
  memcpy(&ix, &x, sizeof ix);
  if (ix - 0x00800000u > 0x3f800000u - 0x00800000u) return false;  // not a normal number in (0, 1]
  const uint32_t tmp = ix - M3T_LOG_OFF;
  const int i = (int)((tmp >> (23 - M3T_LOG_TABLE_BITS)) & ((1u << M3T_LOG_TABLE_BITS) - 1u));
  const int k = (int32_t)tmp >> 23;  // arithmetic shift: x = 2^k z, z in [OFF, 2 OFF)
  const uint32_t iz = ix - (tmp & 0xff800000u);
  float zf;
  memcpy(&zf, &iz, sizeof zf);
  const double z = (double)zf;
  const double invc = table[2 * i], logc = table[2 * i + 1];
  const double r = __builtin_fma(z, invc, -1.0);  // z / c_i - 1, |r| <= 2^-7
  const double y0 = __builtin_fma((double)k, m3t_log_bits_to_double(M3T_LOG_LN2_BITS), logc);
  const double r2 = r * r;
  double p = __builtin_fma(r, -1.0 / 6.0, 1.0 / 5.0);
  p = __builtin_fma(r, p, -1.0 / 4.0);
  p = __builtin_fma(r, p, 1.0 / 3.0);
  p = __builtin_fma(r, p, -1.0 / 2.0);
  const double y = __builtin_fma(r2, p, y0 + r);  // log c_i + k ln 2 + log1p(r)
  const float down = (float)(y * (1.0 - 0x1p-42)), up = (float)(y * (1.0 + 0x1p-42));
  if (down != up) return false;
  *out = down;
  return true;
}
