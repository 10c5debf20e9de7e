// Branch-free fp64 cosine shared by the RFF kernels (rff.hip) and the RFF ascent (meangrad.hip).
#pragma once
#include "common.h"

namespace {

// cos(x) for |x| < 1.6e6 as (-1)^k sin(r), |x| = (2k - 1) pi/2 + r with |r| <= pi/2, k = rint(|x|/pi + 1/2):
// Cody-Waite reduction by three 33-bit pieces of pi/2 (the odd multiplier 2k-1 < 2^21 is exact against each), then
// ONE odd minimax polynomial sin(r) = r P(r^2), nine coefficients (fit error 2e-19, <= 1 ulp in double; the recipe
// is tools/expfit.py sin_fit) -- 17 fp64 instructions + 3 integer ones for the sign (the parity of k moved into
// the sign bit), against ~40 for a library cosine that reduces to pi/4 and evaluates a sine AND a cosine kernel.
// The RFF phases w.x + b are O(sqrt(D)/l) (tens), far inside the fast range; larger arguments take the library
// path behind a wave-uniform branch (rff_cos_slow is out of line: one copy per kernel, not one per element).
constexpr double RFF_COS_FAST_RANGE = 1.6e6;
__device__ __attribute__((noinline)) double rff_cos_slow(double x) { return cos(x); }

struct RffPoly {   // c[k] = amplitude * (coefficient of r^(2k+1)), prepared on the host: the feature scale rides along
  double c[9];
};
static inline RffPoly make_rff_poly(double amplitude) {
  static const double s[9] = {0x1.0000000000000p+0,  -0x1.5555555555555p-3, 0x1.11111111110bcp-7,
                              -0x1.a01a01a0147d9p-13, 0x1.71de3a528c5e5p-19, -0x1.ae6454d01e7a1p-26,
                              0x1.6123ccc2fc0b0p-33,  -0x1.ae4398eddfa1fp-41, 0x1.8837bd66b70acp-49};
  RffPoly p;
  for (int k = 0; k < 9; ++k) p.c[k] = amplitude * s[k];
  return p;
}

// amplitude * cos(x); branch-free, valid for |x| < RFF_COS_FAST_RANGE: independent evaluations interleave
__device__ __forceinline__ double rff_cos_fast(double x, const RffPoly& P) {
  const double ax = fabs(x);
  const double kf = rint(fma(ax, 3.18309886183790671538e-01, 0.5));
  const double n = fma(2.0, kf, -1.0);
  double r = fma(-n, 1.57079632673412561417e+00, ax);
  r = fma(-n, 6.07710050630396597660e-11, r);
  r = fma(-n, 2.02226624871116645580e-21, r);
  const double z = r * r;
  double q = P.c[8];
  q = fma(q, z, P.c[7]);
  q = fma(q, z, P.c[6]);
  q = fma(q, z, P.c[5]);
  q = fma(q, z, P.c[4]);
  q = fma(q, z, P.c[3]);
  q = fma(q, z, P.c[2]);
  q = fma(q, z, P.c[1]);
  q = fma(q, z, P.c[0]);
  const double sn = r * q;
  // (-1)^k: the parity of k goes straight into the sign bit
  const int flip = ((int)kf) << 31;
  return __hiloint2double(__double2hiint(sn) ^ flip, __double2loint(sn));
}

}  // namespace
