// Transcendentals of the camera model and the Lie-group maps (sin, cos, tan, atan, asin, acos) written with
// + - * / sqrt, comparisons and conversions only, so that the SAME source evaluates to the SAME bits on the host
// (g++, the oracle) and on gfx950 (hipcc) when both are built with -ffp-contract=off: IEEE-754 binary64 add, multiply,
// divide and square root are correctly rounded on both.  The vendor libms (device OCML vs glibc) differ in the last
// bit, and PTAM's templates are trunc(bilinear sample) (jni/vision/ImageHandler.cpp:12-19): one ulp in a projection is
// enough to flip a template pixel.  Consumers: ATANCamera (jni/ATANCamera.h:136-150, jni/ATANCamera.cc:133-164),
// mySO3/mySE3 exp and ln (jni/RT.h:134-214, 318-383), mySO2::exp (jni/RT.h:459-465).
//
// The algorithms are the classical argument-reduction + minimax-polynomial ones of the freely distributable fdlibm 5.3
// (which is also what the reference's Android libm, bionic, derives from); the polynomial coefficients and the split
// constants of pi are fdlibm's:
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at SunSoft, a Sun Microsystems, Inc.
//   business.  Permission to use, copy, modify, and distribute this software is freely granted, provided that this
//   notice is preserved.
// Nothing here reads the bits of a double: the "clear the low word" steps of fdlibm are conversions through float.
// Error about 1 ulp (tan about 2).  Domain of the trigonometric functions: |x| < 8e5 (beyond that the result is that of
// a plain, inexact reduction; NaN from 2^51 on and for infinities); the angles of this path are rotations per frame and image-plane radii.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define VLM_FN __host__ __device__ __forceinline__
#else
#define VLM_FN inline
#endif

namespace vlm {

VLM_FN double vabs(double x) { return x < 0.0 ? -x : x; }
// The NaN of a domain error: a constant (0 / 0 evaluates to the negative default NaN on x86 and to the positive one on gfx950)
VLM_FN double vnan() { return __builtin_nan(""); }
// sin / cos / tan of a non-finite argument or of |x| >= 2^51 (where the integer conversions below would be undefined on the host
// and saturating on the device) are domain errors here: the angles of this path are rotations per frame and image-plane radii
VLM_FN bool vtrig_domain(double x) { return vabs(x) < 2251799813685248.0; }
// floor for |x| < 2^51 through the integer conversion (truncation toward zero is exact on both sides)
VLM_FN double vfloor(double x) { const double t = (double)(long long)x; return t > x ? t - 1.0 : t; }

// sin on [-pi/4, pi/4] of x + y (y the tail of the reduced argument)
VLM_FN double ksin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double z = x * x, v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
// cos on [-pi/4, pi/4] of x + y
VLM_FN double kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = x * x;
  const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const double ax = vabs(x);
  if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
  const double qx = ax > 0.78125 ? 0.28125 : (double)(float)(0.25 * ax);   // about x / 4 with a short significand: 1 - qx is exact
  const double hz = 0.5 * z - qx, a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}
// x = n * pi/2 + (y0 + y1), |y0 + y1| <= pi/4 (+ rounding); returns n mod 4.  Three-term Cody-Waite reduction with
// pi/2 in 33-bit pieces: n * piece is exact for |n| < 2^20.
VLM_FN int rem_pio2(double x, double& y0, double& y1) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
  const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
  const double pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;
  double fn = vfloor(x * invpio2 + 0.5);
  if (vabs(fn) > 524288.0) {                       // out of the documented domain: plain reduction, inexact but finite
    const double twopi = 6.28318530717958623200e+00;
    x = x - vfloor(x / twopi) * twopi;
    fn = vfloor(x * invpio2 + 0.5);
  }
  double r = x - fn * pio2_1, w = fn * pio2_1t, t;
  t = r; w = fn * pio2_2; r = t - w; w = fn * pio2_2t - ((t - r) - w);
  t = r; w = fn * pio2_3; r = t - w; w = fn * pio2_3t - ((t - r) - w);
  y0 = r - w;
  y1 = (r - y0) - w;
  const long long n = (long long)fn;
  return (int)(n & 3);
}

VLM_FN double vsin(double x) {
  if (x != x) return x;
  if (!vtrig_domain(x)) return vnan();
  if (vabs(x) < 0.78539816339744830962) {
    if (vabs(x) < 7.450580596923828125e-9) return x;             // 2^-27
    return ksin(x, 0.0);
  }
  double y0, y1;
  const int n = rem_pio2(x, y0, y1);
  switch (n) {
    case 0: return ksin(y0, y1);
    case 1: return kcos(y0, y1);
    case 2: return -ksin(y0, y1);
    default: return -kcos(y0, y1);
  }
}
VLM_FN double vcos(double x) {
  if (x != x) return x;
  if (!vtrig_domain(x)) return vnan();
  if (vabs(x) < 0.78539816339744830962) {
    if (vabs(x) < 7.450580596923828125e-9) return 1.0;
    return kcos(x, 0.0);
  }
  double y0, y1;
  const int n = rem_pio2(x, y0, y1);
  switch (n) {
    case 0: return kcos(y0, y1);
    case 1: return -ksin(y0, y1);
    case 2: return -kcos(y0, y1);
    default: return ksin(y0, y1);
  }
}
VLM_FN double vtan(double x) {
  if (x != x) return x;
  if (!vtrig_domain(x)) return vnan();
  if (vabs(x) < 7.450580596923828125e-9) return x;
  double y0 = x, y1 = 0.0;
  int n = 0;
  if (!(vabs(x) < 0.78539816339744830962)) n = rem_pio2(x, y0, y1);
  const double s = ksin(y0, y1), c = kcos(y0, y1);
  return (n & 1) ? -c / s : s / c;
}

VLM_FN double vatan(double x) {
  if (x != x) return x;
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
               aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
               aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
  const bool neg = x < 0.0;
  double ax = vabs(x), hi = 0.0, lo = 0.0;
  bool direct = false;
  if (ax >= 7.3786976294838206464e19) {                          // 2^66
    const double r = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return neg ? -r : r;
  }
  if (ax < 0.4375) {
    if (ax < 1.862645149230957e-9) return x;                     // 2^-29
    direct = true;
  } else if (ax < 1.1875) {
    if (ax < 0.6875) { hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; ax = (2.0 * ax - 1.0) / (2.0 + ax); }
    else { hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; ax = (ax - 1.0) / (ax + 1.0); }
  } else if (ax < 2.4375) { hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; ax = (ax - 1.5) / (1.0 + 1.5 * ax); }
  else { hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; ax = -1.0 / ax; }
  const double z = ax * ax, w = z * z;
  const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  double r;
  if (direct) r = ax - ax * (s1 + s2);
  else r = hi - ((ax * (s1 + s2) - lo) - ax);
  return neg ? -r : r;
}

// the rational correction shared by asin and acos: R(t) = p(t) / q(t)
VLM_FN double asin_r(double t) {
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
  const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  const double p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
  const double q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
  return p / q;
}

VLM_FN double vasin(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pio4_hi = 7.85398163397448278999e-01;
  if (x != x) return x;
  const double ax = vabs(x);
  if (ax >= 1.0) {
    if (ax == 1.0) return x * pio2_hi + x * pio2_lo;
    return vnan();
  }
  if (ax < 0.5) {
    if (ax < 7.450580596923828125e-9) return x;
    return x + x * asin_r(x * x);
  }
  const double w = 1.0 - ax, t = w * 0.5, r = asin_r(t), s = sqrt(t);
  double res;
  if (ax >= 0.975) res = pio2_hi - (2.0 * (s + s * r) - pio2_lo);
  else {
    const double sw = (double)(float)s;                         // short significand: sw * sw is exact
    const double c = (t - sw * sw) / (s + sw);
    const double p = 2.0 * s * r - (pio2_lo - 2.0 * c);
    const double q = pio4_hi - 2.0 * sw;
    res = pio4_hi - (p - q);
  }
  return x < 0.0 ? -res : res;
}

VLM_FN double vacos(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
  if (x != x) return x;
  const double ax = vabs(x);
  if (ax >= 1.0) {
    if (x == 1.0) return 0.0;
    if (x == -1.0) return pi + 2.0 * pio2_lo;
    return vnan();
  }
  if (ax < 0.5) {
    if (ax <= 6.938893903907228e-18) return pio2_hi + pio2_lo;  // 2^-57
    const double r = asin_r(x * x);
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (x < 0.0) {
    const double z = (1.0 + x) * 0.5, s = sqrt(z), r = asin_r(z);
    const double w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  const double z = (1.0 - x) * 0.5, s = sqrt(z);
  const double df = (double)(float)s;
  const double c = (z - df * df) / (s + df);
  const double r = asin_r(z);
  const double w = r * s + c;
  return 2.0 * (df + w);
}

}  // namespace vlm
