// Device-side math substrate (fp64): SO3/SE3 (jni/RT.h), ATAN/FOV camera (jni/ATANCamera.{h,cc}),
// Tukey M-estimator (jni/MEstimator.h), small dense solves.  Pure functions, no state.
#pragma once
#include <hip/hip_runtime.h>
#include "vslam_libm.h"

#define DEVFN __device__ __forceinline__
#define HDFN __host__ __device__ __forceinline__

struct Pose { double R[9]; double t[3]; };   // camera-from-world, R row-major (mySE3, jni/RT.h:247-312)

struct CamModel {   // ATANCamera after RefreshParams (jni/ATANCamera.cc:37-82)
  double size[2], focal[2], center[2];
  double w, winv, two_tan, distortion_enabled;
  double largest_radius, max_r;
};

struct CamProj { double im[2]; double cam[2]; double r; double factor; int invalid; };

HDFN void pose_xform(const Pose& T, const double p[3], double o[3]) {   // jni/RT.h:492-499
  o[0] = T.t[0] + (T.R[0] * p[0] + T.R[1] * p[1] + T.R[2] * p[2]);
  o[1] = T.t[1] + (T.R[3] * p[0] + T.R[4] * p[1] + T.R[5] * p[2]);
  o[2] = T.t[2] + (T.R[6] * p[0] + T.R[7] * p[1] + T.R[8] * p[2]);
}
HDFN void pose_rot(const Pose& T, const double p[3], double o[3]) {
  o[0] = T.R[0] * p[0] + T.R[1] * p[1] + T.R[2] * p[2];
  o[1] = T.R[3] * p[0] + T.R[4] * p[1] + T.R[5] * p[2];
  o[2] = T.R[6] * p[0] + T.R[7] * p[1] + T.R[8] * p[2];
}
HDFN Pose pose_mul(const Pose& a, const Pose& b) {                       // jni/RT.h:286-295
  Pose r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      r.R[i * 3 + j] = a.R[i * 3 + 0] * b.R[0 * 3 + j] + a.R[i * 3 + 1] * b.R[1 * 3 + j] + a.R[i * 3 + 2] * b.R[2 * 3 + j];
  double rt[3];
  pose_rot(a, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
  return r;
}
HDFN Pose pose_inverse(const Pose& a) {                                 // jni/RT.h:274-282
  Pose r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.R[i * 3 + j] = a.R[j * 3 + i];
  double rt[3];
  pose_rot(r, a.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = -rt[i];
  return r;
}

HDFN void rodrigues(const double w[3], double A, double B, double R[9]) {   // jni/RT.h:98-129
  const double wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
  R[0] = 1.0 - B * (wy2 + wz2); R[4] = 1.0 - B * (wx2 + wz2); R[8] = 1.0 - B * (wx2 + wy2);
  { const double a = A * w[2], b = B * (w[0] * w[1]); R[1] = b - a; R[3] = b + a; }
  { const double a = A * w[1], b = B * (w[0] * w[2]); R[2] = b + a; R[6] = b - a; }
  { const double a = A * w[0], b = B * (w[1] * w[2]); R[5] = b - a; R[7] = b + a; }
}

HDFN void so3_exp(const double w[3], double R[9]) {                      // jni/RT.h:134-165
  const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
  const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double theta = sqrt(theta_sq);
  double A, B;
  if (theta_sq < 1e-8) { A = 1.0 - one_6th * theta_sq; B = 0.5; }
  else if (theta_sq < 1e-6) { B = 0.5 - 0.25 * one_6th * theta_sq; A = 1.0 - theta_sq * one_6th * (1.0 - one_20th * theta_sq); }
  else { const double inv_theta = 1.0 / theta; A = vlm::vsin(theta) * inv_theta; B = (1 - vlm::vcos(theta)) * (inv_theta * inv_theta); }
  rodrigues(w, A, B, R);
}

HDFN void so3_ln(const double M[9], double result[3]) {                  // jni/RT.h:167-214
  const double kSqrt1_2 = 0.70710678118654752440, kPi = 3.14159265358979323846;
  const double cos_angle = (M[0] + M[4] + M[8] - 1.0) * 0.5;
  result[0] = (M[7] - M[5]) / 2; result[1] = (M[2] - M[6]) / 2; result[2] = (M[3] - M[1]) / 2;
  const double sin_angle_abs = sqrt(result[0] * result[0] + result[1] * result[1] + result[2] * result[2]);
  if (cos_angle > kSqrt1_2) {
    if (sin_angle_abs > 0) { const double f = vlm::vasin(sin_angle_abs) / sin_angle_abs; result[0] *= f; result[1] *= f; result[2] *= f; }
  } else if (cos_angle > -kSqrt1_2) {
    const double f = vlm::vacos(cos_angle) / sin_angle_abs;
    result[0] *= f; result[1] *= f; result[2] *= f;
  } else {
    const double angle = kPi - vlm::vasin(sin_angle_abs);
    const double d0 = M[0] - cos_angle, d1 = M[4] - cos_angle, d2 = M[8] - cos_angle;
    double r2[3];
    if (d0 * d0 > d1 * d1 && d0 * d0 > d2 * d2) { r2[0] = d0; r2[1] = (M[3] + M[1]) / 2; r2[2] = (M[2] + M[6]) / 2; }
    else if (d1 * d1 > d2 * d2) { r2[0] = (M[3] + M[1]) / 2; r2[1] = d1; r2[2] = (M[7] + M[5]) / 2; }
    else { r2[0] = (M[2] + M[6]) / 2; r2[1] = (M[7] + M[5]) / 2; r2[2] = d2; }
    if (r2[0] * result[0] + r2[1] * result[1] + r2[2] * result[2] < 0) { r2[0] = -r2[0]; r2[1] = -r2[1]; r2[2] = -r2[2]; }
    const double inv = 1.0 / sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
    for (int i = 0; i < 3; i++) result[i] = (r2[i] * inv) * angle;
  }
}

HDFN Pose se3_exp(const double mu[6]) {                                  // jni/RT.h:318-352
  const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
  Pose result;
  const double w[3] = {mu[3], mu[4], mu[5]};
  const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double theta = sqrt(theta_sq);
  double A, B;
  const double cr[3] = {w[1] * mu[2] - w[2] * mu[1], w[2] * mu[0] - w[0] * mu[2], w[0] * mu[1] - w[1] * mu[0]};
  if (theta_sq < 1e-8) {
    A = 1.0 - one_6th * theta_sq; B = 0.5;
    for (int i = 0; i < 3; i++) result.t[i] = mu[i] + 0.5 * cr[i];
  } else {
    double C;
    if (theta_sq < 1e-6) { C = one_6th * (1.0 - one_20th * theta_sq); A = 1.0 - theta_sq * C; B = 0.5 - 0.25 * one_6th * theta_sq; }
    else { const double inv_theta = 1.0 / theta; A = vlm::vsin(theta) * inv_theta; B = (1 - vlm::vcos(theta)) * (inv_theta * inv_theta); C = (1 - A) * (inv_theta * inv_theta); }
    const double wc[3] = {w[1] * cr[2] - w[2] * cr[1], w[2] * cr[0] - w[0] * cr[2], w[0] * cr[1] - w[1] * cr[0]};
    for (int i = 0; i < 3; i++) result.t[i] = mu[i] + B * cr[i] + C * wc[i];
  }
  rodrigues(w, A, B, result.R);
  return result;
}

HDFN void se3_ln(const Pose& T, double out[6]) {                         // jni/RT.h:354-383
  double rotv[3];
  so3_ln(T.R, rotv);
  const double rr = rotv[0] * rotv[0] + rotv[1] * rotv[1] + rotv[2] * rotv[2];
  const double theta = sqrt(rr);
  double shtot = 0.5;
  if (theta > 0.00001) shtot = vlm::vsin(theta / 2) / theta;
  const double half[3] = {rotv[0] * -0.5, rotv[1] * -0.5, rotv[2] * -0.5};
  Pose hr;
  so3_exp(half, hr.R);
  double rottrans[3];
  pose_rot(hr, T.t, rottrans);
  const double tdr = T.t[0] * rotv[0] + T.t[1] * rotv[1] + T.t[2] * rotv[2];
  const double f = theta > 0.001 ? (tdr * (1 - 2 * shtot) / rr) : (tdr / 24);
  const double inv = 1.0 / (2 * shtot);
  for (int i = 0; i < 3; i++) out[i] = (rottrans[i] - rotv[i] * f) * inv;
  out[3] = rotv[0]; out[4] = rotv[1]; out[5] = rotv[2];
}

// mySE3::generator_field(i, (x, y, z, 1)), jni/RT.h:297-308
HDFN void generator_field(int i, const double pos[3], double out[3]) {
  out[0] = out[1] = out[2] = 0;
  if (i < 3) { out[i] = 1.0; return; }
  out[(i + 1) % 3] = -pos[(i + 2) % 3];
  out[(i + 2) % 3] = pos[(i + 1) % 3];
}

// Image-plane motion (z = 1 plane) of camera-frame point c under generator k of the left-multiplied se3 update:
// (mot.xy - c.xy * mot.z / c.z) / c.z with mot = generator_field(k, c) (jni/TrackerData.h:112-118, jni/Bundle.cc:279-288).
// Same expression tree as the reference for every k; written per generator so nothing is indexed dynamically
// (a dynamically indexed local array lives in scratch memory on the GPU).
HDFN void se3_generator_motion(int k, const double c[3], double ooz, double& f0, double& f1) {
  double m0, m1, m2;
  switch (k) {
    case 0: m0 = 1.0; m1 = 0.0; m2 = 0.0; break;
    case 1: m0 = 0.0; m1 = 1.0; m2 = 0.0; break;
    case 2: m0 = 0.0; m1 = 0.0; m2 = 1.0; break;
    case 3: m0 = 0.0; m1 = -c[2]; m2 = c[1]; break;
    case 4: m0 = c[2]; m1 = 0.0; m2 = -c[0]; break;
    default: m0 = -c[1]; m1 = c[0]; m2 = 0.0; break;
  }
  f0 = (m0 - c[0] * m2 * ooz) * ooz;
  f1 = (m1 - c[1] * m2 * ooz) * ooz;
}

// ---- camera -------------------------------------------------------------------------------------------------
HDFN double cam_rtrans_factor(const CamModel& c, double r) {              // jni/ATANCamera.h:136-142
  if (r < 0.001 || c.w == 0.0) return 1.0;
  return c.winv * vlm::vatan(r * c.two_tan) / r;
}
HDFN CamProj cam_project(const CamModel& c, double cx, double cy) {        // jni/ATANCamera.cc:133-145
  CamProj p;
  p.cam[0] = cx; p.cam[1] = cy;
  p.r = sqrt(cx * cx + cy * cy);
  p.invalid = (p.r > c.max_r);
  p.factor = cam_rtrans_factor(c, p.r);
  p.im[0] = c.center[0] + c.focal[0] * (cx * p.factor);
  p.im[1] = c.center[1] + c.focal[1] * (cy * p.factor);
  return p;
}
HDFN void cam_derivs(const CamModel& c, const CamProj& p, double d[4]) {   // jni/ATANCamera.cc:198-231
  double fx, fy;
  const double k = c.two_tan, x = p.cam[0], y = p.cam[1];
  const double r = p.r * c.distortion_enabled;
  if (r < 0.01) { fx = 0.0; fy = 0.0; }
  else {
    fx = c.winv * (k * x) / (r * r * (1 + k * k * r * r)) - x * p.factor / (r * r);
    fy = c.winv * (k * y) / (r * r * (1 + k * k * r * r)) - y * p.factor / (r * r);
  }
  d[0] = c.focal[0] * (fx * x + p.factor);
  d[2] = c.focal[1] * (fx * y);
  d[1] = c.focal[0] * (fy * x);
  d[3] = c.focal[1] * (fy * y + p.factor);
}

// ATANCamera::UnProject, jni/ATANCamera.cc:149-164
HDFN void cam_unproject(const CamModel& c, double ix, double iy, double out[2]) {
  const double dx = (ix - c.center[0]) * (1.0 / c.focal[0]), dy = (iy - c.center[1]) * (1.0 / c.focal[1]);
  const double dist_r = sqrt(dx * dx + dy * dy);
  const double r = c.w == 0.0 ? dist_r : vlm::vtan(dist_r * c.w) * (1.0 / c.two_tan);
  const double f = dist_r > 0.01 ? r / dist_r : 1.0;
  out[0] = dx * f; out[1] = dy * f;
}

// ---- Tukey (jni/MEstimator.h:42-77) --------------------------------------------------------------------------
HDFN double tukey_sqrt_weight(double e2, double s2) { return e2 > s2 ? 0.0 : 1.0 - (e2 / s2); }
HDFN double tukey_weight(double e2, double s2) { const double d = tukey_sqrt_weight(e2, s2); return d * d; }
HDFN double tukey_objective(double e2, double s2) { if (e2 > s2) return 1.0; const double d = 1.0 - e2 / s2; return 1.0 - d * d * d; }
HDFN double tukey_sigma_squared(double median, unsigned long n) {
  double sigma = 1.4826 * (1 + 5.0 / (n * 2 - 6)) * sqrt(median);   // size_t arithmetic as in the reference
  sigma = 4.6851 * sigma;
  return sigma * sigma;
}

// ---- dense solves --------------------------------------------------------------------------------------------
// Gaussian elimination with partial pivoting (stands in for Eigen's PartialPivLU  A.inverse()*b). A n x n row-major.
HDFN bool lu_solve_n(double* A, double* b, int n) {
  for (int k = 0; k < n; k++) {
    int piv = k; double best = fabs(A[k * n + k]);
    for (int r = k + 1; r < n; r++) if (fabs(A[r * n + k]) > best) { best = fabs(A[r * n + k]); piv = r; }
    if (best == 0.0) return false;
    if (piv != k) { for (int c = 0; c < n; c++) { const double t = A[k * n + c]; A[k * n + c] = A[piv * n + c]; A[piv * n + c] = t; } const double t = b[k]; b[k] = b[piv]; b[piv] = t; }
    const double inv = 1.0 / A[k * n + k];
    for (int r = k + 1; r < n; r++) {
      const double f = A[r * n + k] * inv;
      if (f == 0.0) continue;
      for (int c = k + 1; c < n; c++) A[r * n + c] -= f * A[k * n + c];
      b[r] -= f * b[k];
    }
  }
  for (int k = n - 1; k >= 0; k--) {
    double s = b[k];
    for (int c = k + 1; c < n; c++) s -= A[k * n + c] * b[c];
    b[k] = s / A[k * n + k];
  }
  return true;
}

// 6 x 6 version of lu_solve_n with every index a compile-time constant after unrolling, so A and b stay in registers
// (a dynamically indexed local array lives in scratch memory: ~1 us per dependent access on one lane).  Same pivot rule.
HDFN bool lu_solve6(double A[36], double b[6]) {
#pragma unroll
  for (int k = 0; k < 6; k++) {
    int piv = k; double best = fabs(A[k * 6 + k]);
#pragma unroll
    for (int r = k + 1; r < 6; r++) { const double a = fabs(A[r * 6 + k]); if (a > best) { best = a; piv = r; } }
    if (best == 0.0) return false;
#pragma unroll
    for (int r = k + 1; r < 6; r++) {
      const bool sw = (r == piv);
#pragma unroll
      for (int c = 0; c < 6; c++) { const double x = A[k * 6 + c], y = A[r * 6 + c]; A[k * 6 + c] = sw ? y : x; A[r * 6 + c] = sw ? x : y; }
      const double x = b[k], y = b[r]; b[k] = sw ? y : x; b[r] = sw ? x : y;
    }
    const double inv = 1.0 / A[k * 6 + k];
#pragma unroll
    for (int r = k + 1; r < 6; r++) {
      const double f = A[r * 6 + k] * inv;
#pragma unroll
      for (int c = k + 1; c < 6; c++) A[r * 6 + c] -= f * A[k * 6 + c];
      b[r] -= f * b[k];
    }
  }
#pragma unroll
  for (int k = 5; k >= 0; k--) {
    double s = b[k];
#pragma unroll
    for (int c = k + 1; c < 6; c++) s -= A[k * 6 + c] * b[c];
    b[k] = s / A[k * 6 + k];
  }
  return true;
}

HDFN void inv3(const double m[9], double o[9]) {   // cofactor inverse (Eigen fixed 3x3)
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
HDFN void inv2(const double m[4], double o[4]) {
  const double id = 1.0 / (m[0] * m[3] - m[1] * m[2]);
  o[0] = m[3] * id; o[1] = -m[1] * id; o[2] = -m[2] * id; o[3] = m[0] * id;
}

#ifdef __HIPCC__
// Sum N (a power of two, <= 64) per-lane values over the 64 lanes of a wavefront by recursive halving: at distance d
// the lanes with bit d clear keep the lower half of the values and hand the upper half to their partner (and vice versa),
// so N/2 + N/4 + ... shuffles move the data instead of 6 N.  The pairing tree -- hence every rounding -- is that of
// the plain xor butterfly.  Returns the total of value wave_multi_index<N>(lane); acc is clobbered.
template <int N>
DEVFN double wave_multi_sum(double* acc) {
  const int lane = threadIdx.x & 63;
  int d = 32;
#pragma unroll
  for (int half = N / 2; half >= 1; half >>= 1, d >>= 1) {
    // bit-mask selects: a ?: on the array elements would be folded into a dynamically indexed (scratch) access
    const unsigned long long m = (lane & d) ? ~0ull : 0ull;
#pragma unroll
    for (int j = 0; j < half; j++) {
      const unsigned long long lo = (unsigned long long)__double_as_longlong(acc[j]), hi = (unsigned long long)__double_as_longlong(acc[j + half]);
      const unsigned long long x = (hi ^ lo) & m;
      const double send = __longlong_as_double((long long)(hi ^ x));   // upper lanes hand over the lower half
      const double keep = __longlong_as_double((long long)(lo ^ x));   // ... and keep the upper half
      acc[j] = keep + __shfl_xor(send, d);
    }
  }
#pragma unroll
  for (; d >= 1; d >>= 1) acc[0] += __shfl_xor(acc[0], d);
  return acc[0];
}
template <int N> DEVFN int wave_multi_index(int lane) { return N == 64 ? lane : (N == 32 ? (lane >> 1) & 31 : (N == 16 ? (lane >> 2) & 15 : (lane >> 3) & 7)); }

// k-th smallest (0-based) of n non-negative doubles in global memory or LDS (bit patterns order like the values):
// MSB-first radix select, 8 bits per pass, stopping as soon as the selected bin holds a single value (the usual case
// after 3 passes).  One barrier per pass: the histogram rotates through three LDS buffers (the one for pass p+2 is
// cleared while pass p is scanned) and every wavefront scans the 256 bins redundantly, so nothing is broadcast.
// hist: LDS [768] ints, sel: LDS [1] u64.  All threads of the workgroup call it and get the same result.
// The values are read RS_BATCH at a time per thread before any histogram update, so the round trips of a batch overlap
// (the pointer type is a template parameter: an address_space(1) pointer keeps the loads global instead of flat).
#define RS_BATCH 8
template <int BATCH = RS_BATCH, class VP>
DEVFN double block_radix_select(VP v, int n, int k, int* hist, unsigned long long* sel) {
  for (int i = threadIdx.x; i < 768; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  unsigned long long prefix = 0, mask = 0;
  int kk = k;
  const int l = threadIdx.x & 63;
  for (int pass = 0; pass < 8; pass++) {
    const int shift = 56 - 8 * pass;
    int* H = hist + 256 * (pass % 3);
    for (int i0 = threadIdx.x; i0 < n; i0 += BATCH * blockDim.x) {
      unsigned long long b[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; u++) { const int i = i0 + u * blockDim.x; b[u] = (unsigned long long)__double_as_longlong(v[i < n ? i : n - 1]); }
#pragma unroll
      for (int u = 0; u < BATCH; u++) if (i0 + u * (int)blockDim.x < n && (b[u] & mask) == prefix) atomicAdd(&H[(b[u] >> shift) & 255], 1);
    }
    __syncthreads();
    const int h0 = H[4 * l], h1 = H[4 * l + 1], h2 = H[4 * l + 2], h3 = H[4 * l + 3];   // 4 bins per lane, shuffle prefix sum
    const int tot = h0 + h1 + h2 + h3;
    int inc = tot;
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (l >= d) inc += t; }
    const unsigned long long bm = __ballot(inc > kk);              // first lane whose inclusive prefix passes kk
    const int L = __ffsll((long long)bm) - 1;
    int acc = inc - tot, bin = 4 * l, cnt = h0;
    if (acc + h0 > kk) { }
    else if (acc + h0 + h1 > kk) { acc += h0; bin += 1; cnt = h1; }
    else if (acc + h0 + h1 + h2 > kk) { acc += h0 + h1; bin += 2; cnt = h2; }
    else { acc += h0 + h1 + h2; bin += 3; cnt = h3; }
    bin = __shfl(bin, L); acc = __shfl(acc, L); cnt = __shfl(cnt, L);
    prefix |= (unsigned long long)bin << shift;
    kk -= acc;
    mask |= 255ull << shift;
    int* Z = hist + 256 * ((pass + 2) % 3);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) Z[i] = 0;
    if (cnt == 1 && pass < 7) {                                    // one value left under the prefix: fetch it whole
      for (int i0 = threadIdx.x; i0 < n; i0 += BATCH * blockDim.x) {
        unsigned long long b[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; u++) { const int i = i0 + u * blockDim.x; b[u] = (unsigned long long)__double_as_longlong(v[i < n ? i : n - 1]); }
#pragma unroll
        for (int u = 0; u < BATCH; u++) if (i0 + u * (int)blockDim.x < n && (b[u] & mask) == prefix) sel[0] = b[u];
      }
      __syncthreads();
      prefix = sel[0];
      break;
    }
  }
  return __longlong_as_double((long long)prefix);
}

#endif

HDFN double level_zero_pos(double p, int l) { return (p + 0.5) * (1 << l) - 0.5; }   // jni/LevelHelpers.h:23-25
HDFN double level_n_pos(double p, int l) { return (p + 0.5) / (1 << l) - 0.5; }      // jni/LevelHelpers.h:37-39
