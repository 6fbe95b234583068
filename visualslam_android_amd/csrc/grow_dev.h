// Wavefront-level pieces of the map-maker's patch work shared by mapgrow.hip (AddPointEpipolar, ReFind_Common) and boot.hip
// (InitFromStereo): zero-mean SSD, the sub-pixel iterations, triangulation, unit rays.  See mapgrow.hip for the reference lines.
#pragma once
#include "vslam_internal.h"

DEVFN int wsum_i(int v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; }
DEVFN double wsum_d(double v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; }

// PatchFinder::ZMSSDAtPoint (jni/PatchFinder.cc:352-380) by one wavefront; the template lies in LDS
template <int PS>
DEVFN int wave_zmssd(const uint8_t* tmpl, const uint8_t* img, int ip, int wl, int hl, int cx, int cy, int tsum, int tsumsq, int max_ssd, int lane) {
  constexpr int NPIX = PS * PS, HALF = PS / 2;
  if (!(cx >= HALF && cy >= HALF && cx < wl - HALF && cy < hl - HALF)) return max_ssd + 1;
  int sA = 0, sQ = 0, sX = 0;
  for (int q = lane; q < NPIX; q += 64) {
    const int y = q / PS, x = q - y * PS;
    const int n = img[(size_t)(cy - HALF + y) * ip + (cx - HALF + x)], t = tmpl[q];
    sA += n; sQ += n * n; sX += n * t;
  }
  sA = wsum_i(sA); sQ = wsum_i(sQ); sX = wsum_i(sX);
  const int SA = tsum, SB = sA;
  return ((2 * SA * SB - SA * SA - SB * SB) / NPIX + sQ + tsumsq - 2 * sX);
}

// Several ZMSSDAtPoint evaluations (jni/PatchFinder.cc:352-380) per wavefront step, the way k_searchN scores its
// candidates: G lanes per patch (8 for 8x8 -> 8 patches at once, 16 for 11x11 -> 4), lane r < PS owns row r of template
// and image patch as packed bytes and the three sums are v_dot4_u32_u8 products reduced over the G lanes.
template <int PS> struct GRow { unsigned w[(PS + 3) / 4]; };
template <int PS> DEVFN GRow<PS> grow_load_row(const uint8_t* p) {
  GRow<PS> r;
  _Pragma("unroll") for (int k = 0; k < (PS + 3) / 4; k++) r.w[k] = 0u;
  __builtin_memcpy(&r, p, PS);
  return r;
}
template <int PS> DEVFN GRow<PS> grow_load_row_lds(const uint8_t* p) {      // the template row out of LDS (byte reads: 8-byte rows of an 11-byte pitch are unaligned)
  GRow<PS> r;
  _Pragma("unroll") for (int k = 0; k < (PS + 3) / 4; k++) r.w[k] = 0u;
  _Pragma("unroll") for (int x = 0; x < PS; x++) r.w[x >> 2] |= (unsigned)p[x] << (8 * (x & 3));
  return r;
}
template <int G> DEVFN int grow_grp_sum(int v) { for (int d = 1; d < G; d <<= 1) v += __shfl_xor(v, d); return v; }

// MakeSubPixTemplate (jni/PatchFinder.cc:242-271) + IterateSubPixToConvergence (:273-350) by one wavefront, starting from
// the level-zero position in sub0/sub1; these are left wherever the iteration stopped (ReFind_Common reads them regardless)
template <int PS>
DEVFN bool wave_subpix(const uint8_t* tmpl, const uint8_t* img, int ip, int wl, int hl, int nLevel, int max_its, int lane, double& sub0, double& sub1, double* slab /* LDS [3][Q*Q] of this wavefront */) {
  constexpr int HALF = PS / 2, Q = PS - 2, NQL = (Q * Q + 63) / 64;
  const int nLevelScale = 1 << nLevel;
  double gx[NQL], gy[NQL];
  double h00 = 0, h01 = 0, h02 = 0, h11 = 0, h12 = 0, h22 = 0;
  for (int q = 0; q < NQL; q++) {
    const int k = q * 64 + lane;
    gx[q] = 0; gy[q] = 0;
    if (k < Q * Q) {
      const int x = k / Q + 1, y = k % Q + 1;
      gx[q] = 0.5 * (tmpl[y * PS + x + 1] - tmpl[y * PS + x - 1]);
      gy[q] = 0.5 * (tmpl[(y + 1) * PS + x] - tmpl[(y - 1) * PS + x]);
      h00 += gx[q] * gx[q]; h01 += gx[q] * gy[q]; h02 += gx[q]; h11 += gy[q] * gy[q]; h12 += gy[q]; h22 += 1.0;
    }
  }
  h00 = wsum_d(h00); h01 = wsum_d(h01); h02 = wsum_d(h02); h11 = wsum_d(h11); h12 = wsum_d(h12); h22 = wsum_d(h22);
  const double H[9] = {h00, h01, h02, h01, h11, h12, h02, h12, h22};
  double Hinv[9];
  inv3(H, Hinv);
  double meanDiff = 0.0;
  for (int it = 0; it < max_its; it++) {
    const double cx = level_n_pos(sub0, nLevel), cy = level_n_pos(sub1, nLevel);
    const int xb = (int)(cx > 0.0 ? cx + 0.5 : cx - 0.5), yb = (int)(cy > 0.0 ? cy + 0.5 : cy - 0.5);
    const int b = HALF + 1;
    if (!(xb >= b && yb >= b && xb < wl - b && yb < hl - b)) return false;
    const double bx = cx - HALF, by = cy - HALF;
    const double dX = bx - floor(bx), dY = by - floor(by);
    const float fTL = (float)((1.0 - dX) * (1.0 - dY)), fTR = (float)((dX) * (1.0 - dY));
    const float fBL = (float)((1.0 - dX) * (dY)), fBR = (float)((dX) * (dY));
    // v3Accum in the reference's pixel order, y outer / x inner (:316-340): the per-pixel products go to LDS in that order
    // and lanes 0..2 walk one sum each (a butterfly over the lanes would round differently)
    for (int q = 0; q < NQL; q++) {
      const int k = q * 64 + lane;
      if (k < Q * Q) {
        const int x = k / Q + 1, y = k % Q + 1;
        const uint8_t* tl = img + (size_t)((int)by + y) * ip + (int)bx + x;
        const float fPixel = fTL * tl[0] + fTR * tl[1] + fBL * tl[ip] + fBR * tl[ip + 1];
        const double dDiff = (fPixel - (float)tmpl[y * PS + x]) + meanDiff;
        const int o = (y - 1) * Q + (x - 1);
        slab[o] = dDiff * gx[q]; slab[Q * Q + o] = dDiff * gy[q]; slab[2 * Q * Q + o] = dDiff;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    double acc = 0.0;
    if (lane < 3) {
      const double* sp = slab + lane * Q * Q;
      _Pragma("unroll") for (int o = 0; o < Q * Q; o++) acc += sp[o];
    }
    const double a0 = __shfl(acc, 0), a1 = __shfl(acc, 1), a2 = __shfl(acc, 2);
    __builtin_amdgcn_wave_barrier();
    const double u0 = Hinv[0] * a0 + Hinv[1] * a1 + Hinv[2] * a2;
    const double u1 = Hinv[3] * a0 + Hinv[4] * a1 + Hinv[5] * a2;
    const double u2 = Hinv[6] * a0 + Hinv[7] * a1 + Hinv[8] * a2;
    sub0 -= u0 * nLevelScale; sub1 -= u1 * nLevelScale; meanDiff -= u2;
    if (u0 * u0 + u1 * u1 < 0.03 * 0.03) return true;
  }
  return false;
}

// Eigen::JacobiSVD of the square 4x4 triangulation matrix (jni/MapMaker.cc:191-192): two-sided Jacobi on A itself as
// published for Eigen 3.0-3.1 -- sweeps over the pairs (p, q), q < p; a pair is rotated while max(|m_pq|, |m_qp|) >
// 2 eps max(|m_pp|, |m_qq|); the 2x2 step symmetrises the block with a left rotation, then diagonalises it with the Jacobi
// rotation of the symmetric block; V collects the right rotations; the column of V of the smallest |m_ii| is
// matrixV().col(3).  Statement for statement what oracle/mapgrow.cpp evaluates on the host (one lane, same order: same
// bits).  Third-party arithmetic, version unpinned: parity unpinned against the reference.
DEVFN void svd4_smallest_right_vector(const double Ain[16], double out[4]) {
  double M[16], V[16];
  for (int i = 0; i < 16; i++) { M[i] = Ain[i]; V[i] = (i % 5 == 0) ? 1.0 : 0.0; }
  const double precision = 2.0 * 2.220446049250313e-16;
  _Pragma("unroll 1") for (int sweep = 0; sweep < 64; sweep++) {
    bool finished = true;
    _Pragma("unroll") for (int p = 1; p < 4; p++)
      _Pragma("unroll") for (int q = 0; q < p; q++) {
        const double apq = fabs(M[p * 4 + q]), aqp = fabs(M[q * 4 + p]), off = apq > aqp ? apq : aqp;
        const double app = fabs(M[p * 4 + p]), aqq = fabs(M[q * 4 + q]), dia = app > aqq ? app : aqq;
        if (!(off > dia * precision)) continue;
        finished = false;
        const double m00 = M[p * 4 + p], m01 = M[p * 4 + q], m10 = M[q * 4 + p], m11 = M[q * 4 + q];
        double c1, s1;
        const double t = m00 + m11, d = m10 - m01;
        if (t == 0.0) { c1 = 0.0; s1 = d > 0.0 ? 1.0 : -1.0; }
        else { const double u = d / t; c1 = 1.0 / sqrt(1.0 + u * u); s1 = c1 * u; }
        const double x = c1 * m00 + s1 * m10, y = c1 * m01 + s1 * m11, z = -s1 * m01 + c1 * m11;
        double c2, s2;
        if (y == 0.0) { c2 = 1.0; s2 = 0.0; }
        else {
          const double tau = (x - z) / (2.0 * fabs(y)), w = sqrt(tau * tau + 1.0);
          const double tt = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
          const double sign_t = tt > 0.0 ? 1.0 : -1.0, n = 1.0 / sqrt(tt * tt + 1.0);
          s2 = -sign_t * (y / fabs(y)) * fabs(tt) * n; c2 = n;
        }
        const double cl = c1 * c2 + s1 * s2, sl = s1 * c2 - c1 * s2;
        _Pragma("unroll") for (int k = 0; k < 4; k++) { const double a = M[p * 4 + k], b = M[q * 4 + k]; M[p * 4 + k] = cl * a + sl * b; M[q * 4 + k] = -sl * a + cl * b; }
        _Pragma("unroll") for (int k = 0; k < 4; k++) { const double a = M[k * 4 + p], b = M[k * 4 + q]; M[k * 4 + p] = c2 * a - s2 * b; M[k * 4 + q] = s2 * a + c2 * b; }
        _Pragma("unroll") for (int k = 0; k < 4; k++) { const double a = V[k * 4 + p], b = V[k * 4 + q]; V[k * 4 + p] = c2 * a - s2 * b; V[k * 4 + q] = s2 * a + c2 * b; }
      }
    if (finished) break;
  }
  // the column of the smallest |m_ii| without a dynamically indexed register array (selects)
  double bestv = fabs(M[0]);
  out[0] = V[0]; out[1] = V[4]; out[2] = V[8]; out[3] = V[12];
  _Pragma("unroll") for (int i = 1; i < 4; i++) {
    const double a = fabs(M[i * 4 + i]);
    if (a < bestv) { bestv = a; out[0] = V[i]; out[1] = V[4 + i]; out[2] = V[8 + i]; out[3] = V[12 + i]; }
  }
}

// MapMaker::ReprojectPoint, jni/MapMaker.cc:174-200
DEVFN void reproject_point(const Pose& AfromB, const double v2A[2], const double v2B[2], double out[3]) {
  double PD[12];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) PD[r * 4 + c] = AfromB.R[r * 3 + c]; PD[r * 4 + 3] = AfromB.t[r]; }
  double A[16] = {-1.0, 0.0, v2B[0], 0.0, 0.0, -1.0, v2B[1], 0.0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = 0; c < 4; c++) { A[8 + c] = v2A[0] * PD[8 + c] - PD[0 + c]; A[12 + c] = v2A[1] * PD[8 + c] - PD[4 + c]; }
  double v[4];
  svd4_smallest_right_vector(A, v);
  if (v[3] == 0.0) v[3] = 0.00001;
  out[0] = v[0] / v[3]; out[1] = v[1] / v[3]; out[2] = v[2] / v[3];
}

DEVFN void unit_ray(const CamModel& cam, double ix, double iy, double out[3]) {   // myUnproject + normalize()
  double u[2];
  cam_unproject(cam, ix, iy, u);
  const double n = sqrt(u[0] * u[0] + u[1] * u[1] + 1.0);
  out[0] = u[0] / n; out[1] = u[1] / n; out[2] = 1.0 / n;
}
DEVFN void rot_inv(const Pose& T, const double p[3], double o[3]) {             // R^T p
  o[0] = T.R[0] * p[0] + T.R[3] * p[1] + T.R[6] * p[2];
  o[1] = T.R[1] * p[0] + T.R[4] * p[1] + T.R[7] * p[2];
  o[2] = T.R[2] * p[0] + T.R[5] * p[1] + T.R[8] * p[2];
}

