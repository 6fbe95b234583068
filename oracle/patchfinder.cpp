// ORACLE (test infrastructure only).  jni/vision/ImageHandler.cpp + jni/PatchFinder.cc restated.
#include "ptam_system.hpp"

namespace orc {

// sample(cv::Mat&, double, double, unsigned char&), jni/vision/ImageHandler.cpp:12-19
static inline uint8_t sample_u8(const uint8_t* im, int stride, double x, double y) {
  const int lx = (int)x, ly = (int)y;
  x -= lx; y -= ly;
  const uint8_t* p = im + (size_t)ly * stride + lx;
  return (uint8_t)((1 - y) * ((1 - x) * p[0] + x * p[1]) + y * ((1 - x) * p[stride] + x * p[stride + 1]));
}

// transform_image, jni/vision/ImageHandler.cpp:21-113 (CV_8UC1 branch, defaultValue 0).
// out is P x P tight.  Same accumulated stepping of p as the reference (:54, :59-61).
int transform_image(const uint8_t* in, int iw, int ih, int istride, uint8_t* out, int P, const double M[4],
                    const double inOrig[2], const double outOrig[2]) {
  const int w = P, h = P;
  const double across[2] = {M[0], M[2]}, down[2] = {M[1], M[3]};
  const double p0[2] = {inOrig[0] - (M[0] * outOrig[0] + M[1] * outOrig[1]), inOrig[1] - (M[2] * outOrig[0] + M[3] * outOrig[1])};
  double min_x = p0[0], min_y = p0[1], max_x = min_x, max_y = min_y;
  if (across[0] < 0) min_x += w * across[0]; else max_x += w * across[0];
  if (down[0] < 0) min_x += h * down[0]; else max_x += h * down[0];
  if (across[1] < 0) min_y += w * across[1]; else max_y += w * across[1];
  if (down[1] < 0) min_y += h * down[1]; else max_y += h * down[1];
  const double cr[2] = {down[0] - w * across[0], down[1] - w * across[1]};
  double p[2] = {p0[0], p0[1]};
  if (min_x >= 0 && min_y >= 0 && max_x < iw - 1 && max_y < ih - 1) {
    for (int i = 0; i < h; ++i, p[0] += cr[0], p[1] += cr[1])
      for (int j = 0; j < w; ++j, p[0] += across[0], p[1] += across[1])
        out[i * P + j] = sample_u8(in, istride, p[0], p[1]);
    return 0;
  }
  const float x_bound = (float)(iw - 1), y_bound = (float)(ih - 1);
  int count = 0;
  for (int i = 0; i < h; ++i, p[0] += cr[0], p[1] += cr[1])
    for (int j = 0; j < w; ++j, p[0] += across[0], p[1] += across[1]) {
      if (0 <= p[0] && 0 <= p[1] && p[0] < x_bound && p[1] < y_bound) out[i * P + j] = sample_u8(in, istride, p[0], p[1]);
      else { out[i * P + j] = 0; ++count; }
    }
  return count;
}

// PatchFinder::CalcSearchLevelAndWarpMatrix, jni/PatchFinder.cc:31-68
int finder_calc_level_and_warp(Finder& f, const MapPoint& p, const SE3& pose, const double d[4]) {
  const V3 cam = xform(pose, p.pos);
  const double ooz = 1.0 / cam[2];
  const V3 mr = rot(pose, p.pix_right), md = rot(pose, p.pix_down);
  const double r0 = mr[0] - cam[0] * mr[2] * ooz, r1 = mr[1] - cam[1] * mr[2] * ooz;
  const double d0 = md[0] - cam[0] * md[2] * ooz, d1 = md[1] - cam[1] * md[2] * ooz;
  const double aux1[2] = {(d[0] * r0 + d[1] * r1) * ooz, (d[2] * r0 + d[3] * r1) * ooz};
  const double aux2[2] = {(d[0] * d0 + d[1] * d1) * ooz, (d[2] * d0 + d[3] * d1) * ooz};
  f.warp_inv[0] = aux1[0]; f.warp_inv[1] = aux2[0]; f.warp_inv[2] = aux1[1]; f.warp_inv[3] = aux2[1];
  double det = f.warp_inv[0] * f.warp_inv[3] - f.warp_inv[1] * f.warp_inv[2];
  f.level = 0;
  while (det > 3 && f.level < ORC_LEVELS - 1) { f.level++; det *= 0.25; }
  if (det > 3 || det < 0.25) { f.bad = true; return -1; }
  return f.level;
}

// PatchFinder::MakeTemplateSums, jni/PatchFinder.cc:152-164
static void make_template_sums(Finder& f) {
  int s = 0, sq = 0;
  for (int i = 0; i < f.P * f.P; i++) { const int b = f.tmpl[i]; s += b; sq += b * b; }
  f.tsum = s; f.tsumsq = sq;
}

// PatchFinder::MakeTemplateCoarseCont, jni/PatchFinder.cc:79-125
void finder_make_template(Finder& f, const MapPoint& p, const KeyFrame& src) {
  double inv[4];
  inv2(f.warp_inv, inv);
  const double sc = (double)level_scale(f.level);
  const double m2[4] = {inv[0] * sc, inv[1] * sc, inv[2] * sc, inv[3] * sc};
  bool refresh = !f.have_last;
  for (int i = 0; !refresh && i < 2; i++) {
    const double dx = m2[0 + i] - f.last_warp[0 + i], dy = m2[2 + i] - f.last_warp[2 + i];
    const double lim = 0.07;
    if (dx * dx + dy * dy > lim * lim) refresh = true;
  }
  if (refresh) {
    const int l = p.src_level;
    const double inOrig[2] = {(double)p.irx, (double)p.iry}, outOrig[2] = {(double)(f.P / 2), (double)(f.P / 2)};
    if ((int)f.tmpl.size() != f.P * f.P) f.tmpl.assign(f.P * f.P, 0);
    const int nOutside = transform_image(src.im[l].data(), src.w[l], src.h[l], src.w[l], f.tmpl.data(), f.P, m2, inOrig, outOrig);
    f.bad = nOutside != 0;
    make_template_sums(f);
    f.have_last = true;
    for (int i = 0; i < 4; i++) f.last_warp[i] = m2[i];
  }
}

// PatchFinder::ZMSSDAtPoint, jni/PatchFinder.cc:352-380
int finder_zmssd(const Finder& f, const uint8_t* img, int w, int h, int stride, int icol, int irow) {
  const int b = f.P / 2;
  if (!(icol >= b && irow >= b && icol < w - b && irow < h - b)) return f.max_ssd + 1;
  const int bx = icol - b, by = irow - b;
  int sumsq = 0, sum = 0, cross = 0;
  for (int r = 0; r < f.P; r++) {
    const uint8_t* ip = img + (size_t)(by + r) * stride + bx;
    const uint8_t* tp = f.tmpl.data() + r * f.P;
    for (int c = 0; c < f.P; c++) { const int n = ip[c]; sum += n; sumsq += n * n; cross += n * tp[c]; }
  }
  const int SA = f.tsum, SB = sum, N = f.P * f.P;
  return ((2 * SA * SB - SA * SA - SB * SB) / N + sumsq + f.tsumsq - 2 * cross);
}

// PatchFinder::FindPatchCoarse, jni/PatchFinder.cc:170-235
bool finder_find_coarse(Finder& f, const double irPosIn[2], const KeyFrame& kf, unsigned nRange) {
  f.found = false;
  const int scale = level_scale(f.level);
  const double irPos[2] = {irPosIn[0] / scale, irPosIn[1] / scale};
  nRange = (nRange + scale - 1) / scale;
  int nTop = (int)(irPos[1] - nRange);
  const int nBottomPlusOne = (int)(irPos[1] + nRange + 1);
  const int nLeft = (int)(irPos[0] - nRange);
  const int nRight = (int)(irPos[0] + nRange);
  const int l = f.level, rows = kf.h[l];
  if (nTop < 0) nTop = 0;
  if (nTop >= rows) return false;
  if (nBottomPlusOne <= 0) return false;
  int i = kf.lut[l][nTop];
  const int i_end = nBottomPlusOne >= rows ? (int)kf.corners[l].size() : kf.lut[l][nBottomPlusOne];
  int best_x = -1, best_y = -1;
  int nBestSSD = f.max_ssd + 1;
  for (; i < i_end; i++) {
    const int cx = kf.corners[l][i] & 0xFFFF, cy = kf.corners[l][i] >> 16;
    if (cx < nLeft || cx > nRight) continue;
    const double dx = irPos[0] - cx, dy = irPos[1] - cy;
    if (dx * dx + dy * dy > (double)(nRange * nRange)) continue;
    const int nSSD = finder_zmssd(f, kf.im[l].data(), kf.w[l], kf.h[l], kf.w[l], cx, cy);
    f.n_zmssd++;
    if (nSSD < nBestSSD) { best_x = cx; best_y = cy; nBestSSD = nSSD; }
  }
  if (nBestSSD < f.max_ssd) {
    f.coarse[0] = level_zero_pos(best_x, l); f.coarse[1] = level_zero_pos(best_y, l);
    f.found = true;
  } else f.found = false;
  return f.found;
}

// PatchFinder::MakeSubPixTemplate, jni/PatchFinder.cc:242-271
void finder_make_subpix(Finder& f) {
  const int P = f.P, Q = P - 2;
  f.jac[0].assign(Q * Q, 0.0); f.jac[1].assign(Q * Q, 0.0);
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int x = 1; x < P - 1; x++)
    for (int y = 1; y < P - 1; y++) {
      const double gx = 0.5 * (f.tmpl[y * P + x + 1] - f.tmpl[y * P + x - 1]);
      const double gy = 0.5 * (f.tmpl[(y + 1) * P + x] - f.tmpl[(y - 1) * P + x]);
      f.jac[0][(x - 1) * Q + (y - 1)] = gx; f.jac[1][(x - 1) * Q + (y - 1)] = gy;
      const double g[3] = {gx, gy, 1.0};
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H[r * 3 + c] += g[r] * g[c];
    }
  inv3(H, f.hinv);
  f.subpix[0] = f.coarse[0]; f.subpix[1] = f.coarse[1];
  f.mean_diff = 0.0;
}

// PatchFinder::IterateSubPix, jni/PatchFinder.cc:291-350
static double finder_iterate_subpix(Finder& f, const KeyFrame& kf) {
  const int P = f.P, Q = P - 2, l = f.level;
  const double cx = level_n_pos(f.subpix[0], l), cy = level_n_pos(f.subpix[1], l);
  const int x_border = (int)(cx > 0.0 ? cx + 0.5 : cx - 0.5), y_border = (int)(cy > 0.0 ? cy + 0.5 : cy - 0.5);
  const int b = P / 2 + 1;
  if (!(x_border >= b && y_border >= b && x_border < kf.w[l] - b && y_border < kf.h[l] - b)) return -1.0;
  const double bx = cx - P / 2, by = cy - P / 2;
  double acc[3] = {0, 0, 0};
  const double dX = bx - floor(bx), dY = by - floor(by);
  const float fMixTL = (float)((1.0 - dX) * (1.0 - dY)), fMixTR = (float)((dX) * (1.0 - dY));
  const float fMixBL = (float)((1.0 - dX) * (dY)), fMixBR = (float)((dX) * (dY));
  const int stride = kf.w[l];
  const uint8_t* im = kf.im[l].data();
  for (int y = 1; y < P - 1; y++) {
    const uint8_t* tl = im + (size_t)((int)by + y) * stride + (int)bx + 1;
    for (int x = 1; x < P - 1; x++) {
      const float fPixel = fMixTL * tl[0] + fMixTR * tl[1] + fMixBL * tl[stride] + fMixBR * tl[stride + 1];
      tl++;
      const double dDiff = fPixel - f.tmpl[y * P + x] + f.mean_diff;   // float - int -> float, then + double
      acc[0] += dDiff * f.jac[0][(x - 1) * Q + (y - 1)];
      acc[1] += dDiff * f.jac[1][(x - 1) * Q + (y - 1)];
      acc[2] += dDiff;
    }
  }
  double up[3];
  for (int r = 0; r < 3; r++) up[r] = f.hinv[r * 3 + 0] * acc[0] + f.hinv[r * 3 + 1] * acc[1] + f.hinv[r * 3 + 2] * acc[2];
  const int sc = level_scale(l);
  f.subpix[0] -= up[0] * sc; f.subpix[1] -= up[1] * sc;
  f.mean_diff -= up[2];
  return up[0] * up[0] + up[1] * up[1];
}

// PatchFinder::IterateSubPixToConvergence, jni/PatchFinder.cc:273-289
bool finder_iterate_subpix_to_convergence(Finder& f, const KeyFrame& kf, int nMaxIts) {
  const double dConvLimit = 0.03;
  for (int nIts = 0; nIts < nMaxIts; nIts++) {
    const double d = finder_iterate_subpix(f, kf);
    if (d < 0) return false;
    if (d < dConvLimit * dConvLimit) return true;
  }
  return false;
}

}  // namespace orc
