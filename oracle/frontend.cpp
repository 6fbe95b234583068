// ORACLE (test infrastructure only -- see ptam_oracle.h).  Frame front-end restatement:
// pyramid, FAST-10, row LUT, FAST score, non-max suppression, Shi-Tomasi.
#include "ptam_oracle.h"
#include <cmath>
#include <cstring>
#include <vector>

// Ring offsets in the reference's order, jni/vision/cvfast.cpp:6094-6111 (dx, dy).
static const int kRing[16][2] = {
    {0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

extern "C" void orc_halfsample(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  // jni/KeyFrame.cc:20-23 (cv::resize to size/2); spec: exact 2x2 box mean with rounding.
  const int dw = w / 2, dh = h / 2;
  for (int y = 0; y < dh; y++) {
    const uint8_t* r0 = src + (size_t)(2 * y) * sstride;
    const uint8_t* r1 = r0 + sstride;
    uint8_t* d = dst + (size_t)y * dstride;
    for (int x = 0; x < dw; x++)
      d[x] = (uint8_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2) >> 2);
  }
}

// >= 10 contiguous set bits on a 16-bit ring
static inline bool ring_has_run10(unsigned m16) {
  unsigned m = m16 | (m16 << 16);
  unsigned r2 = m & (m >> 1);
  unsigned r4 = r2 & (r2 >> 2);
  unsigned r8 = r4 & (r4 >> 4);
  unsigned r10 = r8 & (r2 >> 8);
  return (r10 & 0xFFFFu) != 0;
}

extern "C" int orc_fast10(const uint8_t* img, int w, int h, int stride, int threshold,
                          uint32_t* corners, int cap) {
  // jni/vision/cvfast.cpp:6088-9241: the learned decision tree is equivalent to
  // ">= 10 contiguous ring pixels all > c+t or all < c-t" (pinned exhaustively by
  // oracle/pin_fast_tree.py).  Loop bounds :6113-6119, emit order :9237-9238.
  int off[16];
  for (int k = 0; k < 16; k++) off[k] = kRing[k][0] + kRing[k][1] * stride;
  int n = 0;
  for (int y = 3; y < h - 3; y++) {
    const uint8_t* row = img + (size_t)y * stride;
    for (int x = 3; x < w - 3; x++) {
      const uint8_t* p = row + x;
      const int cb = *p + threshold, c_b = *p - threshold;
      // early exit: an arc of 10 contains at least one pixel of every opposite pair
      const int p0 = p[off[0]], p8 = p[off[8]];
      if (!(p0 > cb || p0 < c_b || p8 > cb || p8 < c_b)) continue;
      const int p4 = p[off[4]], p12 = p[off[12]];
      if (!(p4 > cb || p4 < c_b || p12 > cb || p12 < c_b)) continue;
      unsigned mb = 0, md = 0;
      for (int k = 0; k < 16; k++) {
        const int v = p[off[k]];
        mb |= (unsigned)(v > cb) << k;
        md |= (unsigned)(v < c_b) << k;
      }
      if (ring_has_run10(mb) || ring_has_run10(md)) {
        if (n < cap) corners[n] = (uint32_t)x | ((uint32_t)y << 16);
        n++;
      }
    }
  }
  return n;
}

extern "C" void orc_row_lut(const uint32_t* corners, int n, int h, int* lut) {
  // jni/KeyFrame.cc:43-49
  int v = 0;
  for (int y = 0; y < h; y++) {
    while (v < n && y > (int)(corners[v] >> 16)) v++;
    lut[y] = v;
  }
}

extern "C" void orc_fast_score(const uint8_t* img, int w, int h, int stride, const uint32_t* corners,
                               int n, int barrier, int* scores) {
  // jni/vision/cvfast.cpp:9337-9369 (old_style_corner_score) via :9371-9393
  (void)w; (void)h;
  for (int i = 0; i < n; i++) {
    const int x = corners[i] & 0xFFFF, y = corners[i] >> 16;
    const uint8_t* p = img + (size_t)y * stride + x;
    const int cb = *p + barrier, c_b = *p - barrier;
    int sp = 0, sn = 0;
    for (int k = 0; k < 16; k++) {
      const int v = p[kRing[k][0] + kRing[k][1] * stride];
      if (v > cb) sp += v - cb;
      else if (v < c_b) sn += c_b - v;
    }
    scores[i] = sp > sn ? sp : sn;
  }
}

extern "C" int orc_nonmax(const uint32_t* corners, const int* scores, int n, int quirk, uint32_t* out) {
  // jni/vision/cvfast.cpp:9243-9335, statement by statement (including the point_above /
  // point_below cursors), on packed integer coordinates.
  int nout = 0;
  if (n < 1) return 0;
  auto X = [&](int i) { return (int)(corners[i] & 0xFFFF); };
  auto Y = [&](int i) { return (int)(corners[i] >> 16); };
  const int last_row = Y(n - 1);
  std::vector<int> row_start(last_row + 1, -1);
  int prev_row = -1;
  for (int i = 0; i < n; i++)
    if (Y(i) != prev_row) { row_start[Y(i)] = i; prev_row = Y(i); }
  int point_above = 0, point_below = 0;
  const int sz = n;
  for (int i = 0; i < sz; i++) {
    const int score = scores[i];
    const int px = X(i), py = Y(i);
    // check left (:9276-9279)
    if (i > 0)
      if (X(i - 1) == px - 1 && Y(i - 1) == py && scores[i - 1] > score) continue;
    // check right (:9281-9285).  Reference tests corners[i-1](1) == pos(1): quirk #7.
    if (i < sz - 1) {
      if (quirk) {
        // i == 0 reads corners[-1] in the reference (undefined); restated as "no match".
        if (i > 0 && X(i + 1) == px + 1 && Y(i - 1) == py && scores[i + 1] > score) continue;
      } else {
        if (X(i + 1) == px + 1 && Y(i + 1) == py && scores[i + 1] > score) continue;
      }
    }
    bool suppressed = false;
    // check above (:9287-9306)
    if (py != 0 && row_start[py - 1] != -1) {
      if (Y(point_above) < py - 1) point_above = row_start[py - 1];
      for (; Y(point_above) < py && X(point_above) < px - 1; point_above++) {}
      for (int j = point_above; Y(j) < py && X(j) <= px + 1; j++) {
        const int x = X(j);
        if ((x == px - 1 || x == px || x == px + 1) && scores[j] > score) { suppressed = true; break; }
      }
    }
    // check below (:9308-9326)
    if (!suppressed && py != last_row && row_start[py + 1] != -1 && point_below < sz) {
      if (Y(point_below) < py + 1) point_below = row_start[py + 1];
      for (; point_below < sz && Y(point_below) == py + 1 && X(point_below) < px - 1; point_below++) {}
      for (int j = point_below; j < sz && Y(j) == py + 1 && X(j) <= px + 1; j++) {
        const int x = X(j);
        if ((x == px - 1 || x == px || x == px + 1) && scores[j] > score) { suppressed = true; break; }
      }
    }
    if (!suppressed) out[nout++] = corners[i];
  }
  return nout;
}

extern "C" double orc_shi_tomasi(const uint8_t* img, int stride, int nsize, int px, int py) {
  // jni/vision/ImageHandler.cpp:124-155
  double dXX = 0, dYY = 0, dXY = 0;
  const int startx = px - nsize, starty = py - nsize, endx = px + nsize, endy = py + nsize;
  for (int cy = starty; cy <= endy; cy++)
    for (int cx = startx; cx <= endx; cx++) {
      const double dx = (double)(img[(size_t)cy * stride + cx + 1] - img[(size_t)cy * stride + cx - 1]);
      const double dy = (double)(img[(size_t)(cy + 1) * stride + cx] - img[(size_t)(cy - 1) * stride + cx]);
      dXX += dx * dx; dYY += dy * dy; dXY += dx * dy;
    }
  const int nPixels = (endx - startx + 1) * (endy - starty + 1);
  dXX = dXX / (2.0 * nPixels); dYY = dYY / (2.0 * nPixels); dXY = dXY / (2.0 * nPixels);
  return 0.5 * (dXX + dYY - sqrt((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY)));
}

extern "C" int orc_candidates(const uint8_t* img, int w, int h, int stride, const uint32_t* maxcorners, int n,
                              double min_score, int border, uint32_t* out_pos, double* out_score, int cap) {
  // KeyFrame::MakeKeyFrame_Rest candidate loop, jni/KeyFrame.cc:66-95 (colour sampling and the SBI are not part of the path)
  int m = 0;
  for (int i = 0; i < n; i++) {
    const int x = maxcorners[i] & 0xFFFF, y = maxcorners[i] >> 16;
    if (!(x >= border && y >= border && x < w - border && y < h - border)) continue;   // :72-73
    const double st = orc_shi_tomasi(img, stride, 3, x, y);                               // :79
    if (st > min_score) {                                                                 // :81
      if (m < cap) { out_pos[m] = maxcorners[i]; out_score[m] = st; }
      m++;
    }
  }
  return m;
}

extern "C" int orc_thin_candidates(const uint32_t* pos, const double* score, int n, int level, const double* meas_root,
                                   const int* meas_level, int n_meas, uint32_t* out_pos, double* out_score) {
  // MapMaker::ThinCandidates, jni/MapMaker.cc:393-422; rounded() :381-386; LevelScale jni/LevelHelpers.h
  std::vector<double> busy;
  const int scale = 1 << level;
  for (int j = 0; j < n_meas; j++) {
    if (!(meas_level[j] == level || meas_level[j] == level + 1)) continue;
    for (int k = 0; k < 2; k++) {
      const double v = meas_root[2 * j + k] / scale;
      busy.push_back((double)static_cast<int>(v > 0.0 ? v + 0.5 : v - 0.5));
    }
  }
  const unsigned int nMinMagSquared = 10 * 10;
  int m = 0;
  for (int i = 0; i < n; i++) {
    const double cx = pos[i] & 0xFFFF, cy = pos[i] >> 16;
    bool good = true;
    for (size_t j = 0; j < busy.size(); j += 2) {
      const double dx = busy[j] - cx, dy = busy[j + 1] - cy;
      if (dx * dx + dy * dy < nMinMagSquared) { good = false; break; }
    }
    if (good) { out_pos[m] = pos[i]; out_score[m] = score[i]; m++; }
  }
  return m;
}

extern "C" int orc_make_keyframe_lite(const uint8_t* gray, int w, int h, int stride,
                                      const int thr[ORC_LEVELS], uint8_t* const lvl_img[ORC_LEVELS],
                                      uint32_t* const corners[ORC_LEVELS], int cap,
                                      int ncorners[ORC_LEVELS], int* const lut[ORC_LEVELS]) {
  // jni/KeyFrame.cc:5-51
  int lw = w, lh = h;
  for (int y = 0; y < h; y++) memcpy(lvl_img[0] + (size_t)y * w, gray + (size_t)y * stride, w);  // :12
  for (int l = 0; l < ORC_LEVELS; l++) {
    if (l != 0) {  // :19-23
      orc_halfsample(lvl_img[l - 1], lw, lh, lw, lvl_img[l], lw / 2);
      lw /= 2; lh /= 2;
    }
    int n = orc_fast10(lvl_img[l], lw, lh, lw, thr[l], corners[l], cap);  // :32-39
    if (n > cap) n = cap;
    ncorners[l] = n;
    orc_row_lut(corners[l], n, lh, lut[l]);  // :43-49
  }
  return 0;
}
