// ORACLE (test infrastructure only).  jni/MiniPatch.cc restated: 9x9 raw-SSD patch search at FAST corners
// (used by the reference's trail tracking, jni/Tracker.cc:294-346).
#include "ptam_oracle.h"
#include <cstring>

static const int kHalf = 4;   // MiniPatch::mnHalfPatchSize, jni/MiniPatch.cc:86

// MiniPatch::SampleFromImage, jni/MiniPatch.cc:71-83 (the ROI copy of ImageHandler.cpp:115-118)
extern "C" int orc_minipatch_sample(const uint8_t* img, int w, int h, int stride, int x, int y, uint8_t* patch /* 81 */) {
  if (!(x >= kHalf && y >= kHalf && x < w - kHalf && y < h - kHalf)) return 0;   // the reference asserts
  for (int r = 0; r < 2 * kHalf + 1; r++) memcpy(patch + r * (2 * kHalf + 1), img + (size_t)(y - kHalf + r) * stride + (x - kHalf), 2 * kHalf + 1);
  return 1;
}

// MiniPatch::SSDAtPoint, jni/MiniPatch.cc:6-30
static int ssd_at(const uint8_t* patch, const uint8_t* img, int w, int h, int stride, int icol, int irow, int max_ssd) {
  if (!(icol >= kHalf && irow >= kHalf && icol < w - kHalf && irow < h - kHalf)) return max_ssd + 1;
  int s = 0;
  for (int r = 0; r < 9; r++) {
    const uint8_t* ip = img + (size_t)(irow - kHalf + r) * stride + (icol - kHalf);
    const uint8_t* tp = patch + r * 9;
    for (int c = 0; c < 9; c++) { const int d = ip[c] - tp[c]; s += d * d; }
  }
  return s;
}

// MiniPatch::FindPatch, jni/MiniPatch.cc:35-68 (pvRowLUT == NULL as the tracker calls it; the LUT only moves the start
// iterator).  pos in/out: integer pixel position.  Returns 1 if found.
extern "C" int orc_minipatch_find(const uint8_t* patch, const uint8_t* img, int w, int h, int stride, const uint32_t* corners,
                                  int n, int range, int max_ssd, int pos[2]) {
  int bx = 0, by = 0, best = max_ssd + 1;
  const int L = pos[0] - range, R = pos[0] + range, T = pos[1] - range, B = pos[1] + range;
  int i = 0;
  for (; i < n; i++) if ((int)(corners[i] >> 16) >= T) break;
  for (; i < n; i++) {
    const int cx = corners[i] & 0xFFFF, cy = corners[i] >> 16;
    if (cx < L || cx > R) continue;
    if (cy > B) break;
    const int s = ssd_at(patch, img, w, h, stride, cx, cy, max_ssd);
    if (s < best) { bx = cx; by = cy; best = s; }
  }
  if (best < max_ssd) { pos[0] = bx; pos[1] = by; return 1; }
  return 0;
}
