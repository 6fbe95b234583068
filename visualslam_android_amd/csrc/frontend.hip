// Frame front-end on gfx950: KeyFrame::MakeKeyFrame_Lite (jni/KeyFrame.cc:5-51) and fast_nonmax
// (jni/vision/cvfast.cpp:9243-9400) for all streams of a system at once.
//
// Kernels (byte/integer work, one pass over each level image; measured LDS-issue bound):
//   k_pyr_fast0   one workgroup per 16-row band of level 0: the band (+3-row halo) is staged in LDS
//                 with 16-B coalesced loads, levels 1..3 of the band are produced from LDS
//                 ((a+b+c+d+2)>>2) and written, and FAST-10 runs on the staged rows in two phases: a 5-read
//                 quick reject on every pixel, then the full segment test on the compacted survivors, which
//                 set bits of the band's 64-bit corner-mask words in LDS.
//   k_fast_lvl    the same FAST band sweep for levels 1..3 (their rows come back from L2/MALL).
//   k_compact     per (stream, level): block scan over the popcounts of the mask words (raster order); every
//                 thread expands its words at its scanned offset -> bit-exact raster-ordered corner list
//                 and row LUT (jni/KeyFrame.cc:43-49) without ordered atomics.
//   k_score / k_nonmax   compute_fast_score_old + nonmax_suppression, one lane per corner.
//   k_candidates / k_thin_candidates   MakeKeyFrame_Rest's Shi-Tomasi candidates (jni/KeyFrame.cc:66-95) and
//                 MapMaker::ThinCandidates (jni/MapMaker.cc:393-422), ordered block compaction.
#include "vslam_internal.h"
#include <stdlib.h>

#define BAND 16          // level-0 rows per workgroup (multiple of 8: three halvings stay inside a band)
#define HALO 3           // FAST ring radius (cvfast.cpp:6094-6111)
#define FE_THREADS 256

struct FeArgs {
  const uint8_t* in; size_t in_sstride; int in_pitch;
  uint8_t* lvl[NLEV]; size_t lvl_sstride[NLEV]; int lvl_pitch[NLEV];
  int w[NLEV], h[NLEV], nchunk[NLEV], thr[NLEV], cap[NLEV];
  unsigned long long* cmask[NLEV];
  int* rowcnt[NLEV];
  int* rowlut[NLEV];
  uint32_t* corners[NLEV];
  int* ncorners;
  int* overflow;
  int band_first[NLEV];   // k_fast_lvl: first blockIdx.x of each level
};

__device__ __forceinline__ bool ring_run10(unsigned m16) {
  unsigned m = m16 | (m16 << 16);
  unsigned r2 = m & (m >> 1);
  unsigned r4 = r2 & (r2 >> 2);
  unsigned r8 = r4 & (r4 >> 4);
  unsigned r10 = r8 & (r2 >> 8);
  return (r10 & 0xFFFFu) != 0;
}

// FAST-10 segment test (cvfast.cpp:6088-9241; equivalence with the decision tree is pinned in
// tests/golden/fast10_tree_pin.json), split in two so the expensive part runs on densely packed lanes:
//   quick reject: an arc of 10 contiguous ring pixels contains at least one pixel of every opposite pair, so a corner
//                 needs (p0 or p8) AND (p4 or p12) outside [c-t, c+t]; rejects most pixels (fast_band phase A, and
//                 fast10_quick below for one pixel);
//   fast10_full : the 16-pixel brighter/darker masks and the run-of-10 test.
__device__ __forceinline__ bool fast10_quick(const uint8_t* p, int lp, int t) {
  const int c = p[0], cb = c + t, c_b = c - t;
  const int p0 = p[3 * lp], p8 = p[-3 * lp];
  if (!(p0 > cb || p0 < c_b || p8 > cb || p8 < c_b)) return false;
  const int p4 = p[3], p12 = p[-3];
  return p4 > cb || p4 < c_b || p12 > cb || p12 < c_b;
}
__device__ __forceinline__ bool fast10_full(const uint8_t* p, int lp, int t) {
  const int c = p[0], cb = c + t, c_b = c - t;
  unsigned mb = 0, md = 0;
#define RINGPIX(k, dx, dy) { const int v = p[(dx) + (dy) * lp]; mb |= (unsigned)(v > cb) << (k); md |= (unsigned)(v < c_b) << (k); }
  RINGPIX(0, 0, 3) RINGPIX(1, 1, 3) RINGPIX(2, 2, 2) RINGPIX(3, 3, 1) RINGPIX(4, 3, 0) RINGPIX(5, 3, -1)
  RINGPIX(6, 2, -2) RINGPIX(7, 1, -3) RINGPIX(8, 0, -3) RINGPIX(9, -1, -3) RINGPIX(10, -2, -2)
  RINGPIX(11, -3, -1) RINGPIX(12, -3, 0) RINGPIX(13, -3, 1) RINGPIX(14, -2, 2) RINGPIX(15, -1, 3)
#undef RINGPIX
  return ring_run10(mb) || ring_run10(md);
}

// Stage rows [gy0, gy0+nrows) of an image into LDS (row pitch lp, 16-B aligned). Rows outside the image are skipped.
__device__ __forceinline__ void stage_rows(uint8_t* tile, int lp, const uint8_t* img, int pitch, int w, int h,
                                           int gy0, int nrows) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
  for (int r = wave; r < nrows; r += nwave) {
    const int gy = gy0 + r;
    if (gy < 0 || gy >= h) continue;
    const uint8_t* src = img + (size_t)gy * pitch;
    uint8_t* dst = tile + r * lp;
    int done = 0;
    if ((((uintptr_t)src) & 15) == 0) {
      const int nvec = w >> 4;
      for (int v = lane; v < nvec; v += 64) ((uint4*)dst)[v] = ((const uint4*)src)[v];
      done = nvec << 4;
    }
    for (int i = done + lane; i < w; i += 64) dst[i] = src[i];
  }
}

// FAST over nrows image rows starting at y0; tile row 0 holds image row y0 - HALO.
// Per group of FB_ROWS rows: (A) every pixel runs the quick reject, survivors are appended (wave-aggregated) to an LDS
// candidate list; (B) the full test runs over the dense list and sets bits of the band's corner mask in LDS.
// Finally the mask words and the row counts go to global memory with coalesced stores.
#define FB_ROWS 4
__device__ __forceinline__ void fast_band(const uint8_t* tile, int lp, int y0, int nrows, int w, int h, int thr,
                                          int nchunk, unsigned long long* cmask /* [h][nchunk] */,
                                          int* rowcnt /* [h] */, unsigned long long* mask_lds, unsigned short* cand, int* ncand) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
  for (int i = threadIdx.x; i < nrows * nchunk; i += blockDim.x) mask_lds[i] = 0ull;
  for (int r0 = 0; r0 < nrows; r0 += FB_ROWS) {
    if (threadIdx.x == 0) *ncand = 0;
    __syncthreads();
    const int gr = min(FB_ROWS, nrows - r0);
    // (A) quick reject, four pixels per lane: one aligned dword of centres, the dwords 3 rows above / below, and the
    //     pixels 3 to the left / right assembled from the neighbouring aligned dwords (v_alignbyte).  The compiler reads
    //     the bytes straight out of the registers (SDWA), so the LDS sees 5 dword reads per 4 pixels instead of 20 byte reads.
    const int nq = (w + 3) >> 2;                                     // dwords per row
#ifndef VSLAM_FE_SKIP_QUICK
    for (int it = wave * 64 + lane; it - lane < gr * nq; it += nwave * 64) {
      const int r = r0 + it / nq, q = it - (it / nq) * nq;
      const int y = y0 + r, x0 = q << 2;
      unsigned pass = 0;
      if (it < gr * nq && y >= HALO && y < h - HALO) {               // cvfast.cpp:6113-6117
        const uint8_t* row = tile + (r + HALO) * lp + x0;
        const unsigned cw = *(const unsigned*)row, lw = *(const unsigned*)(row - 4), rw = *(const unsigned*)(row + 4);
        const unsigned uw = *(const unsigned*)(row + 3 * lp), dw = *(const unsigned*)(row - 3 * lp);
        const unsigned p12w = __builtin_amdgcn_alignbyte(cw, lw, 1);   // pixels x-3 .. x
        const unsigned p4w = __builtin_amdgcn_alignbyte(rw, cw, 3);    // pixels x+3 .. x+6
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int x = x0 + j;
          const int c = (cw >> (8 * j)) & 255u, cb = c + thr, c_b = c - thr;
          const int p0 = (uw >> (8 * j)) & 255u, p8 = (dw >> (8 * j)) & 255u;
          const int p4 = (p4w >> (8 * j)) & 255u, p12 = (p12w >> (8 * j)) & 255u;
          const bool ok = (p0 > cb || p0 < c_b || p8 > cb || p8 < c_b) && (p4 > cb || p4 < c_b || p12 > cb || p12 < c_b);
          if (ok && x >= HALO && x < w - HALO) pass |= 1u << j;
        }
      }
      unsigned long long bm[4];
      int tot = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) { bm[j] = __ballot((pass >> j) & 1u); tot += __popcll(bm[j]); }
      if (tot) {
        int base = 0;
        if (lane == 0) base = atomicAdd(ncand, tot);
        base = __shfl(base, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if ((pass >> j) & 1u) cand[base + __popcll(bm[j] & ((1ull << lane) - 1ull))] = (unsigned short)((r << 12) | (x0 + j));
          base += __popcll(bm[j]);
        }
      }
    }
#endif
    __syncthreads();
    const int n = *ncand;
#ifndef VSLAM_FE_SKIP_FULL
    for (int i = threadIdx.x; i < n; i += blockDim.x) {               // (B) full segment test on packed lanes
      const int r = cand[i] >> 12, x = cand[i] & 4095;
      if (fast10_full(tile + (r + HALO) * lp + x, lp, thr)) atomicOr(&mask_lds[r * nchunk + (x >> 6)], 1ull << (x & 63));
    }
#endif
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nrows * nchunk; i += blockDim.x) cmask[(size_t)y0 * nchunk + i] = mask_lds[i];
  if ((int)threadIdx.x < nrows) {
    int c = 0;
    for (int k = 0; k < nchunk; k++) c += __popcll(mask_lds[threadIdx.x * nchunk + k]);
    rowcnt[y0 + threadIdx.x] = c;
  }
}

// 2x2 box mean of nrows_out output rows: src rows (2j, 2j+1) of an LDS tile -> LDS tile + global.
__device__ __forceinline__ void halve_rows(const uint8_t* src, int slp, uint8_t* dst, int dlp, int wout, int nrows_out,
                                           uint8_t* gdst, int gpitch) {
  const int q = wout >> 2;
  for (int it = threadIdx.x; it < nrows_out * q; it += blockDim.x) {
    const int j = it / q, xq = it - j * q;
    const uint2 a = *(const uint2*)(src + (2 * j) * slp + 8 * xq);
    const uint2 b = *(const uint2*)(src + (2 * j + 1) * slp + 8 * xq);
    uint32_t o = 0;
#define HQ(k, wa, wb, sh) { const unsigned s = ((wa >> sh) & 255u) + ((wa >> (sh + 8)) & 255u) + ((wb >> sh) & 255u) + ((wb >> (sh + 8)) & 255u) + 2u; o |= (s >> 2) << (8 * k); }
    HQ(0, a.x, b.x, 0) HQ(1, a.x, b.x, 16) HQ(2, a.y, b.y, 0) HQ(3, a.y, b.y, 16)
#undef HQ
    if (dst) *(uint32_t*)(dst + j * dlp + 4 * xq) = o;
    *(uint32_t*)(gdst + (size_t)j * gpitch + 4 * xq) = o;
  }
  const int rem = wout - (q << 2);
  for (int it = threadIdx.x; it < nrows_out * rem; it += blockDim.x) {
    const int j = it / rem, x = (q << 2) + (it - j * rem);
    const uint8_t* r0 = src + (2 * j) * slp + 2 * x;
    const uint8_t* r1 = r0 + slp;
    const uint8_t o = (uint8_t)((r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2);
    if (dst) dst[j * dlp + x] = o;
    gdst[(size_t)j * gpitch + x] = o;
  }
}

__global__ __launch_bounds__(FE_THREADS) void k_pyr_fast0(FeArgs a, int lp0, int lp1, int lp2) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t* t0 = lds;                               // (BAND + 2*HALO) rows of level 0
  uint8_t* t1 = t0 + (BAND + 2 * HALO) * lp0;      // BAND/2 rows of level 1
  uint8_t* t2 = t1 + (BAND / 2) * lp1;             // BAND/4 rows of level 2
  unsigned long long* mask_lds = (unsigned long long*)(((uintptr_t)(t2 + (BAND / 4) * lp2) + 15) & ~(uintptr_t)15);
  unsigned short* cand = (unsigned short*)(mask_lds + BAND * a.nchunk[0]);
  int* ncand = (int*)(cand + FB_ROWS * lp0);
  const int b = blockIdx.x, s = blockIdx.y;
  const uint8_t* in = a.in + (size_t)s * a.in_sstride;
  const int y0 = b * BAND;
  stage_rows(t0, lp0, in, a.in_pitch, a.w[0], a.h[0], y0 - HALO, BAND + 2 * HALO);
  __syncthreads();
  // pyramid rows owned by this band (jni/KeyFrame.cc:19-23; 2:1 area filter)
  int n1 = min(BAND / 2, a.h[1] - y0 / 2); n1 = max(n1, 0);
  halve_rows(t0 + HALO * lp0, lp0, t1, lp1, a.w[1], n1, a.lvl[1] + (size_t)s * a.lvl_sstride[1] + (size_t)(y0 / 2) * a.lvl_pitch[1], a.lvl_pitch[1]);
  __syncthreads();
  int n2 = min(BAND / 4, a.h[2] - y0 / 4); n2 = max(n2, 0);
  halve_rows(t1, lp1, t2, lp2, a.w[2], n2, a.lvl[2] + (size_t)s * a.lvl_sstride[2] + (size_t)(y0 / 4) * a.lvl_pitch[2], a.lvl_pitch[2]);
  __syncthreads();
  int n3 = min(BAND / 8, a.h[3] - y0 / 8); n3 = max(n3, 0);
  halve_rows(t2, lp2, nullptr, 0, a.w[3], n3, a.lvl[3] + (size_t)s * a.lvl_sstride[3] + (size_t)(y0 / 8) * a.lvl_pitch[3], a.lvl_pitch[3]);
  // FAST-10 on the level-0 rows of the band
  const int nrows = min(BAND, a.h[0] - y0);
  fast_band(t0, lp0, y0, nrows, a.w[0], a.h[0], a.thr[0], a.nchunk[0],
            a.cmask[0] + (size_t)s * a.h[0] * a.nchunk[0], a.rowcnt[0] + (size_t)s * a.h[0], mask_lds, cand, ncand);
}

__global__ __launch_bounds__(FE_THREADS) void k_fast_lvl(FeArgs a) {
  extern __shared__ __align__(16) uint8_t lds[];
  int l = 1;
  if ((int)blockIdx.x >= a.band_first[2]) l = 2;
  if ((int)blockIdx.x >= a.band_first[3]) l = 3;
  const int b = blockIdx.x - a.band_first[l], s = blockIdx.y;
  const int lp = (a.w[l] + 15) & ~15;
  uint8_t* t = lds;
  unsigned long long* mask_lds = (unsigned long long*)(((uintptr_t)(t + (BAND + 2 * HALO) * lp) + 15) & ~(uintptr_t)15);
  unsigned short* cand = (unsigned short*)(mask_lds + BAND * a.nchunk[l]);
  int* ncand = (int*)(cand + FB_ROWS * lp);
  const int y0 = b * BAND;
  stage_rows(t, lp, a.lvl[l] + (size_t)s * a.lvl_sstride[l], a.lvl_pitch[l], a.w[l], a.h[l], y0 - HALO, BAND + 2 * HALO);
  __syncthreads();
  const int nrows = min(BAND, a.h[l] - y0);
  fast_band(t, lp, y0, nrows, a.w[l], a.h[l], a.thr[l], a.nchunk[l],
            a.cmask[l] + (size_t)s * a.h[l] * a.nchunk[l], a.rowcnt[l] + (size_t)s * a.h[l], mask_lds, cand, ncand);
}

// Raster-ordered corner lists + row LUT from the corner bit-masks, one workgroup per (band of 16 rows, level, stream): the
// workgroup adds the row counts above its band (its list offset = lut of its first row), scans the popcounts of its band's
// mask words (word index = row * nchunk + chunk, i.e. raster order) and every thread expands its words at its scanned
// offset -> the list is bit-exactly the reference's push_back order (cvfast.cpp:9237-9238) with no ordered atomics;
// lut[y] is the offset of the first word of row y (jni/KeyFrame.cc:43-49).  (One workgroup per (level, stream) walking the
// whole level was the fourth largest kernel of the path: the level-0 workgroup did nearly all the work.)
#define COMPACT_THREADS 256
__global__ __launch_bounds__(COMPACT_THREADS) void k_compact(FeArgs a) {
  __shared__ int wsum[COMPACT_THREADS / 64];
  __shared__ int sh_above;
  const int b = blockIdx.x, l = blockIdx.y, s = blockIdx.z;
  const int h = a.h[l], nchunk = a.nchunk[l], cap = a.cap[l];
  const int y0 = b * BAND;
  if (y0 >= h) return;
  const int nrows = min(BAND, h - y0);
  const int* rc = a.rowcnt[l] + (size_t)s * h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int above = 0;
  for (int y = threadIdx.x; y < y0; y += COMPACT_THREADS) above += rc[y];
  for (int d = 32; d > 0; d >>= 1) above += __shfl_xor(above, d);
  if (lane == 0) wsum[wave] = above;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < COMPACT_THREADS / 64; w++) t += wsum[w]; sh_above = t; }
  __syncthreads();
  const int first = sh_above;
  const int nw = nrows * nchunk;
  const unsigned long long* cm = a.cmask[l] + ((size_t)s * h + y0) * nchunk;
  uint32_t* out = a.corners[l] + (size_t)s * cap;
  int* lut = a.rowlut[l] + (size_t)s * (h + 1);
  const int per = (nw + COMPACT_THREADS - 1) / COMPACT_THREADS;
  const int lo = min((int)threadIdx.x * per, nw), hi = min(lo + per, nw);
  int cnt = 0;
  for (int i = lo; i < hi; i++) cnt += __popcll(cm[i]);
  int inc = cnt;
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
  __syncthreads();
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = first + inc - cnt, total = first;
  for (int w = 0; w < COMPACT_THREADS / 64; w++) { if (w < wave) base += wsum[w]; total += wsum[w]; }
  for (int i = lo; i < hi; i++) {
    unsigned long long m = cm[i];
    const int rr = i / nchunk, c = i - rr * nchunk, y = y0 + rr;
    if (c == 0) lut[y] = min(base, cap);
    while (m) {
      const int bit = __ffsll((long long)m) - 1;
      if (base < cap) out[base] = (uint32_t)((c << 6) + bit) | ((uint32_t)y << 16);
      base++;
      m &= m - 1;
    }
  }
  if (threadIdx.x == 0 && y0 + nrows == h) {                         // the last band of the level closes the list
    lut[h] = min(total, cap);
    a.ncorners[s * NLEV + l] = min(total, cap);
    if (total > cap) *a.overflow = 1;
  }
}

// ---- MakeKeyFrame_Rest: FAST score + non-max (jni/KeyFrame.cc:53-63) ---------------------------------------------
struct NmArgs {
  const uint8_t* img[NLEV]; size_t img_sstride[NLEV]; int img_pitch[NLEV];
  int h[NLEV], cap[NLEV];
  const uint32_t* corners[NLEV]; const int* rowlut[NLEV]; const int* ncorners;
  int* scores[NLEV]; uint32_t* maxcorners[NLEV]; int* nmax;
  int barrier, quirk;
  const TrackerState* gate;   // non-null: only the streams whose tracker asked for a keyframe (kf_pending)
};

// compute_fast_score_old (cvfast.cpp:9337-9393): one lane per corner.
__global__ __launch_bounds__(FE_THREADS) void k_score(NmArgs a) {
  const int l = blockIdx.y, s = blockIdx.z;
  if (a.gate && !a.gate[s].kf_pending) return;
  const int n = a.ncorners[s * NLEV + l];
  for (int i = blockIdx.x * FE_THREADS + threadIdx.x; i < n; i += gridDim.x * FE_THREADS) {
  const uint32_t cxy = a.corners[l][(size_t)s * a.cap[l] + i];
  const int x = cxy & 0xFFFF, y = cxy >> 16, lp = a.img_pitch[l];
  const uint8_t* p = a.img[l] + (size_t)s * a.img_sstride[l] + (size_t)y * lp + x;
  const int cb = p[0] + a.barrier, c_b = p[0] - a.barrier;
  int sp = 0, sn = 0;
#define SC(dx, dy) { const int v = p[(dx) + (dy) * lp]; if (v > cb) sp += v - cb; else if (v < c_b) sn += c_b - v; }
  SC(0, 3) SC(1, 3) SC(2, 2) SC(3, 1) SC(3, 0) SC(3, -1) SC(2, -2) SC(1, -3)
  SC(0, -3) SC(-1, -3) SC(-2, -2) SC(-3, -1) SC(-3, 0) SC(-3, 1) SC(-2, 2) SC(-1, 3)
#undef SC
  a.scores[l][(size_t)s * a.cap[l] + i] = sp > sn ? sp : sn;
  }
}

// nonmax_suppression (cvfast.cpp:9243-9335): a corner survives unless a corner among its 8 neighbours
// has a strictly greater score; rows above/below are found through the row LUT.  One workgroup per
// (stream, level) keeps the output in raster order with a block scan of the keep flags.
__global__ __launch_bounds__(FE_THREADS) void k_nonmax(NmArgs a) {
  __shared__ int wsum[8];
  __shared__ int carry;
  const int l = blockIdx.x, s = blockIdx.y;
  if (a.gate && !a.gate[s].kf_pending) return;
  const int n = a.ncorners[s * NLEV + l], cap = a.cap[l], h = a.h[l];
  const uint32_t* cs = a.corners[l] + (size_t)s * cap;
  const int* sc = a.scores[l] + (size_t)s * cap;
  const int* lut = a.rowlut[l] + (size_t)s * (h + 1);
  uint32_t* out = a.maxcorners[l] + (size_t)s * cap;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i0 = 0; i0 < n; i0 += FE_THREADS) {
    const int i = i0 + threadIdx.x;
    int keep = 0;
    uint32_t me = 0;
    if (i < n) {
      me = cs[i];
      const int x = me & 0xFFFF, y = me >> 16, score = sc[i];
      keep = 1;
      if (i > 0) {                                                     // check left  :9276-9279
        const uint32_t c = cs[i - 1];
        if ((int)(c & 0xFFFF) == x - 1 && (int)(c >> 16) == y && sc[i - 1] > score) keep = 0;
      }
      if (keep && i < n - 1) {                                         // check right :9281-9285
        const uint32_t c = cs[i + 1];
        if (a.quirk) {  // reference tests corners[i-1](1) == pos(1); i == 0 reads out of bounds there -> no match
          if (i > 0 && (int)(c & 0xFFFF) == x + 1 && (int)(cs[i - 1] >> 16) == y && sc[i + 1] > score) keep = 0;
        } else {
          if ((int)(c & 0xFFFF) == x + 1 && (int)(c >> 16) == y && sc[i + 1] > score) keep = 0;
        }
      }
      if (keep && y > 0) {                                             // check above :9287-9306
        for (int j = lut[y - 1]; j < lut[y]; j++) {
          const int xx = cs[j] & 0xFFFF;
          if (xx > x + 1) break;
          if (xx >= x - 1 && sc[j] > score) { keep = 0; break; }
        }
      }
      if (keep && y + 1 < h) {                                         // check below :9308-9326
        for (int j = lut[y + 1]; j < lut[y + 2 > h ? h : y + 2]; j++) {
          const int xx = cs[j] & 0xFFFF;
          if (xx > x + 1) break;
          if (xx >= x - 1 && sc[j] > score) { keep = 0; break; }
        }
      }
    }
    int inc = keep;
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = carry;
    for (int w = 0; w < wave; w++) base += wsum[w];
    if (keep) out[base + inc - 1] = me;
    __syncthreads();
    if (threadIdx.x == FE_THREADS - 1) carry = base + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) a.nmax[s * NLEV + l] = carry;
}

// KeyFrame::MakeKeyFrame_Rest candidate loop (jni/KeyFrame.cc:66-95): maximal corners inside the 10-px border whose
// Shi-Tomasi score (FindShiTomasiScoreAtPoint, jni/vision/ImageHandler.cpp:124-155, half-window 3) exceeds the minimum,
// in raster order.  One workgroup per (level, stream); one lane per maximal corner; block scan keeps the order.
// The three gradient sums are integers (exact in fp64 in any order), the rest follows the reference expression.
struct CandArgs {
  const uint8_t* img[NLEV]; size_t img_sstride[NLEV]; int img_pitch[NLEV];
  int w[NLEV], h[NLEV], cap[NLEV];
  const uint32_t* maxcorners[NLEV]; const int* nmax;
  uint32_t* cand[NLEV]; double* cand_score[NLEV]; int* ncand;
  double min_score; int border;
  const TrackerState* gate;
};

__device__ __forceinline__ double shi_tomasi7(const uint8_t* img, int pitch, int px, int py) {
  int sxx = 0, syy = 0, sxy = 0;
  for (int cy = py - 3; cy <= py + 3; cy++) {
    const uint8_t* r = img + (size_t)cy * pitch;
#pragma unroll
    for (int cx = -3; cx <= 3; cx++) {
      const int dx = (int)r[px + cx + 1] - (int)r[px + cx - 1];
      const int dy = (int)r[px + cx + pitch] - (int)r[px + cx - pitch];
      sxx += dx * dx; syy += dy * dy; sxy += dx * dy;
    }
  }
  const int nPixels = 49;
  const double dXX = (double)sxx / (2.0 * nPixels), dYY = (double)syy / (2.0 * nPixels), dXY = (double)sxy / (2.0 * nPixels);
  return 0.5 * (dXX + dYY - sqrt((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY)));
}

// ordered append of the lanes with keep != 0 (block-wide, raster order preserved); returns the new carry
__device__ __forceinline__ int block_ordered_slot(int keep, int* wsum, int* carry, int& slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = keep;
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = *carry;
  for (int w = 0; w < wave; w++) base += wsum[w];
  slot = base + inc - 1;
  __syncthreads();
  if (threadIdx.x == FE_THREADS - 1) *carry = base + inc;
  __syncthreads();
  return 0;
}

__global__ __launch_bounds__(FE_THREADS) void k_candidates(CandArgs a) {
  __shared__ int wsum[FE_THREADS / 64];
  __shared__ int carry;
  const int l = blockIdx.x, s = blockIdx.y;
  if (a.gate && !a.gate[s].kf_pending) return;
  const int n = a.nmax[s * NLEV + l], cap = a.cap[l];
  const uint32_t* mc = a.maxcorners[l] + (size_t)s * cap;
  const uint8_t* img = a.img[l] + (size_t)s * a.img_sstride[l];
  uint32_t* out = a.cand[l] + (size_t)s * cap;
  double* outs = a.cand_score[l] + (size_t)s * cap;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += FE_THREADS) {
    const int i = i0 + threadIdx.x;
    int keep = 0; uint32_t me = 0; double st = 0.0;
    if (i < n) {
      me = mc[i];
      const int x = me & 0xFFFF, y = me >> 16;
      if (x >= a.border && y >= a.border && x < a.w[l] - a.border && y < a.h[l] - a.border) {   // :72-73
        st = shi_tomasi7(img, a.img_pitch[l], x, y);
        keep = st > a.min_score;                                                                // :81
      }
    }
    int slot;
    block_ordered_slot(keep, wsum, &carry, slot);
    if (keep) { out[slot] = me; outs[slot] = st; }
  }
  __syncthreads();
  if (threadIdx.x == 0) a.ncand[s * NLEV + l] = carry;
}

// MapMaker::ThinCandidates (jni/MapMaker.cc:393-422) on every level: a candidate survives when no measurement of the
// keyframe at the same level or one level up lies within 10 level-pixels (rounded(), :381-386).  One workgroup per
// (level, stream); the busy positions are staged in LDS; the surviving list is compacted in place, order kept.
#define THIN_BUSY_CAP 4096
struct ThinArgs {
  uint32_t* cand[NLEV]; double* cand_score[NLEV]; int* ncand; int cap[NLEV];
  const MeasDev* meas; size_t meas_sstride;   // measurement row of the keyframe per stream
  int new_kf;       // 1: gated on kf_pending and against the keyframe slot k_add_keyframe has just filled (slot n_kf)
  int only_level;   // >= 0: this level only
  size_t kf_row;    // max_points (elements per keyframe row) when new_kf
};

__global__ __launch_bounds__(FE_THREADS) void k_thin_candidates(ThinArgs a, const TrackerState* st) {
  __shared__ int busy[THIN_BUSY_CAP * 2];
  __shared__ int nbusy;
  __shared__ int wsum[FE_THREADS / 64];
  __shared__ int carry;
  const int l = a.only_level >= 0 ? a.only_level : blockIdx.x, s = blockIdx.y;
  if (a.new_kf && !st[s].kf_pending) return;
  const MeasDev* ms = a.meas + (size_t)s * a.meas_sstride + (a.new_kf ? (size_t)st[s].n_kf * a.kf_row : 0);
  const int np = st[s].n_points;
  if (threadIdx.x == 0) { nbusy = 0; carry = 0; }
  __syncthreads();
  const int scale = 1 << l;
  for (int i = threadIdx.x; i < np; i += FE_THREADS) {
    const MeasDev m = ms[i];
    if (!m.valid || !(m.level == l || m.level == l + 1)) continue;
    const double vx = m.root[0] / scale, vy = m.root[1] / scale;
    const int k = atomicAdd(&nbusy, 1);
    if (k < THIN_BUSY_CAP) { busy[2 * k] = (int)(vx > 0.0 ? vx + 0.5 : vx - 0.5); busy[2 * k + 1] = (int)(vy > 0.0 ? vy + 0.5 : vy - 0.5); }
  }
  __syncthreads();
  const int nb = nbusy < THIN_BUSY_CAP ? nbusy : THIN_BUSY_CAP;
  const int n = a.ncand[s * NLEV + l];
  uint32_t* cp = a.cand[l] + (size_t)s * a.cap[l];
  double* cs = a.cand_score[l] + (size_t)s * a.cap[l];
  for (int i0 = 0; i0 < n; i0 += FE_THREADS) {
    const int i = i0 + threadIdx.x;
    int keep = 0; uint32_t me = 0; double sc = 0.0;
    if (i < n) {
      me = cp[i]; sc = cs[i];
      const int cx = me & 0xFFFF, cy = me >> 16;
      keep = 1;
      for (int j = 0; j < nb; j++) {
        const int dx = busy[2 * j] - cx, dy = busy[2 * j + 1] - cy;
        if (dx * dx + dy * dy < 100) { keep = 0; break; }
      }
    }
    int slot;
    block_ordered_slot(keep, wsum, &carry, slot);          // barriers inside: every read of this chunk precedes its writes
    if (keep) { cp[slot] = me; cs[slot] = sc; }            // slot <= i: in-place compaction never overtakes the reads
  }
  __syncthreads();
  if (threadIdx.x == 0) a.ncand[s * NLEV + l] = carry;
}

// ---- host side ------------------------------------------------------------------------------------------------
static void fill_fe_args(vslam_system* sys, FeArgs& a) {
  for (int l = 0; l < NLEV; l++) {
    a.lvl[l] = sys->d_lvl[l];
    a.lvl_sstride[l] = (size_t)sys->geom[l].pitch * sys->geom[l].h;
    a.lvl_pitch[l] = sys->geom[l].pitch;
    a.w[l] = sys->geom[l].w; a.h[l] = sys->geom[l].h; a.nchunk[l] = sys->geom[l].nchunk;
    a.thr[l] = sys->geom[l].thr; a.cap[l] = sys->geom[l].cap;
    a.cmask[l] = sys->fr.cmask[l]; a.rowcnt[l] = sys->fr.rowcnt[l]; a.rowlut[l] = sys->fr.rowlut[l];
    a.corners[l] = sys->fr.corners[l];
  }
  a.ncorners = sys->fr.ncorners;
  a.overflow = sys->fr.overflow;
}

int fe_make_keyframe_lite(vslam_system* sys, const uint8_t* gray, size_t row_stride, size_t stream_stride,
                          int on_device) {
  const LevelGeom* g = sys->geom;
  if (!gray || (int)row_stride < g[0].w) { vslam_set_error("make_keyframe_lite: bad image arguments"); return VSLAM_E_INVALID; }
  // next front-end buffer; it may be rebuilt once the tracking that last read it (two frames ago) has finished
  const int b = sys->fr_idx ^ 1;
  sys->fr_idx = b;
  sys->fr = sys->frbuf[b];
  for (int l = 0; l < NLEV; l++) sys->d_lvl[l] = sys->d_lvl_buf[b][l];
  hipStream_t fs = sys->fe_stream;
  HIPCHK(hipStreamWaitEvent(fs, sys->ev_track_done[b], 0));
  FeArgs a;
  fill_fe_args(sys, a);
  if (on_device) {
    a.in = gray; a.in_sstride = stream_stride; a.in_pitch = (int)row_stride;
  } else {
    // host input: copy into the owned level-0 image (TrackFrame copies its input too, jni/KeyFrame.cc:12)
    for (int s = 0; s < sys->S; s++)
      HIPCHK(hipMemcpy2DAsync(sys->d_lvl[0] + (size_t)s * a.lvl_sstride[0], g[0].pitch, gray + (size_t)s * stream_stride,
                              row_stride, g[0].w, g[0].h, hipMemcpyHostToDevice, fs));
    a.in = sys->d_lvl[0]; a.in_sstride = a.lvl_sstride[0]; a.in_pitch = g[0].pitch;
  }
  sys->fr.img[0] = a.in; sys->fr.img_sstride[0] = a.in_sstride; sys->fr.img_pitch[0] = a.in_pitch;
  for (int l = 1; l < NLEV; l++) { sys->fr.img[l] = a.lvl[l]; sys->fr.img_sstride[l] = a.lvl_sstride[l]; sys->fr.img_pitch[l] = a.lvl_pitch[l]; }
  sys->frbuf[b] = sys->fr;

  const int lp0 = (g[0].w + 15) & ~15, lp1 = (g[1].w + 15) & ~15, lp2 = (g[2].w + 15) & ~15;
  const size_t lds0 = (size_t)(BAND + 2 * HALO) * lp0 + (BAND / 2) * lp1 + (BAND / 4) * lp2 + 16 + (size_t)BAND * g[0].nchunk * 8 + (size_t)FB_ROWS * lp0 * 2 + 16;
  const int nb0 = (g[0].h + BAND - 1) / BAND;
  prof_mark(sys, 0);
  hipLaunchKernelGGL(k_pyr_fast0, dim3(nb0, sys->S), dim3(FE_THREADS), lds0, fs, a, lp0, lp1, lp2);
  int nb = 0;
  a.band_first[0] = 0;
  for (int l = 1; l < NLEV; l++) { a.band_first[l] = nb; nb += (g[l].h + BAND - 1) / BAND; }
  const size_t lds1 = (size_t)(BAND + 2 * HALO) * lp1 + 16 + (size_t)BAND * g[1].nchunk * 8 + (size_t)FB_ROWS * lp1 * 2 + 16;
  prof_mark(sys, 1);
  hipLaunchKernelGGL(k_fast_lvl, dim3(nb, sys->S), dim3(FE_THREADS), lds1, fs, a);
  prof_mark(sys, 2);
  hipLaunchKernelGGL(k_compact, dim3(nb0, NLEV, sys->S), dim3(COMPACT_THREADS), 0, fs, a);
  if (sys->p.use_sbi) {                                           // jni/Tracker.cc:86-97, 104-105
    int r = fe_sbi(sys, sys->have_sbi ? sys->frbuf[b ^ 1] : sys->fr);
    if (r) return r;
    sys->have_sbi = true;
  }
  prof_mark(sys, PROF_FE_END);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(sys->ev_fe_done[b], fs));
  HIPCHK(hipStreamWaitEvent(sys->stream, sys->ev_fe_done[b], 0));   // everything on the main stream sees the new frame
  sys->have_frame = true;
  return VSLAM_OK;
}

static int fe_nonmax_impl(vslam_system* sys, bool gated) {
  if (!sys->have_frame) { vslam_set_error("fast_nonmax: no current frame"); return VSLAM_E_STATE; }
  NmArgs a;
  int maxcap = 0;
  for (int l = 0; l < NLEV; l++) {
    a.img[l] = sys->fr.img[l]; a.img_sstride[l] = sys->fr.img_sstride[l]; a.img_pitch[l] = sys->fr.img_pitch[l];
    a.h[l] = sys->geom[l].h; a.cap[l] = sys->geom[l].cap;
    a.corners[l] = sys->fr.corners[l]; a.rowlut[l] = sys->fr.rowlut[l];
    a.scores[l] = sys->fr.scores[l]; a.maxcorners[l] = sys->fr.maxcorners[l];
    if (a.cap[l] > maxcap) maxcap = a.cap[l];
  }
  a.ncorners = sys->fr.ncorners; a.nmax = sys->fr.nmax;
  a.barrier = sys->p.nonmax_barrier;
  a.quirk = (sys->p.quirks & VSLAM_Q_NONMAX_RIGHT_NEIGHBOUR) ? 1 : 0;
  a.gate = gated ? sys->map.st : nullptr;
  (void)maxcap;
  hipLaunchKernelGGL(k_score, dim3(16, NLEV, sys->S), dim3(FE_THREADS), 0, sys->stream, a);
  hipLaunchKernelGGL(k_nonmax, dim3(NLEV, sys->S), dim3(FE_THREADS), 0, sys->stream, a);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

int fe_fast_nonmax(vslam_system* sys) { return fe_nonmax_impl(sys, false); }

// KeyFrame::MakeKeyFrame_Rest (jni/KeyFrame.cc:53-95) for the current frame of every stream: fast_nonmax + candidates.
static int fe_rest_impl(vslam_system* sys, double min_score, bool gated) {
  int r = fe_nonmax_impl(sys, gated);
  if (r) return r;
  CandArgs a;
  for (int l = 0; l < NLEV; l++) {
    a.img[l] = sys->fr.img[l]; a.img_sstride[l] = sys->fr.img_sstride[l]; a.img_pitch[l] = sys->fr.img_pitch[l];
    a.w[l] = sys->geom[l].w; a.h[l] = sys->geom[l].h; a.cap[l] = sys->geom[l].cap;
    a.maxcorners[l] = sys->fr.maxcorners[l];
    a.cand[l] = sys->cand[l]; a.cand_score[l] = sys->cand_score[l];
  }
  a.nmax = sys->fr.nmax; a.ncand = sys->ncand;
  a.min_score = min_score; a.border = 10;                            // gvdCandidateMinSTScore / border, jni/KeyFrame.cc:57,65
  a.gate = gated ? sys->map.st : nullptr;
  hipLaunchKernelGGL(k_candidates, dim3(NLEV, sys->S), dim3(FE_THREADS), 0, sys->stream, a);
  HIPCHK(hipGetLastError());
  sys->have_candidates = true;
  return VSLAM_OK;
}
int fe_make_keyframe_rest(vslam_system* sys, double min_score) { return fe_rest_impl(sys, min_score, false); }
int fe_keyframe_rest_gated(vslam_system* sys) { return fe_rest_impl(sys, 70.0, true); }   // gvdCandidateMinSTScore, jni/KeyFrame.cc:57

int fe_thin_candidates(vslam_system* sys, int keyframe) {
  if (!sys->have_candidates) { vslam_set_error("thin_candidates: call vslam_make_keyframe_rest first"); return VSLAM_E_STATE; }
  if (keyframe >= sys->p.max_keyframes) { vslam_set_error("thin_candidates: bad keyframe"); return VSLAM_E_INVALID; }
  ThinArgs a;
  for (int l = 0; l < NLEV; l++) { a.cand[l] = sys->cand[l]; a.cand_score[l] = sys->cand_score[l]; a.cap[l] = sys->geom[l].cap; }
  a.ncand = sys->ncand;
  const size_t P = sys->p.max_points, K = sys->p.max_keyframes;
  a.new_kf = 0; a.only_level = -1; a.kf_row = 0;
  if (keyframe < 0) { a.meas = sys->map.cur_meas; a.meas_sstride = P; }           // the tracker's measurements of this frame
  else { a.meas = sys->map.kf_meas + (size_t)keyframe * P; a.meas_sstride = K * P; }
  hipLaunchKernelGGL(k_thin_candidates, dim3(NLEV, sys->S), dim3(FE_THREADS), 0, sys->stream, a, sys->map.st);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

// MapMaker::ThinCandidates(new keyframe, level) inside AddSomeMapPoints (jni/MapMaker.cc:432): against the measurement row of
// the slot k_add_keyframe has just filled, which also holds the points the earlier levels of this keyframe have added.
int fe_thin_new_keyframe(vslam_system* sys, int level) {
  ThinArgs a;
  for (int l = 0; l < NLEV; l++) { a.cand[l] = sys->cand[l]; a.cand_score[l] = sys->cand_score[l]; a.cap[l] = sys->geom[l].cap; }
  a.ncand = sys->ncand;
  const size_t P = sys->p.max_points, K = sys->p.max_keyframes;
  a.meas = sys->map.kf_meas; a.meas_sstride = K * P; a.new_kf = 1; a.only_level = level; a.kf_row = P;
  hipLaunchKernelGGL(k_thin_candidates, dim3(1, sys->S), dim3(FE_THREADS), 0, sys->stream, a, sys->map.st);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

// Level::vCorners of a stored keyframe (map upload with grow_map): the front-end kernels on a one-stream view whose level
// images are the keyframe's own storage; the mask / list scratch is the front-end buffer that is not the current frame's.
__global__ void k_store_kf_corners(const uint32_t* c0, const uint32_t* c1, const uint32_t* c2, const uint32_t* c3, const int* ncorners,
                                   uint32_t* d0, uint32_t* d1, uint32_t* d2, uint32_t* d3, int* dn, int k0, int k1, int k2, int k3) {
  const int l = blockIdx.x;
  const uint32_t* src = l == 0 ? c0 : (l == 1 ? c1 : (l == 2 ? c2 : c3));
  uint32_t* dst = l == 0 ? d0 : (l == 1 ? d1 : (l == 2 ? d2 : d3));
  const int kc = l == 0 ? k0 : (l == 1 ? k1 : (l == 2 ? k2 : k3));
  const int n = ncorners[l] < kc ? ncorners[l] : kc;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x == 0) dn[l] = n;
}

int fe_keyframe_corners(vslam_system* sys, int s, int kf) {
  const LevelGeom* g = sys->geom;
  const FrameDev& scr = sys->frbuf[sys->fr_idx ^ 1];
  const size_t K = sys->p.max_keyframes;
  FeArgs a;
  for (int l = 0; l < NLEV; l++) {
    uint8_t* img = sys->map.kf_img[l] + ((size_t)s * K + kf) * (size_t)g[l].pitch * g[l].h;
    a.lvl[l] = img; a.lvl_sstride[l] = 0; a.lvl_pitch[l] = g[l].pitch;
    a.w[l] = g[l].w; a.h[l] = g[l].h; a.nchunk[l] = g[l].nchunk; a.thr[l] = g[l].thr; a.cap[l] = g[l].cap;
    a.cmask[l] = scr.cmask[l]; a.rowcnt[l] = scr.rowcnt[l]; a.rowlut[l] = scr.rowlut[l]; a.corners[l] = scr.corners[l];
  }
  a.ncorners = scr.ncorners; a.overflow = scr.overflow;
  a.in = a.lvl[0]; a.in_sstride = 0; a.in_pitch = g[0].pitch;
  const int lp0 = (g[0].w + 15) & ~15, lp1 = (g[1].w + 15) & ~15, lp2 = (g[2].w + 15) & ~15;
  const size_t lds0 = (size_t)(BAND + 2 * HALO) * lp0 + (BAND / 2) * lp1 + (BAND / 4) * lp2 + 16 + (size_t)BAND * g[0].nchunk * 8 + (size_t)FB_ROWS * lp0 * 2 + 16;
  hipLaunchKernelGGL(k_pyr_fast0, dim3((g[0].h + BAND - 1) / BAND, 1), dim3(FE_THREADS), lds0, sys->stream, a, lp0, lp1, lp2);
  int nb = 0;
  a.band_first[0] = 0;
  for (int l = 1; l < NLEV; l++) { a.band_first[l] = nb; nb += (g[l].h + BAND - 1) / BAND; }
  const size_t lds1 = (size_t)(BAND + 2 * HALO) * lp1 + 16 + (size_t)BAND * g[1].nchunk * 8 + (size_t)FB_ROWS * lp1 * 2 + 16;
  hipLaunchKernelGGL(k_fast_lvl, dim3(nb, 1), dim3(FE_THREADS), lds1, sys->stream, a);
  hipLaunchKernelGGL(k_compact, dim3((g[0].h + BAND - 1) / BAND, NLEV, 1), dim3(COMPACT_THREADS), 0, sys->stream, a);
  uint32_t* d[NLEV];
  for (int l = 0; l < NLEV; l++) d[l] = sys->map.kf_corners[l] + ((size_t)s * K + kf) * sys->tp.kcap[l];
  hipLaunchKernelGGL(k_store_kf_corners, dim3(NLEV), dim3(256), 0, sys->stream, scr.corners[0], scr.corners[1], scr.corners[2], scr.corners[3], scr.ncorners,
                     d[0], d[1], d[2], d[3], sys->map.kf_ncorners + ((size_t)s * K + kf) * NLEV, sys->tp.kcap[0], sys->tp.kcap[1], sys->tp.kcap[2], sys->tp.kcap[3]);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(sys->stream));
  return VSLAM_OK;
}
