// Frame front-end on gfx950: KeyFrame::MakeKeyFrame_Lite (jni/KeyFrame.cc:5-51) and fast_nonmax
// (jni/vision/cvfast.cpp:9243-9400) for all streams of a system at once.
//
// Kernels (byte/integer work, one pass over each level image; VALU-issue bound):
//   k_fast_slide<PYR>  the form used whenever the level widths are multiples of 4 and the input is 16-B aligned: a lane owns
//                 a strip 16 pixels wide, walks down it with a sliding window of seven rows in registers (no LDS staging),
//                 runs a branch-free quick reject on every pixel, compacts the ~2 % survivors per wavefront and runs the
//                 16-pixel run-of-10 test on them; level 0 also writes its share of levels 1..3 from the same registers.
//   k_pyr_fast0 / k_fast_lvl   the general path (any width / alignment): one workgroup per 16-row band staged in LDS,
//                 pyramid from LDS, quick reject + compacted full test per 4 rows, corner-mask words in LDS.
//   k_compact     per (band of rows, level, stream): the workgroup adds the row counts above its band, scans the popcounts
//                 of its mask words (raster order) and every thread expands its words at its scanned offset -> bit-exact
//                 raster-ordered corner list and row LUT (jni/KeyFrame.cc:43-49) without ordered atomics.
//   k_score / k_nonmax   compute_fast_score_old + nonmax_suppression, one lane per corner.
//   k_candidates / k_thin_candidates   MakeKeyFrame_Rest's Shi-Tomasi candidates (jni/KeyFrame.cc:66-95) and
//                 MapMaker::ThinCandidates (jni/MapMaker.cc:393-422), ordered block compaction.
#include "vslam_internal.h"
#include <stdlib.h>
#include <stdio.h>

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

// ---------------------------------------------------------------------------------------------------------------
// The same FAST-10 sweep WITHOUT staging the image in LDS (the form used whenever the level widths are multiples of 4 and
// the input is 16-B aligned; the band kernels above stay as the general path).  One lane owns a strip 16 pixels wide and
// SL_R rows tall and walks down it: one 16-B global load per row, a sliding window of seven rows in registers -- the centre
// row's neighbours 3 to the left / right come from the neighbouring lanes' registers (wave shuffles; the two edge lanes of a
// wavefront load one extra dword), the pixels 3 above / below are the window's first and last row -- so the quick reject
// reads nothing but registers (it was LDS-issue bound).  Level 0 also produces its share of levels
// 1..3 from the same registers (2x2 sums by v_perm + v_sad_u8) with lane-contiguous stores.
// Quick reject (cvfast.cpp:6088-9241 as a filter): ten contiguous ring pixels contain two ADJACENT compass pixels
// (0, 4, 8, 12), so a corner has such a pair both brighter than c + t or both darker than c - t.  ~2 % of the pixels of the
// feeder's frames pass (0.4 % are corners).  Every seven rows a wavefront compacts its survivors (prefix sum of the lanes'
// bit counts) into its own LDS list and runs the full 16-pixel run-of-10 test on them with dense lanes, the ring read back
// through L1/L2 as aligned dwords; corners are OR-ed into the workgroup's mask words in LDS, which leave with coalesced
// stores together with the row counts.  A workgroup owns 256 / (w / 16) whole strips-rows ("bands") of consecutive frames.
#define SL_RMAX 64         // most rows of a strip; the height R (a multiple of 8: the three halvings stay inside a strip) is chosen per launch
#define SL_THREADS 256
#define SL_CAP 1024        // entries of a wavefront's candidate list (one row of a wavefront: 64 lanes x 16 pixels)

struct SlArgs {
  int w16[NLEV], nb[NLEV], bpw[NLEV], wg_first[NLEV + 1];   // strips per row, bands per frame, bands per workgroup, first workgroup of a level
  int R[NLEV];                                              // rows of a strip
  int S;
};

// sum of the 2x2 block (wa byte k, k+1; wb byte k, k+1) + 2, >> 2, for k = 0 and k = 2 -> two output pixels in bits 0..15
DEVFN unsigned sl_half2(unsigned wa, unsigned wb) {
  const unsigned q0 = __builtin_amdgcn_perm(wb, wa, 0x05040100u), q1 = __builtin_amdgcn_perm(wb, wa, 0x07060302u);
  return (__builtin_amdgcn_sad_u8(q0, 0u, 2u) >> 2) | ((__builtin_amdgcn_sad_u8(q1, 0u, 2u) >> 2) << 8);
}
DEVFN unsigned sl_half4(unsigned a0, unsigned a1, unsigned b0, unsigned b1) { return sl_half2(a0, b0) | (sl_half2(a1, b1) << 16); }

// full segment test of pixel (x, y) of an image read through the caches: per ring row the aligned dwords around x - 3
DEVFN bool sl_full_test(const uint8_t* img, int pitch, int w, int x, int y, int thr) {
  const int xb = (x - 3) & ~3, o = (x - 3) & 3;
  const int x2 = min(xb + 8, w - 4);                                 // the third dword is only needed (and then inside the row) when o >= 2
  unsigned A[7], B[7];
  const uint8_t* r = img + (size_t)(y - 3) * pitch;
#pragma unroll
  for (int k = 0; k < 7; k++) {
    const uint2 v = *(const uint2*)(r + xb);
    const unsigned t = *(const unsigned*)(r + x2);
    A[k] = __builtin_amdgcn_alignbyte(v.y, v.x, o);                  // pixels x-3 .. x
    B[k] = __builtin_amdgcn_alignbyte(t, v.y, o);                    // pixels x+1 .. x+4
    r += pitch;
  }
  const int c = (A[3] >> 24) & 255u, cb = c + thr, c_b = c - thr;
  unsigned mb = 0, md = 0;                                          // ring pixel k -> bit k: the sign bits of cb - v / v - c_b are shifted in, k = 15 first
#define SLPIX(k, dx, dy) { const int v = (int)(((dx) <= 0 ? (A[(dy) + 3] >> (8 * ((dx) + 3))) : (B[(dy) + 3] >> (8 * ((dx) - 1)))) & 255u); mb = __builtin_amdgcn_alignbit(mb, (unsigned)(cb - v), 31); md = __builtin_amdgcn_alignbit(md, (unsigned)(v - c_b), 31); }
  SLPIX(15, -1, 3) SLPIX(14, -2, 2) SLPIX(13, -3, 1) SLPIX(12, -3, 0) SLPIX(11, -3, -1) SLPIX(10, -2, -2)
  SLPIX(9, -1, -3) SLPIX(8, 0, -3) SLPIX(7, 1, -3) SLPIX(6, 2, -2) SLPIX(5, 3, -1) SLPIX(4, 3, 0)
  SLPIX(3, 3, 1) SLPIX(2, 2, 2) SLPIX(1, 1, 3) SLPIX(0, 0, 3)
#undef SLPIX
  return ring_run10(mb) || ring_run10(md);
}

template <bool PYR>
__global__ __launch_bounds__(SL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fast_slide(FeArgs a, SlArgs q) {
  extern __shared__ __align__(16) uint8_t lds[];
  int l = 0;
  if (!PYR) { l = 1; if ((int)blockIdx.x >= q.wg_first[2]) l = 2; if ((int)blockIdx.x >= q.wg_first[3]) l = 3; }
  const int wg = (int)blockIdx.x - (PYR ? 0 : q.wg_first[l]);
  const int w = a.w[l], h = a.h[l], nchunk = a.nchunk[l], thr = a.thr[l];
  const int W16 = q.w16[l], nb = q.nb[l], bpw = q.bpw[l], SL_R = q.R[l], SL_NSTEP = SL_R + 2 * HALO;
  const uint8_t* img = PYR ? a.in : a.lvl[l];
  const size_t sstride = PYR ? a.in_sstride : a.lvl_sstride[l];
  const int pitch = PYR ? a.in_pitch : a.lvl_pitch[l];
  int4* tb = (int4*)lds;                                             // per thread: stream, first row, first column, band of the workgroup
  unsigned short* lists = (unsigned short*)(tb + SL_THREADS);
  unsigned long long* mask = (unsigned long long*)(lists + (SL_THREADS / 64) * SL_CAP);   // [bpw][R][nchunk]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned short* list = lists + wave * SL_CAP;
  const int bl = tid / W16, cg = tid - bl * W16;
  const long gband = (long)wg * bpw + bl;
  const bool active = bl < bpw && gband < (long)q.S * nb;
  const int s = active ? (int)(gband / nb) : 0, b = active ? (int)(gband - (long)s * nb) : 0;
  const int yb = b * SL_R, x0 = cg * 16;
  tb[tid] = make_int4(s, yb, x0, active ? bl : -1);
  for (int i = tid; i < bpw * SL_R * nchunk; i += SL_THREADS) mask[i] = 0ull;
  __syncthreads();
  const uint8_t* base = img + (size_t)s * sstride + x0;
  unsigned xvalid = 0;                                               // columns of the strip FAST runs on (cvfast.cpp:6113-6117)
  for (int j = 0; j < 16; j++) if (x0 + j >= HALO && x0 + j < w - HALO) xvalid |= 1u << j;
  if (!active) xvalid = 0;
  const bool eL = active && lane == 0 && cg > 0, eR = active && lane == 63 && cg < W16 - 1;

  auto load_row = [&](int i, uint4& v, unsigned& vl, unsigned& vr) {  // row step i of the strip: image row yb - 3 + i
    const int y = yb - HALO + i;
    v = make_uint4(0, 0, 0, 0); vl = 0; vr = 0;
    if (active && i < SL_NSTEP && y >= 0 && y < h) {
      const uint8_t* p = base + (size_t)y * pitch;
      v = *(const uint4*)p;
      if (eL) vl = *(const unsigned*)(p - 4);
      if (eR) vr = *(const unsigned*)(p + 16);
    }
  };
  unsigned W[7][6];                                                  // the window: slot = row step % 7; [0] left neighbour dword, [1..4] own, [5] right
#pragma unroll
  for (int u = 0; u < 7; u++)
#pragma unroll
    for (int k = 0; k < 6; k++) W[u][k] = 0;
  uint4 pf0, pf1; unsigned pl0, pr0, pl1, pr1;
  load_row(0, pf0, pl0, pr0);
  load_row(1, pf1, pl1, pr1);
  unsigned l1p[2] = {0, 0}, l2p = 0;                                 // pyramid pieces waiting for their second row

  for (int i0 = 0; i0 < SL_NSTEP; i0 += 7) {
    unsigned pw[4] = {0, 0, 0, 0};                                   // quick-reject survivors of this round: 16 bits per row
#pragma unroll
    for (int u = 0; u < 7; u++) {
      const int i = i0 + u;
      if (i >= SL_NSTEP) continue;
      const uint4 cur = pf0; const unsigned cl = pl0, cr = pr0;
      pf0 = pf1; pl0 = pl1; pr0 = pr1;
      load_row(i + 2, pf1, pl1, pr1);
      unsigned nl = __shfl_up(cur.w, 1), nr = __shfl_down(cur.x, 1);
      if (lane == 0) nl = cl;
      if (lane == 63) nr = cr;
      W[u][0] = nl; W[u][1] = cur.x; W[u][2] = cur.y; W[u][3] = cur.z; W[u][4] = cur.w; W[u][5] = nr;
      if (PYR) {                                                     // jni/KeyFrame.cc:19-23 (2:1 area filter)
        const int rb = i - HALO, y = yb + rb;                        // row of the band
        if (active && rb >= 0 && rb < SL_R && y < h && (rb & 1)) {
          const int up = (u + 6) % 7;
          const unsigned o0 = sl_half4(W[up][1], W[up][2], W[u][1], W[u][2]), o1 = sl_half4(W[up][3], W[up][4], W[u][3], W[u][4]);
          if ((y >> 1) < a.h[1]) *(uint2*)(a.lvl[1] + (size_t)s * a.lvl_sstride[1] + (size_t)(y >> 1) * a.lvl_pitch[1] + cg * 8) = make_uint2(o0, o1);
          if ((rb & 3) == 3) {
            const unsigned o2 = sl_half4(l1p[0], l1p[1], o0, o1);
            if ((y >> 2) < a.h[2]) *(unsigned*)(a.lvl[2] + (size_t)s * a.lvl_sstride[2] + (size_t)(y >> 2) * a.lvl_pitch[2] + cg * 4) = o2;
            if ((rb & 7) == 7) {
              if ((y >> 3) < a.h[3]) *(unsigned short*)(a.lvl[3] + (size_t)s * a.lvl_sstride[3] + (size_t)(y >> 3) * a.lvl_pitch[3] + cg * 2) = (unsigned short)sl_half2(l2p, o2);
            } else l2p = o2;
          } else { l1p[0] = o0; l1p[1] = o1; }
        }
      }
      // quick reject on the centre row (row step i - 3 = slot (u + 4) % 7): band row rel = i - 6
      const int rel = i - 2 * HALO, yc = yb + rel;
      if (rel >= 0 && yc >= HALO && yc < h - HALO) {                  // uniform but for yc (per band)
        const int sc = (u + 4) % 7, sd = (u + 1) % 7;             // centre, row y - 3; row y + 3 is slot u
        // branch-free: a pair of adjacent compass pixels both above c + t  <=>  min(max(p0, p8), max(p4, p12)) - c > t, both
        // below c - t  <=>  c - max(min(p0, p8), min(p4, p12)) > t; the sign of t - max(...) is shifted into the row's bit word
        unsigned pass = 0;
#pragma unroll
        for (int d = 3; d >= 0; d--) {
          const unsigned cw = W[sc][1 + d], uw = W[u][1 + d], dw = W[sd][1 + d];
          const unsigned p12w = __builtin_amdgcn_alignbyte(cw, W[sc][d], 1);        // pixels x-3 .. x
          const unsigned p4w = __builtin_amdgcn_alignbyte(W[sc][2 + d], cw, 3);     // pixels x+3 .. x+6
#pragma unroll
          for (int j = 3; j >= 0; j--) {
            const int c = (cw >> (8 * j)) & 255u;
            const int p0 = (uw >> (8 * j)) & 255u, p8 = (dw >> (8 * j)) & 255u;
            const int p4 = (p4w >> (8 * j)) & 255u, p12 = (p12w >> (8 * j)) & 255u;
            const int bm = min(max(p0, p8), max(p4, p12)), dm = max(min(p0, p8), min(p4, p12));
            const int v = max(bm - c, c - dm);
            pass = __builtin_amdgcn_alignbit(pass, (unsigned)(thr - v), 31);
          }
        }
        pass &= xvalid;
        pw[u >> 1] |= pass << (16 * (u & 1));
      }
    }
    // ---- the round's survivors: compact per wavefront, full test on dense lanes ----
    const int relg = i0 - 2 * HALO;                                  // band row of the round's slot 0
    int cnt = __popc(pw[0]) + __popc(pw[1]) + __popc(pw[2]) + __popc(pw[3]);
    int tot = cnt;
    for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
#ifdef VSLAM_SL_SKIP_FLUSH
    if (tot != 123456) continue;
#endif
    if (tot == 0) continue;
    const int nsub = tot <= SL_CAP ? 1 : 7;                          // a round that overflows the list goes row by row
    for (int sub = 0; sub < nsub; sub++) {
      unsigned m4[4] = {pw[0], pw[1], pw[2], pw[3]};
      if (nsub > 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) m4[k] = (sub >> 1) == k ? (pw[k] & (0xFFFFu << (16 * (sub & 1)))) : 0u;
      }
      const int c4 = __popc(m4[0]) + __popc(m4[1]) + __popc(m4[2]) + __popc(m4[3]);
      int inc = c4;
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
      const int n = __shfl(inc, 63);
      int pos = inc - c4;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        unsigned m = m4[k];
        while (m) {
          const int bit = __ffs((int)m) - 1;
          list[pos++] = (unsigned short)((lane << 7) | (k * 32 + bit));
          m &= m - 1;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < n; k += 64) {
        const int e = list[k];
        const int4 t = tb[wave * 64 + (e >> 7)];
        const int rr = relg + ((e >> 4) & 7), x = t.z + (e & 15), y = t.y + rr;
#ifdef VSLAM_SL_SKIP_FULL
        if (x == 7 && y == 7)
#else
        if (sl_full_test(img + (size_t)t.x * sstride, pitch, w, x, y, thr))
#endif
          atomicOr(&mask[((size_t)t.w * SL_R + rr) * nchunk + (x >> 6)], 1ull << (x & 63));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  // masks and row counts of the workgroup's bands
  const int per = SL_R * nchunk;
  for (int i = tid; i < bpw * per; i += SL_THREADS) {
    const int bb = i / per, rem = i - bb * per, row = rem / nchunk, c = rem - row * nchunk;
    const int4 t = tb[bb * W16];
    if (t.w >= 0 && t.y + row < h) a.cmask[l][((size_t)t.x * h + t.y + row) * nchunk + c] = mask[i];
  }
  for (int i = tid; i < bpw * SL_R; i += SL_THREADS) {
    const int bb = i / SL_R, row = i - bb * SL_R;
    const int4 t = tb[bb * W16];
    if (t.w < 0 || t.y + row >= h) continue;
    int c = 0;
    for (int k = 0; k < nchunk; k++) c += __popcll(mask[(size_t)i * nchunk + k]);
    a.rowcnt[l][(size_t)t.x * h + t.y + row] = c;
  }
}

// Raster-ordered corner lists + row LUT from the corner bit-masks, one workgroup per (band of 16 rows, level, stream): the
// workgroup adds the row counts above its band (its list offset = lut of its first row), scans the popcounts of its band's
// mask words (word index = row * nchunk + chunk, i.e. raster order) and every thread expands its words at its scanned
// offset -> the list is bit-exactly the reference's push_back order (cvfast.cpp:9237-9238) with no ordered atomics;
// lut[y] is the offset of the first word of row y (jni/KeyFrame.cc:43-49).  (One workgroup per (level, stream) walking the
// whole level was the fourth largest kernel of the path: the level-0 workgroup did nearly all the work.)
#define COMPACT_THREADS 256
__global__ __launch_bounds__(COMPACT_THREADS) void k_compact(FeArgs a, int cband /* rows per workgroup */) {
  __shared__ int wsum[COMPACT_THREADS / 64];
  __shared__ int sh_above;
  const int b = blockIdx.x, l = blockIdx.y, s = blockIdx.z;
  const int h = a.h[l], nchunk = a.nchunk[l], cap = a.cap[l];
  const int y0 = b * cband;
  if (y0 >= h) return;
  const int nrows = min(cband, h - y0);
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

// the strip form applies when every level is a multiple of 4 wide and at most 4096, and the level-0 input can be read 16 B at a time
static bool fe_slide_ok(const vslam_system* sys, const FeArgs& a) {
  static const bool off = getenv("VSLAM_FE_BANDS") != nullptr;        // diagnostic: force the band kernels
  if (off) return false;
  for (int l = 0; l < NLEV; l++) if ((a.w[l] & 3) || a.w[l] < 16 || a.w[l] > 4096 || a.h[l] < 7) return false;
  return (((uintptr_t)a.in) & 15) == 0 && (a.in_pitch & 15) == 0 && (a.in_sstride & 15) == 0 && a.in_pitch >= ((a.w[0] + 15) & ~15);
}
static void fe_launch_slide(vslam_system* sys, FeArgs& a, hipStream_t fs, int S) {
  SlArgs q;
  q.S = S;
  int nwg[NLEV];
  size_t lds[NLEV];
  const long slots = 4L * (sys->n_cu > 0 ? sys->n_cu : 256);          // workgroups the device holds at once (four per CU)
  for (int l = 0; l < NLEV; l++) { q.w16[l] = (a.w[l] + 15) >> 4; q.bpw[l] = SL_THREADS / q.w16[l]; }
  // strip height per launch (level 0; levels 1..3): the one that costs the fewest row steps, rounds of resident workgroups x (R + halo)
  auto pick = [&](int l0, int l1) {
    long best = -1; int bestR = 32;
    for (int R = 16; R <= SL_RMAX; R += 8) {
      long wgs = 0;
      bool fits = true;                                                // 64 KB of dynamic LDS
      for (int l = l0; l <= l1; l++) {
        wgs += ((long)S * ((a.h[l] + R - 1) / R) + q.bpw[l] - 1) / q.bpw[l];
        if ((size_t)SL_THREADS * 16 + (size_t)(SL_THREADS / 64) * SL_CAP * 2 + (size_t)q.bpw[l] * R * a.nchunk[l] * 8 > 60000) fits = false;
      }
      if (!fits && R > 16) continue;
      const long cost = ((wgs + slots - 1) / slots) * (R + 2 * HALO);
      if (best < 0 || cost < best) { best = cost; bestR = R; }
    }
    for (int l = l0; l <= l1; l++) q.R[l] = bestR;
  };
  pick(0, 0);
  pick(1, NLEV - 1);
  for (int l = 0; l < NLEV; l++) {
    q.nb[l] = (a.h[l] + q.R[l] - 1) / q.R[l];
    nwg[l] = (int)(((long)S * q.nb[l] + q.bpw[l] - 1) / q.bpw[l]);
    lds[l] = (size_t)SL_THREADS * 16 + (size_t)(SL_THREADS / 64) * SL_CAP * 2 + (size_t)q.bpw[l] * q.R[l] * a.nchunk[l] * 8;
  }
  static const bool dbg = getenv("VSLAM_FE_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "fe slide: S %d n_cu %d R %d %d %d %d nb %d %d %d %d wgs %d %d %d %d lds %zu %zu %zu %zu\n", S, sys->n_cu, q.R[0], q.R[1], q.R[2], q.R[3],
                   q.nb[0], q.nb[1], q.nb[2], q.nb[3], nwg[0], nwg[1], nwg[2], nwg[3], lds[0], lds[1], lds[2], lds[3]);
  q.wg_first[0] = 0; q.wg_first[1] = 0;
  for (int l = 1; l < NLEV; l++) q.wg_first[l + 1] = q.wg_first[l] + nwg[l];
  size_t lds123 = lds[1]; if (lds[2] > lds123) lds123 = lds[2]; if (lds[3] > lds123) lds123 = lds[3];
  prof_mark(sys, 0);
  hipLaunchKernelGGL(k_fast_slide<true>, dim3(nwg[0]), dim3(SL_THREADS), lds[0], fs, a, q);
  prof_mark(sys, 1);
  hipLaunchKernelGGL(k_fast_slide<false>, dim3(q.wg_first[NLEV]), dim3(SL_THREADS), lds123, fs, a, q);
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
  // map bootstrap: the trail tracker of the frame in flight (k_trail_advance, boot.hip) still reads THIS buffer as its previous
  // frame (image, corners, row LUT) -- the buffer is free only when that frame has finished too
  if (sys->p.bootstrap) HIPCHK(hipStreamWaitEvent(fs, sys->ev_track_done[b ^ 1], 0));
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
  if (fe_slide_ok(sys, a)) fe_launch_slide(sys, a, fs, sys->S);
  else {
    prof_mark(sys, 0);
    hipLaunchKernelGGL(k_pyr_fast0, dim3(nb0, sys->S), dim3(FE_THREADS), lds0, fs, a, lp0, lp1, lp2);
    int nb = 0;
    a.band_first[0] = 0;
    for (int l = 1; l < NLEV; l++) { a.band_first[l] = nb; nb += (g[l].h + BAND - 1) / BAND; }
    const size_t lds1 = (size_t)(BAND + 2 * HALO) * lp1 + 16 + (size_t)BAND * g[1].nchunk * 8 + (size_t)FB_ROWS * lp1 * 2 + 16;
    prof_mark(sys, 1);
    hipLaunchKernelGGL(k_fast_lvl, dim3(nb, sys->S), dim3(FE_THREADS), lds1, fs, a);
  }
  prof_mark(sys, 2);
  {
    static const int cb = getenv("VSLAM_COMPACT_BAND") ? atoi(getenv("VSLAM_COMPACT_BAND")) : 32;   // rows of a level per workgroup of the compaction (measured alone, 1024 frames: 16 rows 96 us, 32: 62, 64: 96, 128: 78, 256: 156)
    hipLaunchKernelGGL(k_compact, dim3((g[0].h + cb - 1) / cb, NLEV, sys->S), dim3(COMPACT_THREADS), 0, fs, a, cb);
  }
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
  hipLaunchKernelGGL(k_compact, dim3((g[0].h + BAND - 1) / BAND, NLEV, 1), dim3(COMPACT_THREADS), 0, sys->stream, a, BAND);
  uint32_t* d[NLEV];
  for (int l = 0; l < NLEV; l++) d[l] = sys->map.kf_corners[l] + ((size_t)s * K + kf) * sys->tp.kcap[l];
  hipLaunchKernelGGL(k_store_kf_corners, dim3(NLEV), dim3(256), 0, sys->stream, scr.corners[0], scr.corners[1], scr.corners[2], scr.corners[3], scr.ncorners,
                     d[0], d[1], d[2], d[3], sys->map.kf_ncorners + ((size_t)s * K + kf) * NLEV, sys->tp.kcap[0], sys->tp.kcap[1], sys->tp.kcap[2], sys->tp.kcap[3]);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(sys->stream));
  return VSLAM_OK;
}
