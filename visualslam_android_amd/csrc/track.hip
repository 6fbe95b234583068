// Tracker::TrackMap (jni/Tracker.cc:358-626) for all streams of a system, fully device resident:
// no host round-trip inside a frame.  One launch per dependent stage, each batched over the streams:
//
//   k_motion   ApplyMotionModel (:781-798), one lane per stream
//   k_pvs      per map point TrackerData::Project / GetDerivsUnsafe
//              (jni/TrackerData.h:69-95) + PatchFinder::CalcSearchLevelAndWarpMatrix (jni/PatchFinder.cc:31-68)
//   k_plan     potentially-visible-set lists per level in map order (the reference's random_shuffle is the
//              identity permutation here), coarse-stage selection (:399-461) / fine-stage selection (:493-535)
//   k_searchN  4 (11x11) or 8 (8x8) patches per wavefront, one lane per template row: warped template
//              (MakeTemplateCoarseCont, transform_image), ZMSSD at the FAST corners of the row-LUT window
//              (FindPatchCoarse/ZMSSDAtPoint) with packed-byte dot products and group shuffles
//   k_subpixN  inverse-compositional sub-pixel refinement (MakeSubPixTemplate, IterateSubPix*) of the found patches
//              that carry a sub-pixel budget
//   k_pose     one workgroup (two wavefronts) per stream: the 10 Gauss-Newton iterations of a stage (:466-488 / :543-577):
//              re-projection, 2x6 Jacobians, Tukey sigma (radix select for the median), weighted normal equations summed
//              in the reference's order (one wavefront produces the add_mJ operands into LDS, lane q of the other walks
//              sum q), 6x6 solve, SE3 exp; then measurement export, scene depth, UpdateMotionModel,
//              AssessTrackingQuality and the new-keyframe decision (:594-625, :802-878, :128-132)
// Transcendentals come from vslam_libm.h (one source for this file and the oracle); nothing here may be contracted into FMAs.
#include "vslam_internal.h"

#define TRK_THREADS 256
#define SORT_CAP 4096

// ---------------------------------------------------------------------------------------------------------------
// TrackerData::Project (jni/TrackerData.h:69-87).  Returns true when Cam.Project ran (pr valid).
template <class T>
DEVFN bool td_project(T& td, int& flags, const double* pos, const Pose& pose, const CamModel& cam, CamProj& pr) {
  flags &= ~TDF_IN_IMAGE;
  double c[3];
  pose_xform(pose, pos, c);
  td.cam[0] = c[0]; td.cam[1] = c[1]; td.cam[2] = c[2];
  if (c[2] < 0.001) return false;
  const double implane[2] = {c[0] / c[2], c[1] / c[2]};
  if (implane[0] * implane[0] + implane[1] * implane[1] > cam.largest_radius * cam.largest_radius) return false;
  pr = cam_project(cam, implane[0], implane[1]);
  td.image[0] = pr.im[0]; td.image[1] = pr.im[1];
  if (pr.invalid) return true;
  if (td.image[0] < 0 || td.image[1] < 0 || td.image[0] > cam.size[0] || td.image[1] > cam.size[1]) return true;
  flags |= TDF_IN_IMAGE;
  return true;
}

// TrackerData::ProjectAndDerivs (:98-102); derivatives refreshed only for found points whose projection ran
// (see oracle/tracker.cpp td_project_and_derivs for the one deliberate deviation).
template <class T>
DEVFN void td_project_and_derivs(T& td, int& flags, const double* pos, const Pose& pose, const CamModel& cam) {
  CamProj pr;
  const bool projected = td_project(td, flags, pos, pose, cam, pr);
  if ((flags & TDF_FOUND) && projected) cam_derivs(cam, pr, td.derivs);
}

// TrackerData::CalcJacobian (:107-122)
template <class T>
DEVFN void td_calc_jacobian(const T& td, double* jac) {
  const double ooz = 1.0 / td.cam[2];
  const double c[3] = {td.cam[0], td.cam[1], td.cam[2]};
  const double d0 = td.derivs[0], d1 = td.derivs[1], d2 = td.derivs[2], d3 = td.derivs[3];
#pragma unroll
  for (int m = 0; m < 6; m++) {
    double f0, f1;
    se3_generator_motion(m, c, ooz, f0, f1);
    jac[m] = d0 * f0 + d1 * f1;
    jac[6 + m] = d2 * f0 + d3 * f1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
#ifdef VSLAM_PVS_OCC8
#define VSLAM_PVS_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define VSLAM_PVS_ATTR
#endif
// Map points (104-byte structures) and TrackData (168-byte structures) are arrays of structures: a lane-per-point access is
// 64 separate 8-byte requests per instruction, and it was the REQUEST rate of L2 that set k_pvs's pace, not the bytes.  Both
// directions go through LDS: the workgroup's 256 points come in as consecutive 8-byte words (26 KB), and the 13 doubles a
// point writes (cam, image, derivs, warp_inv -- or only the first 3 / 5 of them, exactly as far as the reference's early
// returns get) leave with consecutive lanes on consecutive words of a structure, which the memory pipeline merges.
#define PVS_OUT 13                                                 // doubles of a TrackData that k_pvs writes: [0, 9) and [12, 16)
struct PvsOut { double cam[3], image[2], derivs[4], warp_inv[4]; };
static_assert(sizeof(PvsOut) == PVS_OUT * 8 && sizeof(MapPointDev) == PVS_OUT * 8 && sizeof(MapPointDev) % 8 == 0, "one LDS buffer serves both directions");
static_assert(offsetof(TrackData, derivs) == 40 && offsetof(TrackData, warp_inv) == 96, "k_pvs's write-out indexes TrackData as doubles");
// ApplyMotionModel (jni/Tracker.cc:781-798) and the per-frame reset of the tracker's counters, one lane per stream: in k_pvs every
// workgroup of a stream repeated the prediction on one lane (a serial se3_exp behind two dependent loads) while its other 255 waited.
__global__ __launch_bounds__(64) void k_motion(MapDev m, int S, const double* sbi_rot /* [S][8] or null */) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= S) return;
  TrackerState* st = &m.st[s];
  const bool tracking = st->map_good && st->lost_frames < 3;       // jni/Tracker.cc:103-104
  st->frame++;                                                     // :100
  st->kf_pending = 0; st->kf_added = 0;
  if (!tracking) return;
  double v[6];
  for (int i = 0; i < 6; i++) v[i] = st->velocity[i];
  if (sbi_rot) { v[0] = 0.0; v[1] = 0.0; for (int i = 3; i < 6; i++) v[i] = sbi_rot[(size_t)s * 8 + i]; }   // mbUseSBIInit :788-794
  const Pose pred = pose_mul(se3_exp(v), st->pose_final);
  st->start_pose = st->pose_final; st->pose_cur = pred;
  for (int l = 0; l < NLEV; l++) { st->attempted[l] = 0; st->found[l] = 0; }   // :360-361
  st->n_search = 0; st->n_coarse = 0; st->n_iter = 0; st->n_l3 = 0; st->coarse_found = 0;
}

__global__ __launch_bounds__(TRK_THREADS) VSLAM_PVS_ATTR void k_pvs(MapDev m, TrackParams tp) {
  const int s = blockIdx.y;
  TrackerState* st = &m.st[s];
  const bool tracking = st->map_good && st->lost_frames < 3;       // jni/Tracker.cc:103-104
  const int n_points = st->n_points, i0 = blockIdx.x * TRK_THREADS;
  if (!tracking || i0 >= n_points) return;
  __shared__ __align__(16) double pbuf[TRK_THREADS * PVS_OUT];
  __shared__ int wn[TRK_THREADS];
  const int nhere = min(TRK_THREADS, n_points - i0);
  {
    const double* src = (const double*)(m.pts + (size_t)s * tp.max_points + i0);
    for (int k = threadIdx.x; k < nhere * PVS_OUT; k += TRK_THREADS) pbuf[k] = src[k];
  }
  const Pose pred = st->pose_cur;                                  // k_motion's prediction (a uniform load)
  __syncthreads();
  const int i = i0 + threadIdx.x;
  PvsOut td;
  int nwr = 0;                                                     // doubles of td this point writes
  if (i < n_points) {
    const MapPointDev p = *(const MapPointDev*)(pbuf + (size_t)threadIdx.x * PVS_OUT);
    const size_t gi = (size_t)s * tp.max_points + i;
    int level = -1;
    if (!p.bad) {
      int flags = m.pt_flags[gi];
      CamProj pr;
      const bool projected = td_project(td, flags, p.pos, pred, tp.cam, pr);     // :379-381
      nwr = projected ? 5 : 3;
      if (flags & TDF_IN_IMAGE) {
        nwr = PVS_OUT;
        cam_derivs(tp.cam, pr, td.derivs);                         // :384 GetDerivsUnsafe
        // CalcSearchLevelAndWarpMatrix, jni/PatchFinder.cc:31-68
        const double ooz = 1.0 / td.cam[2];
        double mr[3], md[3];
        pose_rot(pred, p.right, mr);
        pose_rot(pred, p.down, md);
        const double r0 = mr[0] - td.cam[0] * mr[2] * ooz, r1 = mr[1] - td.cam[1] * mr[2] * ooz;
        const double d0 = md[0] - td.cam[0] * md[2] * ooz, d1 = md[1] - td.cam[1] * md[2] * ooz;
        const double* d = td.derivs;
        td.warp_inv[0] = (d[0] * r0 + d[1] * r1) * ooz; td.warp_inv[2] = (d[2] * r0 + d[3] * r1) * ooz;
        td.warp_inv[1] = (d[0] * d0 + d[1] * d1) * ooz; td.warp_inv[3] = (d[2] * d0 + d[3] * d1) * ooz;
        double det = td.warp_inv[0] * td.warp_inv[3] - td.warp_inv[1] * td.warp_inv[2];
        int lv = 0;
        while (det > 3 && lv < NLEV - 1) { lv++; det *= 0.25; }
        if (det > 3 || det < 0.25) flags |= TDF_TMPL_BAD;          // mbTemplateBad = true; return -1
        else { flags &= ~(TDF_SEARCHED | TDF_FOUND); level = lv; } // :389-390
      }
      m.pt_flags[gi] = flags;
    }
    m.pt_level[gi] = level;
  }
  __syncthreads();                                                 // every lane has its map point out of the buffer
  wn[threadIdx.x] = nwr;
  {
    const double* t = (const double*)&td;
    for (int f = 0; f < PVS_OUT; f++) if (f < nwr) pbuf[threadIdx.x * PVS_OUT + f] = t[f];
  }
  __syncthreads();
  double* dst = (double*)(m.td + (size_t)s * tp.max_points + i0);
  for (int k = threadIdx.x; k < nhere * PVS_OUT; k += TRK_THREADS) {
    const int j = k / PVS_OUT, f = k - j * PVS_OUT;
    if (f < wn[j]) dst[(size_t)j * (sizeof(TrackData) / 8) + (f < 9 ? f : f + 3)] = pbuf[k];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// stage 0: PVS lists + coarse selection; stage 1: fine selection (after the coarse pose update).
__global__ __launch_bounds__(TRK_THREADS) void k_plan(MapDev m, TrackParams tp, int stage) {
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  if (!(st->map_good && st->lost_frames < 3)) return;
  const int P = tp.max_points;
  TrackData* td = m.td + (size_t)s * P;
  const MapPointDev* pts = m.pts + (size_t)s * P;
  int* pvs = m.pvs_list + (size_t)s * NLEV * P;
  int2* slist = m.search_list + (size_t)s * P;
  int* ilist = m.iter_list + (size_t)s * P;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int cnt[NLEV];
  __shared__ int wcnt[TRK_THREADS / 64][NLEV];
  __shared__ int plan[8];
  if (stage == 0) {
    if (threadIdx.x < NLEV) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int n = st->n_points;
    // avPVS[l] in map order (:369-392).  Every thread owns PLAN_CH consecutive map points: their levels are fetched in one
    // batch, counted per level in registers, the counts are scanned over the workgroup (one barrier pair per
    // PLAN_CH * TRK_THREADS points instead of three barriers per TRK_THREADS), and the thread appends its points in order.
    constexpr int PLAN_CH = 16;
    for (int base = 0; base < n; base += PLAN_CH * TRK_THREADS) {
      const int first = base + threadIdx.x * PLAN_CH;
      int lv[PLAN_CH], c[NLEV] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < PLAN_CH; k++) { const int i = first + k; lv[k] = m.pt_level[(size_t)s * P + (i < n ? i : n - 1)]; if (i >= n) lv[k] = -1; }
#pragma unroll
      for (int k = 0; k < PLAN_CH; k++)
#pragma unroll
        for (int l = 0; l < NLEV; l++) c[l] += lv[k] == l;
      int off[NLEV];
#pragma unroll
      for (int l = 0; l < NLEV; l++) {                             // exclusive scan of the per-thread counts
        int inc = c[l];
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wcnt[wave][l] = inc;
        off[l] = inc - c[l];
      }
      __syncthreads();
#pragma unroll
      for (int l = 0; l < NLEV; l++) { off[l] += cnt[l]; for (int w = 0; w < wave; w++) off[l] += wcnt[w][l]; }
#pragma unroll
      for (int k = 0; k < PLAN_CH; k++) {
#pragma unroll
        for (int l = 0; l < NLEV; l++) if (lv[k] == l) pvs[l * P + off[l]++] = first + k;
      }
      __syncthreads();
      if (threadIdx.x < NLEV) { int t = cnt[threadIdx.x]; for (int w = 0; w < TRK_THREADS / 64; w++) t += wcnt[w][threadIdx.x]; cnt[threadIdx.x] = t; }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      for (int l = 0; l < NLEV; l++) { st->pvs_count[l] = cnt[l]; st->pvs_head[l] = 0; }
      unsigned nCoarseMax = tp.coarse_max, nCoarseRange = tp.coarse_range;   // :405-432
      st->did_coarse = 0;
      bool bTry = true;
      if (tp.coarse_disabled || st->msd_vel < tp.coarse_min_vel || nCoarseMax == 0) bTry = false;
      if (st->just_recovered) { bTry = true; nCoarseMax *= 2; nCoarseRange *= 2; st->just_recovered = 0; }
      int c3 = 0, c2 = 0;
      const unsigned n3 = cnt[3], n2 = cnt[2];
      if (bTry && n3 + n2 > (unsigned)tp.coarse_min) {              // :437-461
        unsigned take3 = n3 <= nCoarseMax ? n3 : nCoarseMax;
        st->pvs_head[3] = take3;
        c3 = take3;
        if (take3 < nCoarseMax) {
          const unsigned more = nCoarseMax - take3;
          if (n2 <= more) { c3 = 0; c2 = n2; st->pvs_head[2] = n2; }   // :454-456 replaces the L3 selection (PTAM bug kept)
          else { c2 = more; st->pvs_head[2] = more; }
        }
      }
      plan[0] = c3; plan[1] = c2;
      st->coarse_range = nCoarseRange;
      st->n_coarse = c3 + c2; st->n_search = c3 + c2; st->n_iter = c3 + c2;
    }
    __syncthreads();
    const int c3 = plan[0], c2 = plan[1];
    for (int e = threadIdx.x; e < c3 + c2; e += TRK_THREADS) {
      const int idx = e < c3 ? pvs[3 * P + e] : pvs[2 * P + (e - c3)];
      slist[e] = make_int2(idx, tp.coarse_subpix_its);
      ilist[e] = idx;
    }
  } else {
    // fine stage (:493-535)
    __shared__ Pose pose;
    if (threadIdx.x == 0) {
      pose = st->pose_cur;
      const int nit = st->n_iter;                                    // coarse entries already in vIterationSet
      const int h3 = st->pvs_head[3], n3 = st->pvs_count[3] - h3;
      int nother = (st->pvs_count[2] - st->pvs_head[2]) + st->pvs_count[1] + st->pvs_count[0];
      int nFine = tp.max_patches - (nit + n3);                       // :519-521
      if (nFine < 0) nFine = 0;
      if (nother > nFine) nother = nFine;                            // :522-526 (identity shuffle, then chop)
      plan[0] = nit; plan[1] = n3; plan[2] = nother;
      st->fine_range = st->did_coarse ? 5 : 10;                      // :495-497
      st->n_l3 = n3; st->n_search = n3 + nother; st->n_iter = nit + n3 + nother;
    }
    __syncthreads();
    const int nit = plan[0], n3 = plan[1], nother = plan[2];
    const int h3 = st->pvs_head[3], h2 = st->pvs_head[2];
    const int r2 = st->pvs_count[2] - h2, r1 = st->pvs_count[1];
    const int did_coarse = st->did_coarse;
    for (int e = threadIdx.x; e < n3 + nother; e += TRK_THREADS) {
      int idx, its;
      if (e < n3) { idx = pvs[3 * P + h3 + e]; its = tp.fine_subpix_its; }
      else {
        const int k = e - n3;                                        // order: level 2, 1, 0 (:512-514)
        if (k < r2) idx = pvs[2 * P + h2 + k];
        else if (k < r2 + r1) idx = pvs[1 * P + (k - r2)];
        else idx = pvs[0 * P + (k - r2 - r1)];
        its = 0;
      }
      if (e < n3 || did_coarse) td_project_and_derivs(td[idx], m.pt_flags[(size_t)s * P + idx], pts[idx].pos, pose, tp.cam);   // :503-504, :529-532
      slist[e] = make_int2(idx, its);
      ilist[nit + e] = idx;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
DEVFN int wave_sum_i(int v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; }
DEVFN double wave_sum_d(double v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; }

struct SearchArgs {
  const uint8_t* img[NLEV]; size_t img_sstride[NLEV]; int img_pitch[NLEV];   // current frame pyramid
  const uint32_t* corners[NLEV]; const int* rowlut[NLEV]; const int* ncorners;
  int w[NLEV], h[NLEV], cap[NLEV];
  size_t kf_stride[NLEV];       // bytes per keyframe image at level l
  int kf_pitch[NLEV];
  int nblk, S;                  // the launch's workgroups per stream and streams (xcd_stream_block)
};

// ---------------------------------------------------------------------------------------------------------------
// SearchForPoints body (jni/Tracker.cc:629-674).  Several patches per wavefront: G lanes per patch (8 for the 8x8 BASELINE patches -> 8 patches per wave, 16 for the
// reference's 11x11 default -> 4 per wave), lane r < PS owns template row r as packed dwords (unused bytes zero).  Same
// arithmetic per pixel / per candidate as the reference; no LDS, and the dependent global loads of the patches of a wave
// overlap (a one-patch-per-wavefront version of this kernel was 2.4x slower at 8x8 and 2.7x slower at 11x11).  ZMSSD sums use v_dot4_u32_u8.
template <int PS> struct PRow { unsigned w[(PS + 3) / 4]; };
template <int PS> DEVFN PRow<PS> load_row(const uint8_t* p) {
  PRow<PS> r;
#pragma unroll
  for (int k = 0; k < (PS + 3) / 4; k++) r.w[k] = 0u;
  __builtin_memcpy(&r, p, PS);
  return r;
}
template <int PS> DEVFN int row_byte(const PRow<PS>& r, int x) { return (int)((r.w[x >> 2] >> (8 * (x & 3))) & 255u); }
template <int G> DEVFN int grp_sum_i(int v) { for (int d = 1; d < G; d <<= 1) v += __shfl_xor(v, d); return v; }
template <int G> DEVFN double grp_sum_d(double v) { for (int d = 1; d < G; d <<= 1) v += __shfl_xor(v, d); return v; }
DEVFN unsigned udot4(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_udot4(a, b, c, false); }

// Workgroups go to the eight XCDs of the device round-robin by their linear index, and every XCD has its own L2.  The search kernels
// gather from one stream's frame pyramid and keyframes: with (patch block, stream) as (x, y) of the grid the ~140 workgroups of a stream
// were dealt over all eight L2s, each of which fetched the stream's images for itself.  The grid is one-dimensional instead and the linear
// index i is read as (xcd = i % 8, then patch block, then group of eight streams): all workgroups of a stream share i % 8, i.e. one L2.
#define TRK_XCDS 8
DEVFN void xcd_stream_block(int nblk, int S, int& s, int& blk) {
  const int i = blockIdx.x, x = i % TRK_XCDS, j = i / TRK_XCDS;
  blk = j % nblk;
  s = (j / nblk) * TRK_XCDS + x;
  if (s >= S) s = -1;
}
static int xcd_grid(int nblk, int S) { return nblk * ((S + TRK_XCDS - 1) / TRK_XCDS) * TRK_XCDS; }

template <int PS, int G>
__global__ __launch_bounds__(64) void k_searchN(MapDev m, TrackParams tp, SearchArgs a, int stage) {
  constexpr int NPIX = PS * PS, HALF = PS / 2, PPW = 64 / G, NW = (PS + 3) / 4;
  int s, bx;
  xcd_stream_block(a.nblk, a.S, s, bx);
  if (s < 0) return;
  TrackerState* st = &m.st[s];
  if (!(st->map_good && st->lost_frames < 3)) return;
  const int nsearch = st->n_search;
  if (bx * PPW >= nsearch) return;
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  const bool rowact = sub < PS;                                      // this lane owns a template row
  const int e = bx * PPW + grp;
  bool act = e < nsearch;                                            // this lane group has a patch
  const bool lead = sub == 0;
  const int2 ent = act ? m.search_list[(size_t)s * tp.max_points + e] : make_int2(0, 0);
  const int idx = ent.x, nSubPixIts = ent.y;
  const int nRangeL0 = stage == 0 ? st->coarse_range : st->fine_range;
  TrackData& td = m.td[(size_t)s * tp.max_points + idx];
  int& tdlevel = m.pt_level[(size_t)s * tp.max_points + idx];
  int& tdflags = m.pt_flags[(size_t)s * tp.max_points + idx];
  const MapPointDev& p = m.pts[(size_t)s * tp.max_points + idx];
  uint8_t* gtmpl = m.tmpl + ((size_t)s * tp.max_points + idx) * TMPL_PITCH;
  const int level = act ? tdlevel : 0, scale = 1 << level;
  int flags = tdflags;

  // ---- MakeTemplateCoarseCont, jni/PatchFinder.cc:79-125 ----
  double inv[4];
  inv2(td.warp_inv, inv);
  const double m2[4] = {inv[0] * scale, inv[1] * scale, inv[2] * scale, inv[3] * scale};
  bool refresh = !(flags & TDF_HAVE_LAST);
  for (int i = 0; !refresh && i < 2; i++) {
    const double dx = m2[i] - td.last_warp[i], dy = m2[2 + i] - td.last_warp[2 + i];
    if (dx * dx + dy * dy > 0.07 * 0.07) refresh = true;
  }
  refresh = refresh && act;
  PRow<PS> trow;                                                     // template row `sub`
#pragma unroll
  for (int k = 0; k < NW; k++) trow.w[k] = 0u;
  if (rowact) trow = load_row<PS>(gtmpl + sub * PS);                 // the cached row, fetched along with the tracker data; a refresh overwrites it
  const int tsum_cached = td.tsum, tsumsq_cached = td.tsumsq;
  int tsum, tsumsq;
  if (__any(refresh)) {
    // transform_image (jni/vision/ImageHandler.cpp:21-113): same accumulated stepping of the sample position
    const int sl = p.src_level;
    const uint8_t* src = m.kf_img[sl] + ((size_t)s * tp.max_keyframes + p.src_kf) * a.kf_stride[sl];
    const int sp = a.kf_pitch[sl], iw = a.w[sl], ih = a.h[sl];
    const double across[2] = {m2[0], m2[2]}, down[2] = {m2[1], m2[3]};
    double px = (double)p.irx - (m2[0] * HALF + m2[1] * HALF), py = (double)p.iry - (m2[2] * HALF + m2[3] * HALF);
    const double cr[2] = {down[0] - PS * across[0], down[1] - PS * across[1]};
    // every lane walks the accumulated sample position through the rows above its own (PS steps + the carriage return
    // each, exactly the additions transform_image makes), then samples its row while stepping along it
#pragma unroll 1
    for (int i = 0; i < PS - 1; i++) {
      if (i < sub) {
#pragma unroll
        for (int j = 0; j < PS; j++) { px += across[0]; py += across[1]; }
        px += cr[0]; py += cr[1];
      }
    }
    if (refresh) {
      int nOutside = 0, sum = 0, sumsq = 0;
      const float x_bound = (float)(iw - 1), y_bound = (float)(ih - 1);
      if (rowact) {
#pragma unroll
      for (int k = 0; k < NW; k++) trow.w[k] = 0u;
#pragma unroll
      for (int j = 0; j < PS; j++) {
        double x = px, y = py;
        px += across[0]; py += across[1];
        int v = 0;
        if (0 <= x && 0 <= y && x < x_bound && y < y_bound) {
          const int lx = (int)x, ly = (int)y;                        // sample(), ImageHandler.cpp:12-19
          x -= lx; y -= ly;
          const uint8_t* q0 = src + (size_t)ly * sp + lx;
          v = (uint8_t)((1 - y) * ((1 - x) * q0[0] + x * q0[1]) + y * ((1 - x) * q0[sp] + x * q0[sp + 1]));
        } else nOutside++;
        trow.w[j >> 2] |= (unsigned)v << (8 * (j & 3));
        sum += v; sumsq += v * v;
      }
      __builtin_memcpy(gtmpl + sub * PS, &trow, PS);
      }
      nOutside = grp_sum_i<G>(nOutside);
      tsum = grp_sum_i<G>(sum); tsumsq = grp_sum_i<G>(sumsq);        // MakeTemplateSums :152-164
      flags = nOutside ? (flags | TDF_TMPL_BAD) : (flags & ~TDF_TMPL_BAD);
      flags |= TDF_HAVE_LAST;
      if (lead) { td.tsum = tsum; td.tsumsq = tsumsq; for (int i = 0; i < 4; i++) td.last_warp[i] = m2[i]; }
    }
  }
  if (!refresh) { tsum = tsum_cached; tsumsq = tsumsq_cached; }
  if (act && (flags & TDF_TMPL_BAD)) {                               // jni/Tracker.cc:637-640
    if (lead) tdflags = flags & ~(TDF_IN_IMAGE | TDF_FOUND);
    act = false;
  }
  for (int l = 0; l < NLEV; l++) {                                   // manMeasAttempted[level]++ (:641), one atomic per wave
    const int c = __popcll(__ballot(act && lead && level == l));
    if (lane == 0 && c) atomicAdd(&st->attempted[l], c);
  }

  // ---- FindPatchCoarse, jni/PatchFinder.cc:170-235 ----
  const double irx = td.image[0] / scale, iry = td.image[1] / scale;
  const unsigned nRange = ((unsigned)nRangeL0 + scale - 1) / scale;
  int nTop = (int)(iry - nRange);
  const int nBottomPlusOne = (int)(iry + nRange + 1);
  const int nLeft = (int)(irx - nRange), nRight = (int)(irx + nRange);
  const int rows = a.h[level], cols = a.w[level];
  if (nTop < 0) nTop = 0;
  int nBestSSD = tp.max_ssd + 1;
  uint32_t bestCorner = 0;                                           // packed position of the best candidate so far
  unsigned nEval = 0;
  const uint32_t* corners = a.corners[level] + (size_t)s * a.cap[level];
  const uint8_t* img = a.img[level] + (size_t)s * a.img_sstride[level];
  const int ip = a.img_pitch[level];
  int i0 = 0, i1 = 0;
  if (act && !(nTop >= rows) && !(nBottomPlusOne <= 0)) {
    const int* lut = a.rowlut[level] + (size_t)s * (rows + 1);
    i0 = lut[nTop];
    i1 = nBottomPlusOne >= rows ? a.ncorners[s * NLEV + level] : lut[nBottomPlusOne];
  }
  const double r2max = (double)(nRange * nRange);
  // Filter 4 G corners of the row-LUT window per step (:216-219: x window, then the circular range test): lane `sub` takes
  // corners base + j G + sub, j < 4, so the group's survivors form one bit mask in raster order.  Survivors are scored two
  // at a time (both image rows are in flight before either is used) and compared in raster order (:223).
  constexpr int CPL = 4;
  for (int base = i0; __any(base < i1); base += CPL * G) {
    uint32_t cval[CPL];
    unsigned long long gm = 0;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
      const int ci = base + j * G + sub;
      cval[j] = ci < i1 ? corners[ci] : 0u;
    }
#pragma unroll
    for (int j = 0; j < CPL; j++) {
      const int ci = base + j * G + sub;
      bool ok = false;
      if (ci < i1) {
        const int cx = cval[j] & 0xFFFF, cy = cval[j] >> 16;
        if (!(cx < nLeft || cx > nRight)) {
          const double dx = irx - cx, dy = iry - cy;
          ok = !(dx * dx + dy * dy > r2max);
        }
      }
      gm |= (unsigned long long)((unsigned)(__ballot(ok) >> (grp * G)) & ((1u << G) - 1u)) << (j * G);
    }
    nEval += __popcll(gm);
    while (__any(gm != 0)) {
      int kk[2]; bool has[2], inside[2]; uint32_t c[2]; PRow<PS> n[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        has[u] = gm != 0;
        kk[u] = has[u] ? __ffsll((long long)gm) - 1 : 0;
        gm &= gm - 1;
        const int j = kk[u] / G;
        const uint32_t csel = j == 0 ? cval[0] : (j == 1 ? cval[1] : (j == 2 ? cval[2] : cval[3]));
        c[u] = __shfl(csel, grp * G + (kk[u] % G));
        const int cx = c[u] & 0xFFFF, cy = c[u] >> 16;
        inside[u] = has[u] && cx >= HALF && cy >= HALF && cx < cols - HALF && cy < rows - HALF;   // in_image_with_border
#pragma unroll
        for (int k = 0; k < NW; k++) n[u].w[k] = 0u;
        if (inside[u] && rowact) n[u] = load_row<PS>(img + (size_t)(cy - HALF + sub) * ip + (cx - HALF));
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        // ZMSSDAtPoint (:352-380): one image row per lane; the pad bytes of both rows are zero
        unsigned sA = 0, sQ = 0, sX = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
          sA = udot4(n[u].w[k], 0x01010101u, sA);
          sQ = udot4(n[u].w[k], n[u].w[k], sQ);
          sX = udot4(n[u].w[k], trow.w[k], sX);
        }
        sA = grp_sum_i<G>(sA); sQ = grp_sum_i<G>(sQ); sX = grp_sum_i<G>(sX);
        if (has[u]) {
          int ssd = tp.max_ssd + 1;
          if (inside[u]) {
            const int SA = tsum, SB = (int)sA;
            ssd = ((2 * SA * SB - SA * SA - SB * SB) / NPIX + (int)sQ + tsumsq - 2 * (int)sX);
          }
          if (ssd < nBestSSD) { nBestSSD = ssd; bestCorner = c[u]; }   // first strict minimum in raster order (:223)
        }
      }
    }
  }
  flags |= TDF_SEARCHED;                                             // :645
  {
    const int tot = wave_sum_i(act && lead ? (int)nEval : 0);
    if (lane == 0 && tot) atomicAdd(&st->n_zmssd, (unsigned long long)tot);
  }
  bool found = act && nBestSSD < tp.max_ssd;
  if (act && !found && lead) tdflags = flags & ~TDF_FOUND;          // :646-649
  const uint32_t bc = found ? bestCorner : 0u;
  const double coarse[2] = {level_zero_pos((double)(bc & 0xFFFF), level), level_zero_pos((double)(bc >> 16), level)};
  if (found) flags |= TDF_FOUND;
  const bool dosub = found && nSubPixIts > 0;                        // refined by k_subpixN, which also counts it as found
  if (found) {                                                       // :668-671
    flags = dosub ? (flags | TDF_SUBPIX) : (flags & ~TDF_SUBPIX);
    if (lead) { tdflags = flags; td.sqrt_inv_noise = 1.0 / scale; td.vfound[0] = coarse[0]; td.vfound[1] = coarse[1]; }
  }
  for (int l = 0; l < NLEV; l++) {                                   // manMeasFound[level]++ (:652)
    const int c = __popcll(__ballot(found && !dosub && lead && level == l));
    if (lane == 0 && c) atomicAdd(&st->found[l], c);
  }
}

// MakeSubPixTemplate (jni/PatchFinder.cc:242-271) + IterateSubPixToConvergence (:273-350) for the patches k_searchN found
// with a sub-pixel budget (level-3 points in the fine stage, every point in the coarse stage).  A kernel of its own so
// that its registers (gradients, 3x3 inverse) do not halve the occupancy of the search proper.  Eight patches per
// wavefront, lane y owns interior row y of the template.
template <int PS, int G>
DEVFN void subpix_block(const MapDev& m, const TrackParams& tp, const SearchArgs& a, TrackerState* st, int s, int nsub, int blk) {
  constexpr int HALF = PS / 2, PPW = 64 / G, NW = (PS + 3) / 4, Q = PS - 2;
  const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
  const int e = blk * PPW + grp;
  const bool lead = sub == 0;
  const int2 ent = e < nsub ? m.search_list[(size_t)s * tp.max_points + e] : make_int2(0, 0);
  const int idx = ent.x, nSubPixIts = ent.y;
  TrackData& td = m.td[(size_t)s * tp.max_points + idx];
  int& tdlevel = m.pt_level[(size_t)s * tp.max_points + idx];
  int& tdflags = m.pt_flags[(size_t)s * tp.max_points + idx];
  int flags = tdflags;
  const bool dosub = e < nsub && nSubPixIts > 0 && (flags & TDF_FOUND) && (flags & TDF_SUBPIX);
  if (!__any(dosub)) return;
  const int level = dosub ? tdlevel : 0, scale = 1 << level;
  const int rows = a.h[level], cols = a.w[level];
  const uint8_t* img = a.img[level] + (size_t)s * a.img_sstride[level];
  const int ip = a.img_pitch[level];
  PRow<PS> trow, rup, rdn;
#pragma unroll
  for (int k = 0; k < NW; k++) trow.w[k] = 0u;
  if (sub < PS) trow = load_row<PS>(m.tmpl + ((size_t)s * tp.max_points + idx) * TMPL_PITCH + sub * PS);
  const bool rowok = sub >= 1 && sub <= Q;
#pragma unroll
  for (int k = 0; k < NW; k++) { rup.w[k] = __shfl(trow.w[k], lane - 1); rdn.w[k] = __shfl(trow.w[k], lane + 1); }
  double gx[Q], gy[Q];
  double h00 = 0, h01 = 0, h02 = 0, h11 = 0, h12 = 0, h22 = 0;
#pragma unroll
  for (int x = 1; x <= Q; x++) {
    gx[x - 1] = 0; gy[x - 1] = 0;
    if (rowok) {
      gx[x - 1] = 0.5 * (row_byte(trow, x + 1) - row_byte(trow, x - 1));
      gy[x - 1] = 0.5 * (row_byte(rdn, x) - row_byte(rup, x));
      h00 += gx[x - 1] * gx[x - 1]; h01 += gx[x - 1] * gy[x - 1]; h02 += gx[x - 1];
      h11 += gy[x - 1] * gy[x - 1]; h12 += gy[x - 1]; h22 += 1.0;
    }
  }
  h00 = grp_sum_d<G>(h00); h01 = grp_sum_d<G>(h01); h02 = grp_sum_d<G>(h02);     // quarter-integers: exact in any order
  h11 = grp_sum_d<G>(h11); h12 = grp_sum_d<G>(h12); h22 = grp_sum_d<G>(h22);
  const double H[9] = {h00, h01, h02, h01, h11, h12, h02, h12, h22};
  double Hinv[9];
  inv3(H, Hinv);
  double sub0 = td.vfound[0], sub1 = td.vfound[1], meanDiff = 0.0;      // the coarse position k_searchN left
  bool running = dosub, converged = false;
  for (int it = 0; __any(running); it++) {
    if (it >= nSubPixIts) running = false;
    double cx = 0, cy = 0;
    if (running) {
      cx = level_n_pos(sub0, level); cy = level_n_pos(sub1, level);
      const int xb = (int)(cx > 0.0 ? cx + 0.5 : cx - 0.5), yb = (int)(cy > 0.0 ? cy + 0.5 : cy - 0.5);
      const int b = HALF + 1;
      if (!(xb >= b && yb >= b && xb < cols - b && yb < rows - b)) running = false;   // went off edge -> fail
    }
    // v3Accum over the interior pixels in the reference's order, row by row (:316-340): the lane of row y adds its Q pixels
    // to the running sums it takes over from the lane of row y - 1 (a tree over the rows would round differently)
    double p0[Q], p1[Q], p2[Q];
#pragma unroll
    for (int x = 0; x < Q; x++) { p0[x] = 0; p1[x] = 0; p2[x] = 0; }
    if (running && rowok) {
      const double bx = cx - HALF, by = cy - HALF;
      const double dX = bx - floor(bx), dY = by - floor(by);
      const float fTL = (float)((1.0 - dX) * (1.0 - dY)), fTR = (float)((dX) * (1.0 - dY));
      const float fBL = (float)((1.0 - dX) * (dY)), fBR = (float)((dX) * (dY));
      const uint8_t* r0 = img + (size_t)((int)by + sub) * ip + (int)bx + 1;
      const PRow<PS> q0 = load_row<PS>(r0), q1 = load_row<PS>(r0 + ip);
#pragma unroll
      for (int x = 1; x <= Q; x++) {
        const float fPixel = fTL * row_byte(q0, x - 1) + fTR * row_byte(q0, x) + fBL * row_byte(q1, x - 1) + fBR * row_byte(q1, x);
        const double dDiff = (fPixel - (float)row_byte(trow, x)) + meanDiff;
        p0[x - 1] = dDiff * gx[x - 1]; p1[x - 1] = dDiff * gy[x - 1]; p2[x - 1] = dDiff;
      }
    }
    double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll 1
    for (int y = 1; y <= Q; y++) {
      if (sub == y) {
#pragma unroll
        for (int x = 0; x < Q; x++) { a0 += p0[x]; a1 += p1[x]; a2 += p2[x]; }
      }
      const int src = grp * G + y;
      a0 = __shfl(a0, src); a1 = __shfl(a1, src); a2 = __shfl(a2, src);
    }
    if (running) {
      const double u0 = Hinv[0] * a0 + Hinv[1] * a1 + Hinv[2] * a2;
      const double u1 = Hinv[3] * a0 + Hinv[4] * a1 + Hinv[5] * a2;
      const double u2 = Hinv[6] * a0 + Hinv[7] * a1 + Hinv[8] * a2;
      sub0 -= u0 * scale; sub1 -= u1 * scale; meanDiff -= u2;
      if (u0 * u0 + u1 * u1 < 0.03 * 0.03) { converged = true; running = false; }
    }
  }
  if (dosub && lead) {
    if (!converged) tdflags = flags & ~TDF_FOUND;                   // :658-666 un-finds the point
    else { td.vfound[0] = sub0; td.vfound[1] = sub1; }
  }
  for (int l = 0; l < NLEV; l++) {                                   // manMeasFound[level]++ (:652)
    const int c = __popcll(__ballot(dosub && converged && lead && level == l));
    if (lane == 0 && c) atomicAdd(&st->found[l], c);
  }
}

#define SUBPIX_GRID 16     // the entries with a sub-pixel budget are few (level-3 points / the coarse set): a short grid that strides
template <int PS, int G>
__global__ __launch_bounds__(64) void k_subpixN(MapDev m, TrackParams tp, SearchArgs a, int stage) {
  int s, bx;
  xcd_stream_block(a.nblk, a.S, s, bx);
  if (s < 0) return;
  TrackerState* st = &m.st[s];
  if (!(st->map_good && st->lost_frames < 3)) return;
  const int nsub = stage == 0 ? st->n_search : st->n_l3;             // entries that carry a sub-pixel budget
  for (int blk = bx; blk * (64 / G) < nsub; blk += a.nblk) subpix_block<PS, G>(m, tp, a, st, s, nsub, blk);
}

// ---------------------------------------------------------------------------------------------------------------
// k_pose: one workgroup per stream.  The tracker data of the iteration set is gathered once into a component-major
// working set indexed by iteration-set entry (coalesced, L2 resident) and scattered back after the ten iterations;
// entry e is always handled by thread e % POSE_THREADS, so the working set needs no synchronisation.
#define POSE_THREADS 128
// Diagnostic build only (-DVSLAM_BA_PROF, tools/build_baprof.sh): clock64() stamps of block 0 / thread 0 per phase of k_pose.
#ifdef VSLAM_BA_PROF
__device__ unsigned long long g_pose_prof[16];
#define POSE_STAMP(id) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = clock64(); g_pose_prof[id] += t_ - g_pose_prof[15]; g_pose_prof[15] = t_; } } while (0)
#else
#define POSE_STAMP(id) do { } while (0)
#endif
#define POSE_WAVES (POSE_THREADS / 64)
#ifndef POSE_ILP
#define POSE_ILP 2
#endif

struct PoseItem {            // TrackerData fields used by the pose iterations (jni/TrackerData.h:36-66)
  double cam[3], image[2], derivs[4], vfound[2], sqrt_inv_noise;   // the 2x6 Jacobian is re-derived from cam + derivs where used
  int flags, idx;
};
struct PoseWs { double* d; int* i; int P; };

DEVFN void item_load(PoseItem& it, const PoseWs& w, int e) {
  it.flags = w.i[e]; it.idx = w.i[w.P + e];
#pragma unroll
  for (int i = 0; i < 3; i++) it.cam[i] = w.d[(0 + i) * w.P + e];
#pragma unroll
  for (int i = 0; i < 2; i++) { it.image[i] = w.d[(3 + i) * w.P + e]; it.vfound[i] = w.d[(9 + i) * w.P + e]; }
#pragma unroll
  for (int i = 0; i < 4; i++) it.derivs[i] = w.d[(5 + i) * w.P + e];
  it.sqrt_inv_noise = w.d[11 * w.P + e];
}
DEVFN void item_store(const PoseItem& it, const PoseWs& w, int e, bool all) {   // all: cam/derivs/flags changed too
#pragma unroll
  for (int i = 0; i < 2; i++) w.d[(3 + i) * w.P + e] = it.image[i];
  if (!all) return;
  w.i[e] = it.flags;
#pragma unroll
  for (int i = 0; i < 3; i++) w.d[(0 + i) * w.P + e] = it.cam[i];
#pragma unroll
  for (int i = 0; i < 4; i++) w.d[(5 + i) * w.P + e] = it.derivs[i];
}

// CalcPoseUpdate, jni/Tracker.cc:683-774 (Tukey).  All threads of the workgroup call it; result in up[6] (LDS).
// The working set holds the nf FOUND entries of the iteration set in iteration-set order (k_pose compacts them once).
//
// The weighted normal equations are accumulated in the reference's ORDER: wls.add_mJ row 0, row 1 of every measurement in
// turn (:736-768, jni/myWLS.h:39-50), every one of the 27 sums (21 of the upper triangle of C, 6 of v) a chain of
// 2 nf dependent additions that starts from add_prior's value (:734).  fp addition does not associate, and PTAM's
// templates are trunc(bilinear): a pose that differs in the last bit flips template pixels on flat image regions, so a
// tree reduction would not do.  The workgroup is two wavefronts: wavefront 1 PRODUCES -- one lane per add_mJ call (two
// lanes per measurement): weight, its row of the Jacobian and the call's 27 products (w J[r]) * J[c] and m * (w J[k]) --
// a chunk ahead into LDS (the median's sort buffer is free by then); lane q of wavefront 0 walks sum q through the
// chunk: one LDS read per two calls (immediate offsets), two groups of eight calls read ahead, and one dependent addition
// per call (8 cycles on gfx950, tools/probes/dep_add.hip).  Four such workgroups share a CU (one chain per SIMD).  A single
// wavefront issues an fp64 instruction every 8 cycles at best, a SIMD one every ~4.3 from two wavefronts (same probe): the
// producer's ~130 fp64 instructions per chunk of 64 calls are the pace (1650 cycles per chunk measured, 23 per call).
#define POSE_CHUNK 32                 // measurements per chunk: two producer lanes each
#define POSE_ENT 27                   // doubles per add_mJ call in LDS: its 27 products (odd stride: bank spread of the producer's stores)
#define POSE_GRP 8                    // add_mJ calls per register set of the chain (three sets in rotation)
static_assert(2 * 2 * POSE_CHUNK * POSE_ENT + 2 * POSE_GRP * POSE_ENT + 64 <= SORT_CAP, "two record buffers (and the chain's two groups of read-ahead) in the sort buffer");
static_assert(POSE_THREADS == 128 && POSE_CHUNK == 32 && POSE_GRP % 2 == 0, "wavefront 0 chains, wavefront 1 produces");

DEVFN void pose_chain_load(double (&v)[POSE_GRP], const double __attribute__((address_space(3)))* p) {
#pragma unroll
  for (int u = 0; u < POSE_GRP; u++) v[u] = p[u * POSE_ENT];
}
DEVFN void pose_chain_add(double& acc, const double (&v)[POSE_GRP]) {
#pragma unroll
  for (int u = 0; u < POSE_GRP; u++) acc += v[u];
}

#ifdef VSLAM_BA_PROF   // the chain loop's two wavefronts apart: slots 10 / 11 = wavefront 0 working / at the barrier, 12 / 13 = wavefront 1
#define POSE_ACC_PROF_DECL unsigned long long tw_ = 0, tb_ = 0, tl_ = clock64()
#define POSE_ACC_BARRIER() do { const unsigned long long a_ = clock64(); tw_ += a_ - tl_; __syncthreads(); tl_ = clock64(); tb_ += tl_ - a_; } while (0)
#define POSE_ACC_PROF_END() do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { g_pose_prof[10 + 2 * (threadIdx.x >> 6)] += tw_; g_pose_prof[11 + 2 * (threadIdx.x >> 6)] += tb_; } } while (0)
#else
#define POSE_ACC_PROF_DECL do { } while (0)
#define POSE_ACC_BARRIER() __syncthreads()
#define POSE_ACC_PROF_END() do { } while (0)
#endif
#define PAS1 __attribute__((address_space(1)))
#define PAS3 __attribute__((address_space(3)))
// The accumulation itself is its own function (not inlined): it gets its own register allocation -- three register sets of
// tracker data in flight, three of chain operands -- instead of sharing k_pose's, which is full.
// Each wavefront runs its OWN loop (the branch is scalar: readfirstlane), with the same count of workgroup barriers on
// both sides: in one shared loop the compiler's wait-count bookkeeping merges the two roles at every join and ends up
// waiting for ALL outstanding loads before the producer's first use, a global-memory round trip in every chunk.
// The producer's tracker data comes from the working set in global memory THREE chunks ahead (three register sets in
// rotation, the loop unrolled by three so that no set is copied): a load has two chunk periods to land.
// (The outlier marks of the last iteration, :749-756, are a pass of their own in calc_pose_update: a global STORE anywhere in this
// loop makes loads and stores share the wait counter, and the compiler then waits for zero before every use.)
__device__ __attribute__((noinline)) double pose_accumulate(const double PAS1* wd, int P, int nf, double wls_prior, bool qint, double sigma2, double PAS3* sortbuf) {
  const int lane = threadIdx.x & 63;
  nf = __builtin_amdgcn_readfirstlane(nf); P = __builtin_amdgcn_readfirstlane(P);   // (arguments arrive in vector registers: make the loops scalar again)
  const int nchunks = (nf + POSE_CHUNK - 1) / POSE_CHUNK;
  constexpr int BUF = 2 * POSE_CHUNK * POSE_ENT;
  // sum q of the chain: upper triangle of C row by row (0..20), then v (21..26); the diagonal starts at the prior
  POSE_ACC_PROF_DECL;
  double acc = 0.0;
  if (lane == 0 || lane == 6 || lane == 11 || lane == 15 || lane == 18 || lane == 20) acc = 0.0 + wls_prior;   // add_prior, :734
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 1) {
    // producer lane = (measurement, row): the products of one add_mJ call; a measurement past the end, or one the reference
    // skips (weight 0), is a record of zeros: x + 0 = x
    const int row = lane & 1, pm = lane >> 1;
    struct Item { double cam[3], image[2], vfound[2], da, db, sqrt_inv_noise; };
    // The loads are issued by hand and waited for by hand (POSE_ITEM_LOADS per set, in-order return): the compiler's own
    // bookkeeping puts a wait for ALL outstanding loads in front of the first use of the oldest set.
#define POSE_ITEM_LOADS 10
    auto gld = [&](double& dst, int comp, int e) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(wd + (size_t)comp * P + e)); };
    auto load_chunk = [&](Item& t, int k) {
      int e = k * POSE_CHUNK + pm;
      e = e < nf ? e : nf - 1;
#pragma unroll
      for (int i = 0; i < 3; i++) gld(t.cam[i], i, e);
#pragma unroll
      for (int i = 0; i < 2; i++) { gld(t.image[i], 3 + i, e); gld(t.vfound[i], 9 + i, e); }
      gld(t.da, 5 + 2 * row, e); gld(t.db, 6 + 2 * row, e);         // this row of m2CamDerivs
      gld(t.sqrt_inv_noise, 11, e);
    };
    // the set is usable once at most `newer` later loads are outstanding; its registers are operands so that no use moves above the wait
#define POSE_WAIT_SET(t, N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(t.cam[0]), "+v"(t.cam[1]), "+v"(t.cam[2]), "+v"(t.image[0]), "+v"(t.image[1]), \
                                         "+v"(t.vfound[0]), "+v"(t.vfound[1]), "+v"(t.da), "+v"(t.db), "+v"(t.sqrt_inv_noise))
    auto produce = [&](const Item& t, int e, double PAS3* buf) {
      double PAS3* dst = buf + lane * POSE_ENT;
      double w = 0.0, err[2] = {0.0, 0.0};
      if (e < nf) {
        err[0] = (t.vfound[0] - t.image[0]) * t.sqrt_inv_noise; err[1] = (t.vfound[1] - t.image[1]) * t.sqrt_inv_noise;
        const double es = err[0] * err[0] + err[1] * err[1];
        w = tukey_weight(es, sigma2);
      }
      if (w == 0.0) {
#pragma unroll
        for (int k = 0; k < POSE_ENT; k++) dst[k] = 0.0;
        return;
      }
      // this row of CalcJacobian (jni/TrackerData.h:107-122), then wls.add_mJ(v2(row), sqrt_inv_noise * J.row(row), w), :760-767
      const double ooz = 1.0 / t.cam[2];
      const double c[3] = {t.cam[0], t.cam[1], t.cam[2]};
      double J[6], wJ[6];
#pragma unroll
      for (int m = 0; m < 6; m++) {
        double f0, f1;
        se3_generator_motion(m, c, ooz, f0, f1);
        J[m] = t.sqrt_inv_noise * (t.da * f0 + t.db * f1);
        wJ[m] = w * J[m];
      }
      const double e_ = row ? err[1] : err[0];
      const double mm = qint ? (double)(int)e_ : e_;
      int q = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int cc = r; cc < 6; cc++) dst[q++] = wJ[r] * J[cc];
#pragma unroll
      for (int k = 0; k < 6; k++) dst[21 + k] = mm * wJ[k];
    };
    Item it0, it1, it2;
    load_chunk(it0, 0); load_chunk(it1, 1); load_chunk(it2, 2);
    POSE_ACC_BARRIER();                                                // the median has left the sort buffer
    POSE_WAIT_SET(it0, 20);
    produce(it0, pm, sortbuf);
    POSE_ACC_BARRIER();
    auto prod_step = [&](int k, Item& tprod, Item& tload) {         // chunk k + 1 while the chain walks chunk k; then the load of chunk k + 3
      if (k + 1 < nchunks) {
        load_chunk(tload, k + 3);                                   // (set k mod 3: chunk k was produced a step ago)
        POSE_WAIT_SET(tprod, 20);                                   // (two newer sets in flight: 2 * POSE_ITEM_LOADS)
        produce(tprod, (k + 1) * POSE_CHUNK + pm, sortbuf + ((k + 1) & 1) * BUF);
      }
      POSE_ACC_BARRIER();
    };
    for (int k = 0; k < nchunks; k += 3) {                          // chunk j's data is in set j mod 3
      prod_step(k, it1, it0);
      if (k + 1 < nchunks) prod_step(k + 1, it2, it1);
      if (k + 2 < nchunks) prod_step(k + 2, it0, it2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the clamped loads past the last chunk
  } else {
    POSE_ACC_BARRIER();
    POSE_ACC_BARRIER();
    for (int k = 0; k < nchunks; k++) {
      const int left = nf - k * POSE_CHUNK;
      const int ng = (2 * (left < POSE_CHUNK ? left : POSE_CHUNK) + POSE_GRP - 1) / POSE_GRP;    // groups of calls in this chunk (the tail is zero records)
      constexpr int GSTR = POSE_GRP * POSE_ENT;
      const double PAS3* rp = sortbuf + (k & 1) * BUF + lane;
      // three register sets in rotation, two groups of read-ahead: an LDS read has sixteen additions to land (the
      // scheduling barriers keep the compiler from sinking the reads down to their use, which exposes the latency)
      double x0[POSE_GRP], x1[POSE_GRP], x2[POSE_GRP];
      pose_chain_load(x0, rp);
      pose_chain_load(x1, rp + GSTR);
      for (int g = 0; g < ng; g += 3) {                              // the read-ahead past the last group stays inside the sort buffer
        pose_chain_load(x2, rp + (g + 2) * GSTR);
        __builtin_amdgcn_sched_barrier(0);
        pose_chain_add(acc, x0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 >= ng) break;
        pose_chain_load(x0, rp + (g + 3) * GSTR);
        __builtin_amdgcn_sched_barrier(0);
        pose_chain_add(acc, x1);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 >= ng) break;
        pose_chain_load(x1, rp + (g + 4) * GSTR);
        __builtin_amdgcn_sched_barrier(0);
        pose_chain_add(acc, x2);
        __builtin_amdgcn_sched_barrier(0);
      }
      POSE_ACC_BARRIER();
    }
  }
  POSE_ACC_PROF_END();
  return acc;
}

DEVFN void calc_pose_update(const PoseWs& ws, MapPointDev* pts, int nf, const TrackParams& tp,
                            double dOverrideSigma, bool bMarkOutliers, double* sortbuf, double* red /* [28] */,
                            double* up /* [6] */, int* hist /* [768] */, unsigned long long* sel /* [1] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (nf == 0) {                                                    // :712-716
    if (threadIdx.x < 6) up[threadIdx.x] = 0.0;
    __syncthreads();
    return;
  }
  POSE_STAMP(2);
  double sigma2;
  if (dOverrideSigma > 0) sigma2 = dOverrideSigma;                  // :720-721
  else {                                                            // Tukey::FindSigmaSquared, jni/MEstimator.h:67-77
    const double med = block_radix_select(sortbuf, nf, nf / 2, hist, sel);   // same order statistic as sort + [n/2]
    sigma2 = tukey_sigma_squared(med, (unsigned long)nf);
  }
  POSE_STAMP(3);
  const double acc = pose_accumulate((const double PAS1*)ws.d, ws.P, nf, tp.wls_prior, (tp.quirks & VSLAM_Q_POSE_INT_RESIDUAL) != 0, sigma2, (double PAS3*)sortbuf);
  if (bMarkOutliers) {                                              // :749-756: a measurement the M-estimator gave no weight counts against its point
    for (int e = threadIdx.x; e < nf; e += POSE_THREADS) {
      const double sn = ws.d[11 * ws.P + e];
      const double e0 = (ws.d[9 * ws.P + e] - ws.d[3 * ws.P + e]) * sn, e1 = (ws.d[10 * ws.P + e] - ws.d[4 * ws.P + e]) * sn;
      MapPointDev* mp = pts + ws.i[ws.P + e];
      if (tukey_weight(e0 * e0 + e1 * e1, sigma2) == 0.0) mp->n_out++; else mp->n_in++;
    }
  }
  POSE_STAMP(4);
  if (wave == 0 && lane < 27) red[lane] = acc;
  __syncthreads();
  POSE_STAMP(5);
  if (threadIdx.x == 0) {
    double C[36], v[6];
    int q = 0;
    for (int r = 0; r < 6; r++)
      for (int c = r; c < 6; c++) { const double x = red[q++]; C[r * 6 + c] = x; C[c * 6 + r] = x; }   // myWLS::compute mirrors the triangle, :54-58
    for (int r = 0; r < 6; r++) v[r] = red[21 + r];
    if (!lu_solve6(C, v)) for (int r = 0; r < 6; r++) v[r] = 0.0;
    for (int r = 0; r < 6; r++) up[r] = v[r];
  }
  __syncthreads();
  POSE_STAMP(6);
}

// KeyFrameLinearDist, jni/MapMaker.cc:705-712
DEVFN double kf_linear_dist(const Pose& a, const Pose& b) {
  const Pose ia = pose_inverse(a), ib = pose_inverse(b);
  const double d0 = ib.t[0] - ia.t[0], d1 = ib.t[1] - ia.t[1], d2 = ib.t[2] - ia.t[2];
  return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}

// stage 0: coarse GN iterations (:463-490); stage 1: fine GN iterations + end of TrackMap/TrackFrame.
#ifndef VSLAM_POSE_WAVES
#define VSLAM_POSE_WAVES 2
#endif
__global__ __launch_bounds__(POSE_THREADS) __attribute__((amdgpu_waves_per_eu(VSLAM_POSE_WAVES, VSLAM_POSE_WAVES))) void k_pose(MapDev m, TrackParams tp, int stage) {
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  if (!(st->map_good && st->lost_frames < 3)) return;
  const int P = tp.max_points;
  TrackData* td = m.td + (size_t)s * P;
  MapPointDev* pts = m.pts + (size_t)s * P;
  const int* ilist = m.iter_list + (size_t)s * P;
  __shared__ double sortbuf[SORT_CAP];
  __shared__ double red[28];
  __shared__ double up[6], last_up[6];
  __shared__ int gcnt[4 * POSE_WAVES];
  __shared__ int hist[768];
  __shared__ unsigned long long sel[1];
  __shared__ Pose pose;
  if (threadIdx.x == 0) pose = st->pose_cur;
  if (threadIdx.x < 6) last_up[threadIdx.x] = 0.0;
  const int n = stage == 0 ? st->n_coarse : st->n_iter;
  if (stage == 0) {
    if (n == 0) return;
    const int nFound = st->found[0] + st->found[1] + st->found[2] + st->found[3];
    if (nFound < tp.coarse_min) return;                              // :465
  }
#ifdef VSLAM_BA_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0) g_pose_prof[15] = clock64();
#endif
  const PoseWs ws = {m.pose_ws + (size_t)s * POSE_WS_COMPS * P, m.pose_wsi + (size_t)s * 2 * P, P};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Gather: the FOUND entries of the iteration set, in iteration-set order, into the working set (every loop of the pose
  // iterations skips the others: TrackerData::bFound is fixed for the stage).  Ordered compaction by ballots, four blocks
  // of POSE_THREADS entries per barrier pair; the entry indices, the flags, then every operand of the batch are loaded
  // before anything is stored, so the dependent round trips (list -> flags -> tracker data) are paid once per batch.
  constexpr int POSE_GB = 4;
  int nf = 0;
  for (int e0 = 0; e0 < n; e0 += POSE_GB * POSE_THREADS) {
    int idx[POSE_GB], fl[POSE_GB]; bool fnd[POSE_GB]; unsigned long long bm[POSE_GB]; double v[POSE_GB][12];
#pragma unroll
    for (int u = 0; u < POSE_GB; u++) { const int e = e0 + u * POSE_THREADS + threadIdx.x; idx[u] = ilist[e < n ? e : n - 1]; }
#pragma unroll
    for (int u = 0; u < POSE_GB; u++) {
      const int e = e0 + u * POSE_THREADS + threadIdx.x;
      fl[u] = m.pt_flags[(size_t)s * P + idx[u]];
      fnd[u] = e < n && (fl[u] & TDF_FOUND);
    }
#pragma unroll
    for (int u = 0; u < POSE_GB; u++) {
      const TrackData& t = td[idx[u]];
      for (int i = 0; i < 3; i++) v[u][i] = t.cam[i];
      for (int i = 0; i < 2; i++) { v[u][3 + i] = t.image[i]; v[u][9 + i] = t.vfound[i]; }
      for (int i = 0; i < 4; i++) v[u][5 + i] = t.derivs[i];
      v[u][11] = t.sqrt_inv_noise;
    }
#pragma unroll
    for (int u = 0; u < POSE_GB; u++) { bm[u] = __ballot(fnd[u]); if (lane == 0) gcnt[u * POSE_WAVES + wave] = __popcll(bm[u]); }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < POSE_GB; u++) {
      int off = nf;
      for (int w = 0; w < POSE_WAVES; w++) { const int c = gcnt[u * POSE_WAVES + w]; if (w < wave) off += c; nf += c; }
      if (!fnd[u]) continue;
      const int f = off + __popcll(bm[u] & ((1ull << lane) - 1ull));
      ws.i[f] = fl[u]; ws.i[P + f] = idx[u];
#pragma unroll
      for (int c = 0; c < 12; c++) ws.d[c * P + f] = v[u][c];
    }
    __syncthreads();
  }
  POSE_STAMP(0);
  for (int iter = 0; iter < 10; iter++) {                            // coarse :466-488, fine :543-577
    const bool nonlinear = stage == 0 || iter == 0 || iter == 4 || iter == 9;
    // POSE_ILP entries per thread in flight: all their loads are issued (index clamped, no branch in between) before any
    // of them is advanced -- the loop is latency-bound at one workgroup of four waves per stream
    for (int e0 = threadIdx.x; e0 < nf; e0 += POSE_ILP * POSE_THREADS) {
      PoseItem t[POSE_ILP]; double pos[POSE_ILP][3];
#pragma unroll
      for (int u = 0; u < POSE_ILP; u++) {
        const int e = e0 + u * POSE_THREADS;
        item_load(t[u], ws, e < nf ? e : nf - 1);
      }
      if (nonlinear && iter != 0) {
#pragma unroll
        for (int u = 0; u < POSE_ILP; u++)
#pragma unroll
          for (int k = 0; k < 3; k++) pos[u][k] = pts[t[u].idx].pos[k];
      }
#pragma unroll
      for (int u = 0; u < POSE_ILP; u++) {
        const int e = e0 + u * POSE_THREADS;
        if (e >= nf) continue;
        if (iter != 0) {
          if (nonlinear) td_project_and_derivs(t[u], t[u].flags, pos[u], pose, tp.cam);
          else {                                                   // LinearUpdate, jni/TrackerData.h:125-131
            double jac[12], a = 0, b = 0;
            td_calc_jacobian(t[u], jac);                           // m26Jacobian of the last non-linear iteration
#pragma unroll
            for (int k = 0; k < 6; k++) { a += jac[k] * last_up[k]; b += jac[6 + k] * last_up[k]; }
            t[u].image[0] += a; t[u].image[1] += b;
          }
          item_store(t[u], ws, e, nonlinear);
        }
        const double r0 = (t[u].vfound[0] - t[u].image[0]) * t[u].sqrt_inv_noise;   // v2Error_CovScaled, :707
        const double r1 = (t[u].vfound[1] - t[u].image[1]) * t[u].sqrt_inv_noise;
        sortbuf[e] = r0 * r0 + r1 * r1;
      }
    }
    __syncthreads();
    POSE_STAMP(1);
    const double override_sigma = iter > 5 ? (stage == 0 ? 1.0 : 16.0) : 0.0;
    calc_pose_update(ws, pts, nf, tp, override_sigma, stage == 1 && iter == 9, sortbuf, red, up, hist, sel);
    if (threadIdx.x == 0) pose = pose_mul(se3_exp(up), pose);        // :487 / :573
    if (threadIdx.x < 6) last_up[threadIdx.x] = up[threadIdx.x];
    __syncthreads();
    POSE_STAMP(7);
  }
  POSE_STAMP(8);
  // scatter what the iterations changed, batched like the gather; the fine stage exports the measurements of the found
  // patches (:594-607) and leaves their depths in LDS for the scene-depth sums (:610-625)
  MeasDev* cm = m.cur_meas + (size_t)s * P;
  if (stage != 0) {
    for (int i = threadIdx.x; i < st->n_points; i += POSE_THREADS) cm[i].valid = 0;
    __syncthreads();
  }
  for (int e0 = threadIdx.x; e0 < nf; e0 += POSE_GB * POSE_THREADS) {
    int idx[POSE_GB], fl[POSE_GB], lv[POSE_GB]; double v[POSE_GB][11];
#pragma unroll
    for (int u = 0; u < POSE_GB; u++) {
      const int e = e0 + u * POSE_THREADS, ec = e < nf ? e : nf - 1;
      idx[u] = ws.i[P + ec]; fl[u] = ws.i[ec];
#pragma unroll
      for (int c = 0; c < 11; c++) v[u][c] = ws.d[c * P + ec];
    }
    if (stage != 0) {
#pragma unroll
      for (int u = 0; u < POSE_GB; u++) lv[u] = m.pt_level[(size_t)s * P + idx[u]];
    }
#pragma unroll
    for (int u = 0; u < POSE_GB; u++) {
      const int e = e0 + u * POSE_THREADS;
      if (e >= nf) continue;
      TrackData& t = td[idx[u]];
      m.pt_flags[(size_t)s * P + idx[u]] = fl[u];
      for (int i = 0; i < 3; i++) t.cam[i] = v[u][i];
      for (int i = 0; i < 2; i++) t.image[i] = v[u][3 + i];
      for (int i = 0; i < 4; i++) t.derivs[i] = v[u][5 + i];
      if (stage != 0) {
        MeasDev mm;
        mm.root[0] = v[u][9]; mm.root[1] = v[u][10];
        mm.valid = 1; mm.level = (signed char)lv[u]; mm.subpix = (fl[u] & TDF_SUBPIX) ? 1 : 0; mm.source = 0 /* SRC_TRACKER */; mm.pad = 0;
        cm[idx[u]] = mm;
        sortbuf[e] = v[u][2];
      }
    }
  }
  if (stage == 0) {
    if (threadIdx.x == 0) { st->pose_cur = pose; st->did_coarse = 1; }
    return;
  }
  // ---- scene depth (:610-625): dSum += z, dSumSq += z * z over the found points in iteration-set order ----
  __syncthreads();
  if (threadIdx.x < 2) {
    double a = 0.0;
    const bool sq = threadIdx.x == 1;
    int k = 0;
    for (; k + 8 <= nf; k += 8) {
      double z[8];
#pragma unroll
      for (int u = 0; u < 8; u++) z[u] = sortbuf[k + u];
#pragma unroll
      for (int u = 0; u < 8; u++) a += sq ? z[u] * z[u] : z[u];
    }
    for (; k < nf; k++) { const double z = sortbuf[k]; a += sq ? z * z : z; }
    red[threadIdx.x] = a;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double dSum = red[0], dSumSq = red[1];
  const int nNum = nf;
  if (nNum > 20) {
    st->depth_mean = dSum / nNum;
    st->depth_sigma = sqrt((dSumSq / nNum) - (st->depth_mean) * (st->depth_mean));
  }
  st->pose_cur = pose; st->pose_final = pose;
  {                                                                  // UpdateMotionModel, :802-820
    const Pose nfo = pose_mul(pose, pose_inverse(st->start_pose));
    double motion[6];
    se3_ln(nfo, motion);
    double ss = 0;
    for (int i = 0; i < 6; i++) {
      st->velocity[i] = 0.9 * (0.5 * motion[i] + 0.5 * st->velocity[i]);
      double v = st->velocity[i];
      if (i < 3) v *= 1.0 / st->depth_mean;
      ss += v * v;
    }
    st->msd_vel = sqrt(ss);
  }
  const Pose* kfp = m.kf_pose + (size_t)s * tp.max_keyframes;
  double closest = 9999999999.9;                                     // ClosestKeyFrame, jni/MapMaker.cc:737-758
  for (int k = 0; k < st->n_kf; k++) { const double d = kf_linear_dist(pose, kfp[k]); if (d < closest) closest = d; }
  {                                                                  // AssessTrackingQuality, :832-878
    int ta = 0, tf = 0, la = 0, lf = 0;
    for (int i = 0; i < NLEV; i++) { ta += st->attempted[i]; tf += st->found[i]; if (i >= 2) { la += st->attempted[i]; lf += st->found[i]; } }
    int quality;
    if (tf == 0 || ta == 0) quality = 0;
    else {
      const double dTotal = (double)tf / ta;
      const double dLarge = la > 10 ? (double)lf / la : dTotal;
      if (dTotal > 0.3) quality = 2; else if (dLarge < 0.13) quality = 0; else quality = 1;
    }
    if (quality == 1 && closest > tp.wiggle_scale * 10.0) quality = 0;   // IsDistanceToNearestKeyFrameExcessive
    if (quality == 0) st->lost_frames++; else st->lost_frames = 0;
    st->quality = quality;
  }
  {                                                                  // jni/Tracker.cc:128-132 + NeedNewKeyFrame, jni/MapMaker.cc:761-773
    double dDist = closest;
    dDist *= (1.0 / st->depth_mean);
    const bool need = dDist > tp.max_kf_dist_wiggle_mult * st->wiggle_depth_norm;
    if (st->quality == 2 && need && st->frame - st->last_kf_dropped > tp.min_frames_between_kf && st->n_kf < tp.max_keyframes) {
      st->kf_pending = 1;
      st->last_kf_dropped = st->frame;
    }
  }
  POSE_STAMP(9);
}

#ifdef VSLAM_BA_PROF
extern "C" int vslam_debug_pose_prof(unsigned long long* out16, int reset) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_pose_prof), sizeof(unsigned long long) * 16));
  if (reset) { unsigned long long z[16] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_pose_prof), z, sizeof(z))); }
  return VSLAM_OK;
}
#endif

// ---- host side ----------------------------------------------------------------------------------------------------
template <class T>
static int dalloc(vslam_system* sys, T** out, size_t count) {
  void* ptr = nullptr;
  HIPCHK(hipMalloc(&ptr, count * sizeof(T) + 64));
  HIPCHK(hipMemsetAsync(ptr, 0, count * sizeof(T) + 64, sys->stream));
  sys->allocs.push_back(ptr);
  *out = (T*)ptr;
  return VSLAM_OK;
}
#define TALLOC(ptr, count) do { int _r = dalloc(sys, &(ptr), (count)); if (_r) return _r; } while (0)

int trk_alloc(vslam_system* sys) {
  const vslam_params& p = sys->p;
  const size_t S = sys->S, P = p.max_points, K = p.max_keyframes;
  if (P > SORT_CAP) { vslam_set_error("max_points %d exceeds %d", (int)P, SORT_CAP); return VSLAM_E_INVALID; }
  MapDev& m = sys->map;
  TALLOC(m.pts, S * P); TALLOC(m.td, S * P); TALLOC(m.tmpl, S * P * TMPL_PITCH);
  TALLOC(m.kf_meas, S * K * P); TALLOC(m.cur_meas, S * P);
  TALLOC(m.kf_pose, S * K); TALLOC(m.kf_fixed, S * K); TALLOC(m.kf_depth, S * K * 2);
  for (int l = 0; l < NLEV; l++) TALLOC(m.kf_img[l], S * K * (size_t)sys->geom[l].pitch * sys->geom[l].h);
  TALLOC(m.st, S); TALLOC(m.pvs_list, S * NLEV * P); TALLOC(m.search_list, S * P); TALLOC(m.iter_list, S * P); TALLOC(m.pt_level, S * P); TALLOC(m.pt_flags, S * P);
  TALLOC(m.pose_ws, S * POSE_WS_COMPS * P); TALLOC(m.pose_wsi, S * 2 * P);
  trk_fill_params(p, sys->tp);
  return VSLAM_OK;
}

// device camera block per ATANCamera::SetImageSize + RefreshParams (jni/ATANCamera.cc:31-82)
void cam_fill(CamModel& c, const double cam5[5], double width, double height, int quirks) {
  c.size[0] = width; c.size[1] = height;
  c.focal[0] = c.size[0] * cam5[0]; c.focal[1] = c.size[1] * cam5[1];
  c.center[0] = c.size[0] * cam5[2] - 0.5; c.center[1] = c.size[1] * cam5[3] - 0.5;
  c.w = cam5[4];
  double one_over_2tan = 0;
  if (c.w != 0.0) { c.two_tan = 2.0 * vlm::vtan(c.w / 2.0); one_over_2tan = 1.0 / c.two_tan; c.winv = 1.0 / c.w; c.distortion_enabled = 1.0; }
  else { c.winv = 0; c.two_tan = 0; c.distortion_enabled = 0; }
  double v2[2];
  if (quirks & VSLAM_Q_CAM_INT_RADIUS) {   // :70-78 int-typed operands (quirk #5)
    int m1 = (int)cam5[2], m2 = (int)(1.0 - cam5[2]);
    v2[0] = (m1 > m2 ? m1 : m2) / cam5[0];
    m1 = (int)cam5[3]; m2 = (int)(1.0 - cam5[3]);
    v2[1] = (m1 > m2 ? m1 : m2) / cam5[1];
  } else {
    v2[0] = (cam5[2] > 1.0 - cam5[2] ? cam5[2] : 1.0 - cam5[2]) / cam5[0];
    v2[1] = (cam5[3] > 1.0 - cam5[3] ? cam5[3] : 1.0 - cam5[3]) / cam5[1];
  }
  const double rr = sqrt(v2[0] * v2[0] + v2[1] * v2[1]);
  c.largest_radius = c.w == 0.0 ? rr : vlm::vtan(rr * c.w) * one_over_2tan;   // invrtrans, jni/ATANCamera.h:145-150
  c.max_r = 1.5 * c.largest_radius;
}

// device parameter block
void trk_fill_params(const vslam_params& p, TrackParams& t) {
  cam_fill(t.cam, p.cam, p.width, p.height, p.quirks);
  t.P = p.patch_size; t.max_ssd = 500 * p.patch_size * p.patch_size;
  t.max_patches = p.max_patches_per_frame; t.coarse_min = p.coarse_min; t.coarse_max = p.coarse_max; t.coarse_range = p.coarse_range;
  t.coarse_subpix_its = p.coarse_subpix_its; t.coarse_disabled = p.coarse_disabled; t.fine_subpix_its = p.fine_subpix_its;
  t.coarse_min_vel = p.coarse_min_vel; t.wls_prior = p.wls_prior;
  t.min_frames_between_kf = p.min_frames_between_kf; t.max_kf_dist_wiggle_mult = p.max_kf_dist_wiggle_mult; t.wiggle_scale = p.wiggle_scale;
  t.ba_max_iterations = p.ba_max_iterations; t.ba_convergence_limit = p.ba_convergence_limit;
  t.ba_min_sigma2 = p.ba_min_tukey_sigma * p.ba_min_tukey_sigma; t.ba_window = p.ba_window; t.ba_min_keyframes = p.ba_min_keyframes;
  t.quirks = p.quirks; t.max_points = p.max_points; t.max_keyframes = p.max_keyframes; t.ba_delay = p.ba_delay_frames;
  t.ba_batch = p.ba_batch_frames > 1 ? p.ba_batch_frames : 1;
  t.ba_sum_order = p.ba_sum_order ? 1 : 0;
  t.grow_map = p.grow_map;
  t.idle = p.idle_iterations != 0; t.fq_cap = 8192;
  {                                                                  // ATANCamera::OnePixelDist, jni/ATANCamera.cc:86-91
    double a[2], b[2];
    cam_unproject(t.cam, t.cam.size[0] / 2, t.cam.size[1] / 2, a);
    cam_unproject(t.cam, t.cam.size[0] / 2 + 1, t.cam.size[1] / 2 + 1, b);
    const double d0 = a[0] - b[0], d1 = a[1] - b[1];
    t.one_pixel_dist = sqrt(d0 * d0 + d1 * d1) / sqrt(2.0);
  }
  const int kcap_max[NLEV] = {16384, 8192, 4096, 2048};               // stored corner list of a keyframe (grow_map): see DESIGN.md
  for (int l = 0; l < NLEV; l++) t.kcap[l] = p.max_corners[l] < kcap_max[l] ? p.max_corners[l] : kcap_max[l];

}

static void trk_search_args(vslam_system* sys, SearchArgs& a) {
  for (int l = 0; l < NLEV; l++) {
    a.img[l] = sys->fr.img[l]; a.img_sstride[l] = sys->fr.img_sstride[l]; a.img_pitch[l] = sys->fr.img_pitch[l];
    a.corners[l] = sys->fr.corners[l]; a.rowlut[l] = sys->fr.rowlut[l];
    a.w[l] = sys->geom[l].w; a.h[l] = sys->geom[l].h; a.cap[l] = sys->geom[l].cap;
    a.kf_pitch[l] = sys->geom[l].pitch; a.kf_stride[l] = (size_t)sys->geom[l].pitch * sys->geom[l].h;
  }
  a.ncorners = sys->fr.ncorners;
}

// One search stage of TrackMap on the current frame.  stage 0: ApplyMotionModel + the potentially visible set (:369-392),
// coarse selection (:399-461) and SearchForPoints of the coarse set; stage 1: fine selection (:493-535) and SearchForPoints of
// the level-3 points (with sub-pixel refinement) and of the rest.
int trk_search_stage(vslam_system* sys, int stage) {
  const int S = sys->S, P = sys->p.max_points;
  MapDev& m = sys->map;
  const TrackParams& tp = sys->tp;
  SearchArgs a;
  trk_search_args(sys, a);
  a.S = S; a.nblk = 1;
  if (stage == 0) {
    prof_mark(sys, 3);
    hipLaunchKernelGGL(k_motion, dim3((S + 63) / 64), dim3(64), 0, sys->stream, m, S, sys->p.use_sbi ? (const double*)sys->fr.sbi_rot : (const double*)nullptr);
    hipLaunchKernelGGL(k_pvs, dim3((P + TRK_THREADS - 1) / TRK_THREADS, S), dim3(TRK_THREADS), 0, sys->stream, m, tp);
    prof_mark(sys, 4);
    hipLaunchKernelGGL(k_plan, dim3(S), dim3(TRK_THREADS), 0, sys->stream, m, tp, 0);
    prof_mark(sys, 5);
    if (!tp.coarse_disabled) {
      const int nc = 2 * tp.coarse_max;
      if (tp.P == 8) {
        a.nblk = (nc + 7) / 8; hipLaunchKernelGGL((k_searchN<8, 8>), dim3(xcd_grid(a.nblk, S)), dim3(64), 0, sys->stream, m, tp, a, 0);
        if (tp.coarse_subpix_its > 0) { a.nblk = SUBPIX_GRID; hipLaunchKernelGGL((k_subpixN<8, 8>), dim3(xcd_grid(a.nblk, S)), dim3(64), 0, sys->stream, m, tp, a, 0); }
      } else {
        a.nblk = (nc + 3) / 4; hipLaunchKernelGGL((k_searchN<11, 16>), dim3(xcd_grid(a.nblk, S)), dim3(64), 0, sys->stream, m, tp, a, 0);
        if (tp.coarse_subpix_its > 0) { a.nblk = SUBPIX_GRID; hipLaunchKernelGGL((k_subpixN<11, 16>), dim3(xcd_grid(a.nblk, S)), dim3(64), 0, sys->stream, m, tp, a, 0); }
      }
    }
    prof_mark(sys, 6);
  } else {
    const int maxSearch = tp.max_patches + 2 * tp.coarse_max < P ? tp.max_patches + 2 * tp.coarse_max : P;
    prof_mark(sys, 7);
    hipLaunchKernelGGL(k_plan, dim3(S), dim3(TRK_THREADS), 0, sys->stream, m, tp, 1);
    prof_mark(sys, 8);
    if (tp.P == 8) {
      a.nblk = (maxSearch + 7) / 8; hipLaunchKernelGGL((k_searchN<8, 8>), dim3(xcd_grid(a.nblk, S)), dim3(64), 0, sys->stream, m, tp, a, 1);
      if (tp.fine_subpix_its > 0) { a.nblk = SUBPIX_GRID; hipLaunchKernelGGL((k_subpixN<8, 8>), dim3(xcd_grid(a.nblk, S)), dim3(64), 0, sys->stream, m, tp, a, 1); }
    } else {
      a.nblk = (maxSearch + 3) / 4; hipLaunchKernelGGL((k_searchN<11, 16>), dim3(xcd_grid(a.nblk, S)), dim3(64), 0, sys->stream, m, tp, a, 1);
      if (tp.fine_subpix_its > 0) { a.nblk = SUBPIX_GRID; hipLaunchKernelGGL((k_subpixN<11, 16>), dim3(xcd_grid(a.nblk, S)), dim3(64), 0, sys->stream, m, tp, a, 1); }
    }
    prof_mark(sys, 9);
  }
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

// The ten Gauss-Newton iterations of a stage (:466-488 / :543-577); stage 1 also ends TrackMap and TrackFrame on device
// (measurement export, scene depth, UpdateMotionModel, AssessTrackingQuality, the keyframe decision).
int trk_pose_stage(vslam_system* sys, int stage) {
  if (stage == 0 && sys->tp.coarse_disabled) return VSLAM_OK;
  hipLaunchKernelGGL(k_pose, dim3(sys->S), dim3(POSE_THREADS), 0, sys->stream, sys->map, sys->tp, stage);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

int trk_track_map(vslam_system* sys) {
  int r = trk_search_stage(sys, 0);
  if (!r) r = trk_pose_stage(sys, 0);
  if (!r) r = trk_search_stage(sys, 1);
  if (!r) r = trk_pose_stage(sys, 1);
  return r;
}
