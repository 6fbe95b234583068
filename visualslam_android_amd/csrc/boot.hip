// Map bootstrap on the device (vslam_params.bootstrap), for every stream that has no map yet:
//   Tracker::TrackForInitialMap / TrailTracking_Start / TrailTracking_Advance      jni/Tracker.cc:247-346
//   HomographyInit::Compute                                                        jni/HomographyInit.cc:43-71
//   MapMaker::InitFromStereo, RefreshSceneDepth                                    jni/MapMaker.cc:204-376, 1236-1252
//   MapMaker::CalcPlaneAligner, ApplyGlobalTransformationToMap                     jni/MapMaker.cc:1104-1231, 440-449
// One-shot work, not a hot path: everything is gated per stream on device state (no host round trip), the data-parallel parts
// (MiniPatch searches, sub-pixel alignment, the epipolar growth, the bundle adjustments) are the kernels the tracker and the
// map-maker already use, the hypothesis loops of bootstrap_math.h are spread over a workgroup, the rest runs on one lane.
//
// The frame a spacebar press is consumed in runs, after the tracker's part of the frame:
//   k_boot_gate -> [start]  MakeKeyFrame_Rest (candidates), k_trail_start, k_add_keyframe + corner copy into keyframe slot 0
//               -> k_trail_advance (every frame while trails exist)
//               -> [second press]  k_boot_homography, k_add_keyframe + corners into slot 1, k_boot_points, 5 x BundleAdjustAll,
//                  k_boot_scene_depth, AddSomeMapPoints(0, 3, 1, 2), BundleAdjustAll until converged, k_boot_plane.
// The existing map-maker kernels are gated on kf_pending and address "the keyframe being added" as slot n_kf: k_boot_phase sets
// those two for the streams InitFromStereo runs for, and back.
#include "vslam_internal.h"
#include "grow_dev.h"
#include "bootstrap_math.h"

#define BOOT_THREADS 256
#define BOOT_WAVES (BOOT_THREADS / 64)
#define MPH 4            // MiniPatch::mnHalfPatchSize, jni/MiniPatch.cc:86
#define MPS 9
#define MPP 81
#define BOOT_MAX_SSD 100000   // Tracker.MiniPatchMaxSSD, jni/Tracker.cc:249

struct BootFrame {       // level 0 of the current and of the previous frame (the front end's two buffers)
  const uint8_t* img; size_t sstride; int pitch;
  const uint32_t* corners; const int* rowlut; const int* ncorners; int cap;
};
struct BootArgs {
  BootFrame cur, prev;
  int w, h;
  const uint32_t* cand; const double* cand_score; const int* ncand; int cand_cap;   // level-0 candidates of the current frame (MakeKeyFrame_Rest)
  int kf_pitch; size_t kf_stride;                                                  // level 0 of the stored keyframes
};

DEVFN Pose pose_identity() { Pose T; for (int i = 0; i < 9; i++) T.R[i] = i % 4 == 0 ? 1.0 : 0.0; T.t[0] = T.t[1] = T.t[2] = 0.0; return T; }
DEVFN uint8_t* trail_patches(const MapDev& m, int s, int buf) { return m.trail_patch + ((size_t)s * 2 + buf) * BOOT_MAX_TRAILS * MPP; }
DEVFN int* trail_positions(const MapDev& m, int s, int buf) { return m.trail_pos + ((size_t)s * 2 + buf) * BOOT_MAX_TRAILS * 4; }

// What this frame does for a stream without a map (TrackForInitialMap's switch, :252-288), one lane per stream.  Every kf_pending
// of the frame has been served by now; the flag is cleared for all streams so that the gated map-maker kernels below see only
// the streams the bootstrap sets it for.
__global__ void k_boot_gate(MapDev m, int S) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  TrackerState* st = &m.st[s];
  st->kf_pending = 0;
  st->boot_action = 0;
  if (st->map_good || st->init_stage == 2) return;
  if (st->init_stage == 0) {
    if (st->spacebar) { st->spacebar = 0; st->boot_action = 1; st->kf_pending = 1; st->n_kf = 0; st->n_points = 0; st->pose_final = pose_identity(); }
  } else st->boot_action = 2;
}

// TrailTracking_Start, :290-318: the (at most) 1000 level-0 candidates that come first in the reference's sort -- ascending
// Shi-Tomasi score (its functor compares the NEGATED scores with >), equal scores in candidate order -- become trails.
__global__ __launch_bounds__(BOOT_THREADS) void k_trail_start(MapDev m, BootArgs a) {
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  if (st->boot_action != 1) return;
  __shared__ int sh_cnt;
  if (threadIdx.x == 0) sh_cnt = 0;
  __syncthreads();
  const int n = a.ncand[s * NLEV + 0];
  const uint32_t* cand = a.cand + (size_t)s * a.cand_cap;
  const double* score = a.cand_score + (size_t)s * a.cand_cap;
  const uint8_t* img = a.cur.img + (size_t)s * a.cur.sstride;
  uint8_t* patches = trail_patches(m, s, 0);
  int* pos = trail_positions(m, s, 0);
  auto inb = [&](uint32_t c) { const int x = c & 0xFFFF, y = c >> 16; return x >= MPH && y >= MPH && x < a.w - MPH && y < a.h - MPH; };
  int mine = 0;
  for (int i = threadIdx.x; i < n; i += BOOT_THREADS) {
    const uint32_t c = cand[i];
    if (!inb(c)) continue;
    mine++;
    const double sc = score[i];
    int rank = 0;
    for (int j = 0; j < n; j++) { const double sj = score[j]; if ((sj < sc || (sj == sc && j < i)) && inb(cand[j])) rank++; }
    if (rank >= BOOT_MAX_TRAILS) continue;
    const int x = c & 0xFFFF, y = c >> 16;
    for (int r = 0; r < MPS; r++) for (int cc = 0; cc < MPS; cc++) patches[(size_t)rank * MPP + r * MPS + cc] = img[(size_t)(y - MPH + r) * a.cur.pitch + (x - MPH + cc)];   // SampleFromImage
    pos[4 * rank] = x; pos[4 * rank + 1] = y; pos[4 * rank + 2] = x; pos[4 * rank + 3] = y;
  }
  atomicAdd(&sh_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) { st->n_trails = sh_cnt < BOOT_MAX_TRAILS ? sh_cnt : BOOT_MAX_TRAILS; st->trail_buf = 0; }
}

// after the keyframe copy of a start frame: the first keyframe sits in slot 0, the stage advances (:257-258)
__global__ void k_boot_started(MapDev m, int S) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  TrackerState* st = &m.st[s];
  if (st->boot_action != 1) return;
  st->kf_pending = 0; st->n_kf = 1; st->init_stage = 1;
}

// MiniPatch::FindPatch (jni/MiniPatch.cc:35-68) by one wavefront: the corners of rows [y - range, y + range] through the row
// look-up table, eight candidates per step (eight lanes each), first strict minimum in list order
DEVFN bool mp_find(const BootFrame& f, int s, int w, int h, const uint8_t* tmpl /* LDS */, int* cand /* LDS [64] */, int& px, int& py, int lane) {
  const uint8_t* img = f.img + (size_t)s * f.sstride;
  const uint32_t* corners = f.corners + (size_t)s * f.cap;
  const int* rowlut = f.rowlut + (size_t)s * (h + 1);
  const int ncorners = f.ncorners[s * NLEV + 0];
  const int range = 10;
  const int L = px - range, R = px + range, T = py - range, B = py + range;
  int best = BOOT_MAX_SSD + 1, bestIdx = 0x7fffffff;
  const int y0 = T < 0 ? 0 : T, y1 = B + 1;
  const int i0 = y0 >= h ? ncorners : rowlut[y0];
  const int i1 = y1 >= h ? ncorners : (y1 < 0 ? 0 : rowlut[y1]);
  const int grp = lane >> 3, sub = lane & 7;
  for (int base = i0; base < i1; base += 64) {
    const int ci = base + lane;
    bool ok = false;
    if (ci < i1) { const int cx = corners[ci] & 0xFFFF; ok = !(cx < L || cx > R); }
    const unsigned long long bm = __ballot(ok);
    const int nc = __popcll(bm);
    if (ok) cand[__popcll(bm & ((1ull << lane) - 1ull))] = ci;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int c0 = 0; c0 < nc; c0 += 8) {
      const int k = c0 + grp;
      int ssd = 0x7fffffff, cidx = 0x7fffffff;
      if (k < nc) {
        cidx = cand[k];
        const uint32_t c = corners[cidx];
        const int cx = c & 0xFFFF, cy = c >> 16;
        const bool inside = cx >= MPH && cy >= MPH && cx < w - MPH && cy < h - MPH;
        int sum = 0;
        if (inside) {
          const uint8_t* ib = img + (size_t)(cy - MPH) * f.pitch + (cx - MPH);
          for (int q = sub; q < MPP; q += 8) { const int r = q / MPS, cc = q - r * MPS; const int d = (int)ib[r * f.pitch + cc] - (int)tmpl[q]; sum += d * d; }
        }
        for (int d = 1; d < 8; d <<= 1) sum += __shfl_xor(sum, d);
        ssd = inside ? sum : BOOT_MAX_SSD + 1;
      }
      for (int d = 8; d < 64; d <<= 1) {
        const int os = __shfl_xor(ssd, d), oi = __shfl_xor(cidx, d);
        if (os < ssd || (os == ssd && oi < cidx)) { ssd = os; cidx = oi; }
      }
      if (ssd < best) { best = ssd; bestIdx = cidx; }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (best < BOOT_MAX_SSD) { const uint32_t c = corners[bestIdx]; px = c & 0xFFFF; py = c >> 16; return true; }
  return false;
}

// TrailTracking_Advance, :321-346, one wavefront per trail; the survivors are compacted in order into the other trail buffer.
// Then TrackForInitialMap's tail (:265-285): too few trails -> Reset; a pending spacebar -> InitFromStereo runs for this stream.
__global__ __launch_bounds__(BOOT_THREADS) void k_trail_advance(MapDev m, BootArgs a) {
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  if (st->boot_action != 2) return;
  __shared__ int sh_host;
  if (threadIdx.x == 0) sh_host = st->boot_host_matches;
  __syncthreads();
  if (sh_host) {                                                     // vslam_init_from_stereo: the matches are the caller's, nothing to search
    if (threadIdx.x == 0) { st->boot_host_matches = 0; if (st->spacebar) { st->spacebar = 0; st->boot_run = 1; st->boot_ok = 0; } }
    return;
  }
  __shared__ uint8_t sh_tmpl[BOOT_WAVES][MPP + 3];
  __shared__ int sh_cand[BOOT_WAVES][64];
  __shared__ int keep[BOOT_MAX_TRAILS + 1];
  __shared__ int sh_good, wsum[BOOT_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = st->n_trails, buf = st->trail_buf;
  uint8_t* patches = trail_patches(m, s, buf);
  int* pos = trail_positions(m, s, buf);
  if (threadIdx.x == 0) sh_good = 0;
  __syncthreads();
  const uint8_t* img = a.cur.img + (size_t)s * a.cur.sstride;
  int good = 0;
  for (int t = wave; t < n; t += BOOT_WAVES) {
    for (int k = lane; k < MPP; k += 64) sh_tmpl[wave][k] = patches[(size_t)t * MPP + k];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const int sx = pos[4 * t + 2], sy = pos[4 * t + 3];
    int ex = sx, ey = sy;
    bool found = mp_find(a.cur, s, a.w, a.h, sh_tmpl[wave], sh_cand[wave], ex, ey, lane);
    if (found) {                                                     // the married-matches check (:334-339)
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < MPP; k += 64) { const int r = k / MPS, c = k - r * MPS; sh_tmpl[wave][k] = img[(size_t)(ey - MPH + r) * a.cur.pitch + (ex - MPH + c)]; }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      int bx = ex, by = ey;
      found = mp_find(a.prev, s, a.w, a.h, sh_tmpl[wave], sh_cand[wave], bx, by, lane);
      const int dx = bx - sx, dy = by - sy;
      if (dx * dx + dy * dy > 2) found = false;
      if (lane == 0) { pos[4 * t + 2] = ex; pos[4 * t + 3] = ey; }
      good++;                                                        // counted before the backward check, as there (:341)
    }
    if (lane == 0) keep[t] = found ? 1 : 0;
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) atomicAdd(&sh_good, good);
  __syncthreads();
  // ordered compaction (the list's erase, :343-344)
  const int per = (n + BOOT_THREADS - 1) / BOOT_THREADS;
  const int lo = min((int)threadIdx.x * per, n), hi = min(lo + per, n);
  int cnt = 0;
  for (int i = lo; i < hi; i++) cnt += keep[i];
  int inc = cnt;
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = inc - cnt, total = 0;
  for (int w = 0; w < BOOT_WAVES; w++) { if (w < wave) base += wsum[w]; total += wsum[w]; }
  uint8_t* np = trail_patches(m, s, buf ^ 1);
  int* npos = trail_positions(m, s, buf ^ 1);
  for (int i = lo; i < hi; i++) {
    if (!keep[i]) continue;
    for (int k = 0; k < MPP; k++) np[(size_t)base * MPP + k] = patches[(size_t)i * MPP + k];
    for (int k = 0; k < 4; k++) npos[4 * base + k] = pos[4 * i + k];
    base++;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    st->n_trails = total; st->trail_buf = buf ^ 1;
    if (sh_good < 10) { st->n_trails = 0; st->init_stage = 0; st->n_kf = 0; }        // Reset(), :266-269
    else if (st->spacebar) { st->spacebar = 0; st->boot_run = 1; st->boot_ok = 0; }  // :272-281
  }
}

// ATANCamera::UnProject followed by GetProjectionDerivs (jni/ATANCamera.cc:149-164, 198-231)
DEVFN void unproject_with_derivs(const CamModel& c, double ix, double iy, double out[2], double jac[4]) {
  const double dx = (ix - c.center[0]) * (1.0 / c.focal[0]), dy = (iy - c.center[1]) * (1.0 / c.focal[1]);
  const double dist_r = sqrt(dx * dx + dy * dy);
  const double rr = c.w == 0.0 ? dist_r : vlm::vtan(dist_r * c.w) * (1.0 / c.two_tan);
  const double f = dist_r > 0.01 ? rr / dist_r : 1.0;
  const double last_factor = 1.0 / f;
  out[0] = dx * f; out[1] = dy * f;
  double fx, fy;
  const double k = c.two_tan, x = out[0], y = out[1], r = rr * c.distortion_enabled;
  if (r < 0.01) { fx = 0.0; fy = 0.0; }
  else {
    fx = c.winv * (k * x) / (r * r * (1 + k * k * r * r)) - x * last_factor / (r * r);
    fy = c.winv * (k * y) / (r * r * (1 + k * k * r * r)) - y * last_factor / (r * r);
  }
  jac[0] = c.focal[0] * (fx * x + last_factor); jac[2] = c.focal[1] * (fx * y);
  jac[1] = c.focal[0] * (fy * x); jac[3] = c.focal[1] * (fy * y + last_factor);
}

// InitFromStereo up to the second camera's pose (jni/MapMaker.cc:204-258): the matches, HomographyInit::Compute, the scale
__global__ __launch_bounds__(BOOT_THREADS) void k_boot_homography(MapDev m, TrackParams tp) {
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  if (!st->boot_run) return;
  __shared__ double sh_err[BOOT_THREADS];
  __shared__ int sh_trial[BOOT_THREADS];
  __shared__ double sh_H[9];
  __shared__ int wsum[BOOT_WAVES], sh_ninl;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = st->n_trails;
  const int* pos = trail_positions(m, s, st->trail_buf);
  bm::Match* mt = (bm::Match*)(m.boot_match + (size_t)s * BOOT_MAX_TRAILS * 8);
  int* inl = m.boot_inl + (size_t)s * BOOT_MAX_TRAILS;
  for (int i = threadIdx.x; i < n; i += BOOT_THREADS) {              // :210-229 (the derivatives of the second position are the ones kept)
    bm::Match q; double j0[4];
    unproject_with_derivs(tp.cam, (double)pos[4 * i], (double)pos[4 * i + 1], q.first, j0);
    unproject_with_derivs(tp.cam, (double)pos[4 * i + 2], (double)pos[4 * i + 3], q.second, q.jac);
    mt[i] = q;
  }
  __syncthreads();
  const double max2 = 5.0 * 5.0;                                     // HomographyInit.Compute(vMatches, 5.0, se3), :236
  bool ok = n >= 4;
  if (ok) {
    if (n < 10) { if (threadIdx.x == 0) bm::homography_from_matches(mt, nullptr, n, sh_H); }
    else {                                                           // BestHomographyFromMatches_MLESAC, :232-262
      double best = 999999999999999999.9; int bt = 0x7fffffff;
      for (int t = threadIdx.x; t < 300; t += BOOT_THREADS) {
        double H[9];
        const double e = bm::mlesac_trial(mt, n, st->boot_seed, t, max2, H);
        if (e < best) { best = e; bt = t; }
      }
      sh_err[threadIdx.x] = best; sh_trial[threadIdx.x] = bt;
      __syncthreads();
      if (threadIdx.x == 0) {
        double be = 999999999999999999.9; int t0 = -1;
        for (int t = 0; t < 300; t++) {                              // the first strict minimum in trial order
          const int owner = t % BOOT_THREADS;
          if (sh_trial[owner] == t && sh_err[owner] < be) { be = sh_err[owner]; t0 = t; }
        }
        // a thread's best is its FIRST minimum, and a later trial of the same thread can only have won with a strictly smaller
        // error: the scan above sees every trial that could be the global first minimum
        for (int i = 0; i < 9; i++) sh_H[i] = i % 4 == 0 ? 1.0 : 0.0;
        if (t0 >= 0) { double H[9]; bm::mlesac_trial(mt, n, st->boot_seed, t0, max2, H); for (int i = 0; i < 9; i++) sh_H[i] = H[i]; }
      }
    }
    __syncthreads();
    // the inlier set, in match order (:53-56)
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += BOOT_THREADS) {
      const int i = i0 + threadIdx.x;
      const bool in = i < n && bm::pixel_error_squared(sh_H, mt[i]) < max2;
      const unsigned long long bmk = __ballot(in);
      __syncthreads();
      if (lane == 0) wsum[wave] = __popcll(bmk);
      __syncthreads();
      int off = base;
      for (int w = 0; w < wave; w++) off += wsum[w];
      if (in) inl[off + __popcll(bmk & ((1ull << lane) - 1ull))] = i;
      for (int w = 0; w < BOOT_WAVES; w++) base += wsum[w];
    }
    if (threadIdx.x == 0) sh_ninl = base;
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  Pose se3 = pose_identity();
  if (ok) {
    double H[9];
    for (int i = 0; i < 9; i++) H[i] = sh_H[i];
    double* ws = m.boot_ws + (size_t)s * (3 * (size_t)tp.max_points > BOOT_MAX_TRAILS ? 3 * (size_t)tp.max_points : BOOT_MAX_TRAILS);
    for (int it = 0; it < 5; it++) bm::refine_homography(H, mt, inl, sh_ninl, ws);   // :58-59
    bm::Decomposition d[8];
    if (bm::decompose_homography(H, d) != 8) ok = false;             // :62-66
    else {
      bm::choose_best_decomposition(d, H, mt, n, inl, sh_ninl, max2);
      for (int i = 0; i < 9; i++) se3.R[i] = d[0].R[i];
      for (int i = 0; i < 3; i++) se3.t[i] = d[0].t[i];
      const double mag = sqrt(se3.t[0] * se3.t[0] + se3.t[1] * se3.t[1] + se3.t[2] * se3.t[2]);
      if (mag == 0) ok = false;                                      // :243-248
      else for (int i = 0; i < 3; i++) se3.t[i] *= tp.wiggle_scale / mag;   // :250
    }
    st->n_hom_inliers = sh_ninl;
  }
  if (!ok) { st->boot_run = 0; st->boot_ok = 0; st->init_stage = 2; return; }        // the tracker's stage is COMPLETE either way, jni/Tracker.cc:279
  st->boot_ok = 1;
  const size_t K = tp.max_keyframes;
  m.kf_pose[(size_t)s * K + 0] = pose_identity(); m.kf_fixed[(size_t)s * K + 0] = 1;   // pkFirst, :256-257
  st->pose_final = se3; st->depth_mean = 0; st->depth_sigma = 0;                       // pkSecond's pose for k_add_keyframe (slot n_kf = 1)
  st->kf_pending = 1;
}

// The points of the stereo pair (jni/MapMaker.cc:263-337), one wavefront per match: template of the first keyframe, sub-pixel
// alignment in the second, triangulation; appended in match order.
template <int PS>
__global__ __launch_bounds__(BOOT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_boot_points(MapDev m, TrackParams tp, BootArgs a) {
  constexpr int NPIX = PS * PS, HALF = PS / 2;
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  if (!st->boot_run) return;
  struct Res { int ok; double pos[3], right[3], down[3], sub[2]; int cx, cy; };
  __shared__ Res res[BOOT_WAVES];
  __shared__ uint8_t sh_tmpl[BOOT_WAVES][128];
  __shared__ double sh_slab[BOOT_WAVES][3 * (PS - 2) * (PS - 2)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = tp.max_keyframes, P = tp.max_points;
  const int n = st->n_trails;
  const int* pos = trail_positions(m, s, st->trail_buf);
  const bm::Match* mt = (const bm::Match*)(m.boot_match + (size_t)s * BOOT_MAX_TRAILS * 8);
  const uint8_t* img0 = m.kf_img[0] + ((size_t)s * K + 0) * a.kf_stride;
  const uint8_t* img1 = m.kf_img[0] + ((size_t)s * K + 1) * a.kf_stride;
  const Pose se3 = m.kf_pose[(size_t)s * K + 1];
  uint8_t* tmpl = sh_tmpl[wave];
  for (int c0 = 0; c0 < n; c0 += BOOT_WAVES) {
    const int i = c0 + wave;
    bool alive = i < n;
    if (lane == 0) res[wave].ok = 0;
    int cx = 0, cy = 0;
    double sub0 = 0, sub1 = 0;
    if (alive) {
      cx = pos[4 * i]; cy = pos[4 * i + 1]; sub0 = (double)pos[4 * i + 2]; sub1 = (double)pos[4 * i + 3];
      const int bord = HALF + 1;                                     // MakeTemplateCoarseNoWarp, jni/PatchFinder.cc:130-142
      if (!(cx >= bord && cy >= bord && cx < a.w - bord && cy < a.h - bord)) alive = false;
    }
    if (alive) for (int q = lane; q < NPIX; q += 64) { const int y = q / PS, x = q - y * PS; tmpl[q] = img0[(size_t)(cy - HALF + y) * a.kf_pitch + (cx - HALF + x)]; }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    if (alive && !wave_subpix<PS>(tmpl, img1, a.kf_pitch, a.w, a.h, 0, 10, lane, sub0, sub1, sh_slab[wave])) alive = false;   // :298-304
    if (alive && lane == 0) {
      double uB[2], wp[3];
      cam_unproject(tp.cam, sub0, sub1, uB);
      reproject_point(se3, uB, mt[i].first, wp);                     // :309-311
      if (!(wp[2] < 0.0)) {
        // RefreshPixelVectors (jni/MapPoint.cc:4-29) in the first keyframe (identity pose); :271-292 take the "right" neighbour one
        // pixel DOWN and the "down" neighbour one pixel RIGHT -- the reference's
        double cen[3], rgt[3], dwn[3];
        unit_ray(tp.cam, (double)cx, (double)cy, cen); unit_ray(tp.cam, (double)cx + 0, (double)cy + 1, rgt); unit_ray(tp.cam, (double)cx + 1, (double)cy + 0, dwn);
        const double hgt = fabs(-wp[2]), rc = fabs(-cen[2]), rr = fabs(-rgt[2]), rd = fabs(-dwn[2]);
        Res& r = res[wave];
        for (int k = 0; k < 3; k++) { const double cop = cen[k] * hgt / rc; r.right[k] = rgt[k] * hgt / rr - cop; r.down[k] = dwn[k] * hgt / rd - cop; r.pos[k] = wp[k]; }
        r.sub[0] = sub0; r.sub[1] = sub1; r.cx = cx; r.cy = cy; r.ok = 1;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                          // mMap.vpPoints.push_back + the two measurements, :318-336
      for (int wv = 0; wv < BOOT_WAVES; wv++) {
        if (!res[wv].ok) continue;
        const int pid = st->n_points;
        if (pid >= P) continue;
        const Res& r = res[wv];
        MapPointDev mp;
        for (int k = 0; k < 3; k++) { mp.pos[k] = r.pos[k]; mp.right[k] = r.right[k]; mp.down[k] = r.down[k]; }
        mp.src_kf = 0; mp.src_level = 0; mp.irx = r.cx; mp.iry = r.cy; mp.bad = 0; mp.n_in = 0; mp.n_out = 0; mp.n_meas_kfs = 2;
        m.pts[(size_t)s * P + pid] = mp;
        TrackData td;
        for (int k = 0; k < 3; k++) td.cam[k] = 0;
        for (int k = 0; k < 2; k++) { td.image[k] = 0; td.vfound[k] = 0; }
        for (int k = 0; k < 4; k++) { td.derivs[k] = 0; td.warp_inv[k] = 0; td.last_warp[k] = 0; }
        td.sqrt_inv_noise = 0; td.tsum = 0; td.tsumsq = 0;
        td.last_warp[0] = 9999.9; td.last_warp[3] = 9999.9;
        m.td[(size_t)s * P + pid] = td;
        m.pt_level[(size_t)s * P + pid] = -1; m.pt_flags[(size_t)s * P + pid] = 0;
        MeasDev mm;
        mm.valid = 1; mm.level = 0; mm.subpix = 1; mm.pad = 0;
        mm.source = 2 /* SRC_ROOT */; mm.root[0] = (double)r.cx; mm.root[1] = (double)r.cy;
        m.kf_meas[((size_t)s * K + 0) * P + pid] = mm;
        mm.source = 3 /* SRC_TRAIL */; mm.root[0] = r.sub[0]; mm.root[1] = r.sub[1];
        m.kf_meas[((size_t)s * K + 1) * P + pid] = mm;
        m.cur_meas[(size_t)s * P + pid].valid = 0;
        st->n_points = pid + 1;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { st->n_init_points = st->n_points; st->newq_head = st->n_points; }
}

// Switches between "the second keyframe is being added" (slot n_kf = 1, kf_pending: what the growth kernels expect) and "two
// keyframes in the map" (what the bundle adjustment expects) for the streams InitFromStereo runs for.
//   1: after the stereo points -- two keyframes, the map counts as good for the adjustment kernels
//   2: before AddSomeMapPoints -- slot 1 pending again
//   3: after AddSomeMapPoints -- two keyframes, mbBundleConverged_Full = false (:358-359)
__global__ void k_boot_phase(MapDev m, int S, int what) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  TrackerState* st = &m.st[s];
  if (!st->boot_run) return;
  if (what == 1) { st->kf_pending = 0; st->n_kf = 2; st->map_good = 1; st->kf_added = 0; }
  else if (what == 2) { st->kf_pending = 1; st->n_kf = 1; }
  else { st->kf_pending = 0; st->n_kf = 2; st->ba_converged_full = 0; st->ba_converged_recent = 0; }
}

// RefreshSceneDepth of the two keyframes (jni/MapMaker.cc:1236-1252; :349-351), sums in point order
__global__ void k_boot_scene_depth(MapDev m, TrackParams tp) {
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  if (!st->boot_run) return;
  const int k = threadIdx.x;
  if (k >= 2) return;
  const size_t K = tp.max_keyframes, P = tp.max_points;
  const Pose T = m.kf_pose[(size_t)s * K + k];
  double sum = 0.0, sumsq = 0.0; int n = 0;
  for (int i = 0; i < st->n_points; i++) {
    if (!m.kf_meas[((size_t)s * K + k) * P + i].valid) continue;
    double c[3];
    pose_xform(T, m.pts[(size_t)s * P + i].pos, c);
    sum += c[2]; sumsq += c[2] * c[2]; n++;
  }
  const double mean = sum / n;
  m.kf_depth[((size_t)s * K + k) * 2] = mean;
  m.kf_depth[((size_t)s * K + k) * 2 + 1] = sqrt((sumsq / n) - (mean) * (mean));
  if (k == 0) st->wiggle_depth_norm = tp.wiggle_scale / mean;        // mdWiggleScaleDepthNormalized
}

// CalcPlaneAligner + ApplyGlobalTransformationToMap, then the end of InitFromStereo (:366-372) and the tracker's side of it
__global__ __launch_bounds__(BOOT_THREADS) void k_boot_plane(MapDev m, TrackParams tp) {
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  if (!st->boot_run) return;
  __shared__ double sh_err[128];
  __shared__ double sh_T[12];
  __shared__ int sh_have;
  const size_t K = tp.max_keyframes, P = tp.max_points;
  const int n = st->n_points, nk = st->n_kf;
  MapPointDev* pts = m.pts + (size_t)s * P;
  double* pos = m.boot_ws + (size_t)s * (3 * P > BOOT_MAX_TRAILS ? 3 * P : BOOT_MAX_TRAILS);
  for (int i = threadIdx.x; i < n; i += BOOT_THREADS) for (int k = 0; k < 3; k++) pos[3 * i + k] = pts[i].pos[k];
  __syncthreads();
  if (n >= 10) {                                                     // :1107-1110
    if (threadIdx.x < 100) { double mean[3], nrm[3]; sh_err[threadIdx.x] = bm::plane_trial(pos, n, st->boot_seed + 1u, (int)threadIdx.x, mean, nrm); }
    __syncthreads();
    if (threadIdx.x == 0) {
      double best = 9999999999999999.9; int bt = -1;
      for (int t = 0; t < 100; t++) if (!(sh_err[t] < 0.0) && sh_err[t] < best) { best = sh_err[t]; bt = t; }
      double mean[3] = {0, 0, 0}, nrm[3] = {0, 0, 1};
      if (bt >= 0) bm::plane_trial(pos, n, st->boot_seed + 1u, bt, mean, nrm);
      sh_have = bm::plane_aligner(pos, n, mean, nrm, sh_T, sh_T + 9) ? 1 : 0;
    }
  } else if (threadIdx.x == 0) sh_have = 0;
  __syncthreads();
  if (sh_have) {                                                     // ApplyGlobalTransformationToMap, :440-449
    Pose T;
    for (int i = 0; i < 9; i++) T.R[i] = sh_T[i];
    for (int i = 0; i < 3; i++) T.t[i] = sh_T[9 + i];
    const Pose Tinv = pose_inverse(T);
    for (int k = threadIdx.x; k < nk; k += BOOT_THREADS) m.kf_pose[(size_t)s * K + k] = pose_mul(m.kf_pose[(size_t)s * K + k], Tinv);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += BOOT_THREADS) {
      MapPointDev p = pts[i];
      double np3[3];
      pose_xform(T, p.pos, np3);
      for (int k = 0; k < 3; k++) p.pos[k] = np3[k];
      // RefreshPixelVectors (jni/MapPoint.cc:4-29) against the transformed source keyframe; the stereo points keep the swapped neighbours
      const Pose Tk = m.kf_pose[(size_t)s * K + p.src_kf];
      const int sc = 1 << p.src_level;
      const double cx = level_zero_pos((double)p.irx, p.src_level), cy = level_zero_pos((double)p.iry, p.src_level);
      const bool stereo = p.src_kf == 0 && i < st->n_init_points;
      double cen[3], rgt[3], dwn[3], pc[3];
      unit_ray(tp.cam, cx, cy, cen);
      if (stereo) { unit_ray(tp.cam, cx, cy + 1, rgt); unit_ray(tp.cam, cx + 1, cy, dwn); }
      else { unit_ray(tp.cam, cx + sc, cy, rgt); unit_ray(tp.cam, cx, cy + sc, dwn); }
      pose_xform(Tk, p.pos, pc);
      const double hgt = fabs(-pc[2]), rc = fabs(-cen[2]), rr = fabs(-rgt[2]), rd = fabs(-dwn[2]);
      double dr[3], dd[3];
      for (int k = 0; k < 3; k++) { const double cop = cen[k] * hgt / rc; dr[k] = rgt[k] * hgt / rr - cop; dd[k] = dwn[k] * hgt / rd - cop; }
      rot_inv(Tk, dr, p.right); rot_inv(Tk, dd, p.down);
      pts[i] = p;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const Pose T1 = m.kf_pose[(size_t)s * K + 1];                    // se3TrackerPose = pkSecond->se3CfromW, :370
    st->pose_final = T1; st->pose_cur = T1; st->start_pose = T1;
    for (int i = 0; i < 6; i++) st->velocity[i] = 0.0;
    st->lost_frames = 0; st->quality = 2;
    st->init_stage = 2; st->boot_run = 0; st->kf_pending = 0; st->kf_added = 0;
  }
}

__global__ void k_boot_press(MapDev m, int S, int stream, unsigned seed, int set_seed) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S || (stream >= 0 && s != stream)) return;
  if (set_seed) m.st[s].boot_seed = seed; else m.st[s].spacebar = 1;
}

// ---- host --------------------------------------------------------------------------------------------------------------------
int boot_alloc(vslam_system* sys) {
  sys->map.trail_patch = nullptr; sys->map.trail_pos = nullptr; sys->map.boot_match = nullptr; sys->map.boot_inl = nullptr; sys->map.boot_ws = nullptr;
  if (!sys->p.bootstrap) return VSLAM_OK;
  const size_t S = sys->S, P = sys->p.max_points;
  auto get = [&](size_t bytes, void** out) -> int { void* q = nullptr; HIPCHK(hipMalloc(&q, bytes + 64)); HIPCHK(hipMemsetAsync(q, 0, bytes + 64, sys->stream)); sys->allocs.push_back(q); *out = q; return VSLAM_OK; };
  int r;
  if ((r = get(S * 2 * BOOT_MAX_TRAILS * MPP, (void**)&sys->map.trail_patch))) return r;
  if ((r = get(S * 2 * BOOT_MAX_TRAILS * 4 * sizeof(int), (void**)&sys->map.trail_pos))) return r;
  if ((r = get(S * BOOT_MAX_TRAILS * 8 * sizeof(double), (void**)&sys->map.boot_match))) return r;
  if ((r = get(S * BOOT_MAX_TRAILS * sizeof(int), (void**)&sys->map.boot_inl))) return r;
  if ((r = get(S * (3 * P > BOOT_MAX_TRAILS ? 3 * P : BOOT_MAX_TRAILS) * sizeof(double), (void**)&sys->map.boot_ws))) return r;
  return VSLAM_OK;
}

static void boot_args(vslam_system* sys, BootArgs& a) {
  const LevelGeom& g = sys->geom[0];
  const FrameDev& c = sys->fr; const FrameDev& p = sys->frbuf[sys->fr_idx ^ 1];
  a.cur.img = c.img[0]; a.cur.sstride = c.img_sstride[0]; a.cur.pitch = c.img_pitch[0]; a.cur.corners = c.corners[0]; a.cur.rowlut = c.rowlut[0]; a.cur.ncorners = c.ncorners; a.cur.cap = g.cap;
  a.prev.img = p.img[0]; a.prev.sstride = p.img_sstride[0]; a.prev.pitch = p.img_pitch[0]; a.prev.corners = p.corners[0]; a.prev.rowlut = p.rowlut[0]; a.prev.ncorners = p.ncorners; a.prev.cap = g.cap;
  a.w = g.w; a.h = g.h;
  a.cand = sys->cand[0]; a.cand_score = sys->cand_score[0]; a.ncand = sys->ncand; a.cand_cap = g.cap;
  a.kf_pitch = g.pitch; a.kf_stride = (size_t)g.pitch * g.h;
}

// Tracker::TrackForInitialMap for every stream without a map, after the tracker's part of the frame
int boot_frame(vslam_system* sys) {
  if (!sys->p.bootstrap) return VSLAM_OK;
  const int S = sys->S;
  const dim3 gs((S + 63) / 64), bs(64);
  hipStream_t q = sys->stream;
  BootArgs a;
  boot_args(sys, a);
  const bool key = sys->boot_key_pressed;
  sys->boot_key_pressed = false;
  int r;
  hipLaunchKernelGGL(k_boot_gate, gs, bs, 0, q, sys->map, S);
  if (key) {                                                         // a first press may be consumed: TrailTracking_Start
    if ((r = fe_keyframe_rest_gated(sys))) return r;                 // mCurrentKF.MakeKeyFrame_Rest(), :292
    hipLaunchKernelGGL(k_trail_start, dim3(S), dim3(BOOT_THREADS), 0, q, sys->map, a);
    if ((r = ba_launch_add_keyframe(sys))) return r;                 // mFirstKF = mCurrentKF, :293 (kept in keyframe slot 0)
    if ((r = grow_copy_corners(sys))) return r;
    hipLaunchKernelGGL(k_boot_started, gs, bs, 0, q, sys->map, S);
  }
  hipLaunchKernelGGL(k_trail_advance, dim3(S), dim3(BOOT_THREADS), 0, q, sys->map, a);
  if (key) {                                                         // a second press may be consumed: InitFromStereo
    hipLaunchKernelGGL(k_boot_homography, dim3(S), dim3(BOOT_THREADS), 0, q, sys->map, sys->tp);
    if ((r = fe_keyframe_rest_gated(sys))) return r;                 // the second keyframe's candidates (its MakeKeyFrame_Rest, :342)
    if ((r = ba_launch_add_keyframe(sys))) return r;                 // *pkSecond = kS, :253
    if ((r = grow_copy_corners(sys))) return r;
    if (sys->tp.P == 8) hipLaunchKernelGGL(k_boot_points<8>, dim3(S), dim3(BOOT_THREADS), 0, q, sys->map, sys->tp, a);
    else hipLaunchKernelGGL(k_boot_points<11>, dim3(S), dim3(BOOT_THREADS), 0, q, sys->map, sys->tp, a);
    hipLaunchKernelGGL(k_boot_phase, gs, bs, 0, q, sys->map, S, 1);
    for (int i = 0; i < 5; i++) if ((r = ba_run(sys, 5))) return r;  // :344-345
    hipLaunchKernelGGL(k_boot_scene_depth, dim3(S), dim3(64), 0, q, sys->map, sys->tp);
    hipLaunchKernelGGL(k_boot_phase, gs, bs, 0, q, sys->map, S, 2);
    const int order[NLEV] = {0, 3, 1, 2};                            // :353-356
    if ((r = grow_levels(sys, order, NLEV))) return r;
    hipLaunchKernelGGL(k_boot_phase, gs, bs, 0, q, sys->map, S, 3);
    for (int i = 0; i < 50; i++) if ((r = ba_run(sys, 6))) return r; // while (!mbBundleConverged_Full) BundleAdjustAll(), :361-365 (bounded)
    hipLaunchKernelGGL(k_boot_plane, dim3(S), dim3(BOOT_THREADS), 0, q, sys->map, sys->tp);
  }
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

extern "C" int vslam_press_spacebar(vslam_system* sys, int stream) {
  if (!sys || stream >= sys->S) { vslam_set_error("press_spacebar: bad argument"); return VSLAM_E_INVALID; }
  if (!sys->p.bootstrap) { vslam_set_error("press_spacebar: created with bootstrap = 0"); return VSLAM_E_STATE; }
  hipLaunchKernelGGL(k_boot_press, dim3((sys->S + 63) / 64), dim3(64), 0, sys->stream, sys->map, sys->S, stream, 0u, 0);
  HIPCHK(hipGetLastError());
  sys->boot_key_pressed = true;
  return VSLAM_OK;
}

__global__ void k_boot_host_matches(MapDev m, int stream, int n) {
  TrackerState* st = &m.st[stream];
  st->n_trails = n; st->trail_buf = 0; st->boot_host_matches = 1;
}

// MapMaker::InitFromStereo(KeyFrame &kFirst, KeyFrame &kSecond, vector<pair<ImageRef, ImageRef>> &vMatches, mySE3 &se3CameraPos)
// (jni/MapMaker.h:38, jni/MapMaker.cc:204-376) for a caller that owns the two frames and the matches: the first image becomes the first
// keyframe exactly as a first spacebar press makes it (jni/Tracker.cc:290-318), the matches replace the trails, and the second image
// is the frame that consumes the second press -- HomographyInit, the stereo points, the five + until-converged BundleAdjustAll,
// AddSomeMapPoints, CalcPlaneAligner -- without TrailTracking_Advance's search.  One-stream systems (the reference's shape).
extern "C" int vslam_init_from_stereo(vslam_system* sys, const uint8_t* gray_first, const uint8_t* gray_second, size_t row_stride, int n_matches,
                                      const int* matches_xyxy, double pose12_out[12]) {
  if (!sys || !gray_first || !gray_second || n_matches < 0 || (n_matches > 0 && !matches_xyxy)) { vslam_set_error("init_from_stereo: bad argument"); return VSLAM_E_INVALID; }
  if (!sys->p.bootstrap || sys->S != 1) { vslam_set_error("init_from_stereo: needs a one-stream system created with bootstrap = 1"); return VSLAM_E_STATE; }
  if (n_matches > BOOT_MAX_TRAILS) { vslam_set_error("init_from_stereo: at most %d matches (MaxInitialTrails, jni/Tracker.cc:305)", BOOT_MAX_TRAILS); return VSLAM_E_CAPACITY; }
  int info[6];
  int r = vslam_get_init_info(sys, 0, info); if (r) return r;
  if (info[5] || info[0] != 0) { vslam_set_error("init_from_stereo: the stream has a map or an initialisation in progress"); return VSLAM_E_STATE; }
  r = vslam_press_spacebar(sys, 0); if (r) return r;
  r = vslam_update(sys, gray_first, row_stride, 0); if (r) return r;                 // TrailTracking_Start: the first keyframe
  r = vslam_get_init_info(sys, 0, info); if (r) return r;
  if (info[0] != 1) { vslam_set_error("init_from_stereo: the first frame did not start the initialisation"); return VSLAM_E_STATE; }
  if (n_matches > 0) HIPCHK(hipMemcpy(sys->map.trail_pos, matches_xyxy, sizeof(int) * 4 * (size_t)n_matches, hipMemcpyHostToDevice));   // stream 0, trail buffer 0
  hipLaunchKernelGGL(k_boot_host_matches, dim3(1), dim3(1), 0, sys->stream, sys->map, 0, n_matches);
  r = vslam_press_spacebar(sys, 0); if (r) return r;
  r = vslam_update(sys, gray_second, row_stride, 0); if (r) return r;                // the frame of the second press: InitFromStereo
  r = vslam_get_init_info(sys, 0, info); if (r) return r;
  if (pose12_out && info[5]) {
    vslam_track_state st;
    r = vslam_get_state(sys, 0, &st); if (r) return r;
    for (int i = 0; i < 12; i++) pose12_out[i] = st.pose[i];
  }
  return info[2] && info[5] ? 1 : 0;                                                  // InitFromStereo's bool
}

extern "C" int vslam_set_boot_seed(vslam_system* sys, int stream, unsigned seed) {
  if (!sys || stream >= sys->S) { vslam_set_error("set_boot_seed: bad argument"); return VSLAM_E_INVALID; }
  if (!sys->p.bootstrap) { vslam_set_error("set_boot_seed: created with bootstrap = 0"); return VSLAM_E_STATE; }
  hipLaunchKernelGGL(k_boot_press, dim3((sys->S + 63) / 64), dim3(64), 0, sys->stream, sys->map, sys->S, stream, seed, 1);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

extern "C" int vslam_get_init_info(vslam_system* sys, int stream, int out[6]) {
  if (!sys || stream < 0 || stream >= sys->S || !out) { vslam_set_error("get_init_info: bad argument"); return VSLAM_E_INVALID; }
  TrackerState st;
  HIPCHK(hipStreamSynchronize(sys->stream));
  HIPCHK(hipMemcpy(&st, sys->map.st + stream, sizeof(st), hipMemcpyDeviceToHost));
  out[0] = st.init_stage; out[1] = st.n_trails; out[2] = st.boot_ok; out[3] = st.n_hom_inliers; out[4] = st.n_init_points; out[5] = st.map_good;
  return VSLAM_OK;
}

extern "C" int vslam_get_trails(vslam_system* sys, int stream, int* out4, int cap, int* n) {
  if (!sys || stream < 0 || stream >= sys->S || (cap > 0 && !out4)) { vslam_set_error("get_trails: bad argument"); return VSLAM_E_INVALID; }
  if (!sys->p.bootstrap) { vslam_set_error("get_trails: created with bootstrap = 0"); return VSLAM_E_STATE; }
  TrackerState st;
  HIPCHK(hipStreamSynchronize(sys->stream));
  HIPCHK(hipMemcpy(&st, sys->map.st + stream, sizeof(st), hipMemcpyDeviceToHost));
  if (n) *n = st.n_trails;
  const int k = st.n_trails < cap ? st.n_trails : cap;
  if (k > 0) HIPCHK(hipMemcpy(out4, sys->map.trail_pos + ((size_t)stream * 2 + st.trail_buf) * BOOT_MAX_TRAILS * 4, sizeof(int) * 4 * (size_t)k, hipMemcpyDeviceToHost));
  return VSLAM_OK;
}
