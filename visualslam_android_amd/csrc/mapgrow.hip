// Map growth on a new keyframe (first part; vslam_params.grow_map = 1), all streams at once and
// gated on the tracker's device-side keyframe request (kf_pending), between k_add_keyframe and the bundle adjustment:
//   MakeKeyFrame_Rest candidates (frontend.hip, gated)                          jni/KeyFrame.cc:53-95
//   for level in 3, 0, 1, 2 (jni/MapMaker.cc:498-501):
//     MapMaker::ThinCandidates(new keyframe, level)  (frontend.hip)             jni/MapMaker.cc:393-422
//     MapMaker::AddPointEpipolar for every remaining candidate  (k_epipolar)    jni/MapMaker.cc:525-703
//       PatchFinder::MakeTemplateCoarseNoWarp :130-142, ZMSSDAtPoint :352-380, MakeSubPixTemplate / IterateSubPixToConvergence
//       :242-350, MapMaker::ReprojectPoint :174-200, MapPoint::RefreshPixelVectors jni/MapPoint.cc:4-29
// One wavefront per candidate; the candidates of a level are independent of each other, new points are appended in
// candidate order by an ordered commit (the reference's push_back order).  Not built: ReFind* (see DESIGN.md).
// Eigen::JacobiSVD of the 4x4 triangulation matrix is third-party arithmetic, restated as a two-sided Jacobi SVD (parity unpinned).
#include "vslam_internal.h"

#define GROW_THREADS 256
#ifdef VSLAM_BA_PROF
// Diagnostic build only: clock64() stamps of block 0 / wavefront 0 per stage of k_epipolar (vslam_debug_grow_prof)
__device__ unsigned long long g_grow_prof[16];
#define REFIND_STAMP(id) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { const unsigned long long t_ = clock64(); g_grow_prof[id] += t_ - g_grow_prof[15]; g_grow_prof[15] = t_; } } while (0)
#define GROW_STAMP(id) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = clock64(); g_grow_prof[id] += t_ - g_grow_prof[15]; g_grow_prof[15] = t_; } } while (0)
#else
#define REFIND_STAMP(id) do { } while (0)
#define GROW_STAMP(id) do { } while (0)
#endif
#define GROW_WAVES (GROW_THREADS / 64)
#define REFIND_BLOCKS 32

struct GrowArgs {
  const uint32_t* cand[NLEV]; const int* ncand; int cap[NLEV];     // current frame's candidate lists [S][cap], counts [S][NLEV]
  double* cand_score[NLEV];   // a candidate AddPointEpipolar rejects gets its score replaced by -(stage at which it gave up): 1 ray, 2 line,
                              // 3 radius, 4 template border, 5 no corner on the line, 6 sub-pixel, 7 map full (vslam_read_candidates)
  int w[NLEV], h[NLEV], kf_pitch[NLEV]; size_t kf_stride[NLEV];
  const int* rowlut[NLEV];    // the current frame's row look-up tables [S][h_l + 1] (k_refind searches the new keyframe = this frame)
  double* tgt_implane; int tgt_cap;   // vImplaneCorners of the target keyframe at the level being processed, [S][tgt_cap][2] (k_target_implane)
};

struct EpiResult { int ok; double pos[3], right[3], down[3], root[2], sub[2]; int irx, iry; };

#include "grow_dev.h"

// MapMaker::AddKeyFrame's deep copy of Level::vCorners for the new keyframe (jni/KeyFrame.cc:104-112)
__global__ void k_copy_kf_corners(MapDev m, TrackParams tp, const uint32_t* c0, const uint32_t* c1, const uint32_t* c2, const uint32_t* c3,
                                  const int* ncorners, int cap0, int cap1, int cap2, int cap3) {
  const int l = blockIdx.x, s = blockIdx.y;
  const TrackerState* st = &m.st[s];
  if (!st->kf_pending) return;
  const uint32_t* src = (l == 0 ? c0 : (l == 1 ? c1 : (l == 2 ? c2 : c3))) + (size_t)s * (l == 0 ? cap0 : (l == 1 ? cap1 : (l == 2 ? cap2 : cap3)));
  const size_t slot = (size_t)s * tp.max_keyframes + st->n_kf;
  uint32_t* dst = m.kf_corners[l] + slot * tp.kcap[l];
  const int n = ncorners[s * NLEV + l] < tp.kcap[l] ? ncorners[s * NLEV + l] : tp.kcap[l];
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x == 0) m.kf_ncorners[slot * NLEV + l] = n;
}

// ClosestKeyFrame(new keyframe) among the existing ones, jni/MapMaker.cc:737-758 with KeyFrameLinearDist :705-712
DEVFN int closest_keyframe(const Pose* kfp, int n_old, const Pose& self) {
  const Pose is = pose_inverse(self);
  double best = 9999999999.9; int n = -1;
  for (int i = 0; i < n_old; i++) {
    const Pose ii = pose_inverse(kfp[i]);
    const double d0 = ii.t[0] - is.t[0], d1 = ii.t[1] - is.t[1], d2 = ii.t[2] - is.t[2];
    const double d = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    if (d < best) { best = d; n = i; }
  }
  return n;
}

// AddPointEpipolar's vImplaneCorners (jni/MapMaker.cc:612-620; the reference caches them per keyframe level): the target
// keyframe's corners of level nLevel un-projected once per keyframe event instead of once per candidate.
__global__ __launch_bounds__(256) void k_target_implane(MapDev m, TrackParams tp, GrowArgs a, int nLevel) {
  const int s = blockIdx.y;
  const TrackerState* st = &m.st[s];
  if (!st->kf_pending) return;
  const int K = tp.max_keyframes, ksrc = st->n_kf;
  __shared__ int sh_tgt;
  const Pose* kfp = m.kf_pose + (size_t)s * K;
  if (threadIdx.x == 0) sh_tgt = closest_keyframe(kfp, ksrc, kfp[ksrc]);
  __syncthreads();
  const int ktgt = sh_tgt;
  if (ktgt < 0) return;
  const uint32_t* tcorners = m.kf_corners[nLevel] + ((size_t)s * K + ktgt) * tp.kcap[nLevel];
  const int ntc = m.kf_ncorners[((size_t)s * K + ktgt) * NLEV + nLevel];
  double* out = a.tgt_implane + (size_t)s * a.tgt_cap * 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ntc; i += gridDim.x * blockDim.x) {
    const uint32_t cv = tcorners[i];
    double v2Im[2];   // UnProject of the level-zero position truncated to integer pixels
    cam_unproject(tp.cam, (double)(int)level_zero_pos((double)(cv & 0xFFFF), nLevel), (double)(int)level_zero_pos((double)(cv >> 16), nLevel), v2Im);
    out[2 * i] = v2Im[0]; out[2 * i + 1] = v2Im[1];
  }
}

template <int PS>
__global__ __launch_bounds__(GROW_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_epipolar(MapDev m, TrackParams tp, GrowArgs a, int nLevel) {
  constexpr int NPIX = PS * PS, HALF = PS / 2;
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  if (!st->kf_pending) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = tp.max_keyframes, P = tp.max_points;
  const int ksrc = st->n_kf;                                       // the slot k_add_keyframe has just filled (n_kf advances in k_ba_assemble)
  __shared__ int sh_tgt;
  __shared__ EpiResult res[GROW_WAVES];
  __shared__ uint8_t sh_tmpl[GROW_WAVES][128];
  __shared__ double sh_slab[GROW_WAVES][3 * (PS - 2) * (PS - 2)];
  constexpr int G = PS <= 8 ? 8 : 16, NPW = 64 / G;                 // lanes per patch, patches scored per wavefront step
  __shared__ int2 sh_sel[GROW_WAVES][NPW];
  int2* sel = sh_sel[wave];
  const int grp = lane / G, sub = lane % G;
  const Pose* kfp = m.kf_pose + (size_t)s * K;
  if (threadIdx.x == 0) sh_tgt = closest_keyframe(kfp, ksrc, kfp[ksrc]);
  __syncthreads();
  const int ktgt = sh_tgt;
  if (ktgt < 0) return;
  const Pose Tsrc = kfp[ksrc], Ttgt = kfp[ktgt];
  const int nLevelScale = 1 << nLevel;
  const int wl = a.w[nLevel], hl = a.h[nLevel], ip = a.kf_pitch[nLevel];
  const uint8_t* img_src = m.kf_img[nLevel] + ((size_t)s * K + ksrc) * a.kf_stride[nLevel];
  const uint8_t* img_tgt = m.kf_img[nLevel] + ((size_t)s * K + ktgt) * a.kf_stride[nLevel];
  const uint32_t* tcorners = m.kf_corners[nLevel] + ((size_t)s * K + ktgt) * tp.kcap[nLevel];
  const int ntc = m.kf_ncorners[((size_t)s * K + ktgt) * NLEV + nLevel];
  const double* timp = a.tgt_implane + (size_t)s * a.tgt_cap * 2;
  const uint32_t* cand = a.cand[nLevel] + (size_t)s * a.cap[nLevel];
  const int ncand = a.ncand[s * NLEV + nLevel];
  const double dMean = m.kf_depth[((size_t)s * K + ksrc) * 2], dSigma = m.kf_depth[((size_t)s * K + ksrc) * 2 + 1];
  uint8_t* tmpl = sh_tmpl[wave];

#ifdef VSLAM_BA_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0) g_grow_prof[15] = clock64();
#endif
  for (int c0 = 0; c0 < ncand; c0 += GROW_WAVES) {
    const int ci = c0 + wave;
    bool alive = ci < ncand;
    int why = 0;
    if (lane == 0) res[wave].ok = 0;
    double irx = 0, iry = 0, root0 = 0, root1 = 0, along0 = 0, along1 = 0, normal0 = 0, normal1 = 0, dNormDist = 0, dMinLen = 0, dMaxLen = 0;
    int ca = 0, cb = 0;
    if (alive) {                                                   // ---- geometry of the epipolar line, :544-591 (every lane alike)
      const uint32_t cpos = cand[ci];
      irx = (double)(cpos & 0xFFFF); iry = (double)(cpos >> 16);
      root0 = level_zero_pos(irx, nLevel); root1 = level_zero_pos(iry, nLevel);
      double ray[3], tmp[3], dirn[3];
      unit_ray(tp.cam, root0, root1, ray);
      rot_inv(Tsrc, ray, tmp);
      pose_rot(Ttgt, tmp, dirn);
      const double dStartDepth = tp.wiggle_scale > dMean - dSigma ? tp.wiggle_scale : dMean - dSigma;
      const double dEndDepth = 40 * tp.wiggle_scale < dMean + dSigma ? 40 * tp.wiggle_scale : dMean + dSigma;
      const Pose srcInv = pose_inverse(Tsrc);
      double centre[3];
      pose_xform(Ttgt, srcInv.t, centre);
      double rs[3], re[3];
      for (int i = 0; i < 3; i++) { rs[i] = centre[i] + dStartDepth * dirn[i]; re[i] = centre[i] + dEndDepth * dirn[i]; }
      if (re[2] <= rs[2]) { alive = false; why = 1; }
      if (re[2] <= 0.0) { alive = false; why = 1; }
      if (alive) {
        if (rs[2] <= 0.0) { const double f = 0.001 - rs[2] / dirn[2]; for (int i = 0; i < 3; i++) rs[i] += dirn[i] * f; }
        const double v2A[2] = {rs[0] / rs[2], rs[1] / rs[2]}, v2B[2] = {re[0] / re[2], re[1] / re[2]};
        along0 = v2A[0] - v2B[0]; along1 = v2A[1] - v2B[1];
        if (along0 * along0 + along1 * along1 < 0.00000001) { alive = false; why = 2; }
        else {
          const double n = sqrt(along0 * along0 + along1 * along1);
          along0 /= n; along1 /= n;
          normal0 = along1; normal1 = -along0;
          dNormDist = v2A[0] * normal0 + v2A[1] * normal1;
          if (fabs(dNormDist) > tp.cam.largest_radius) { alive = false; why = 3; }
          const double la = along0 * v2A[0] + along1 * v2A[1], lb = along0 * v2B[0] + along1 * v2B[1];
          dMinLen = (la < lb ? la : lb) - 0.05; dMaxLen = (la > lb ? la : lb) + 0.05;
          if (dMinLen < -2.0) dMinLen = -2.0;
          if (dMaxLen < -2.0) dMaxLen = -2.0;
          if (dMinLen > 2.0) dMinLen = 2.0;
          if (dMaxLen > 2.0) dMaxLen = 2.0;
        }
      }
      // ---- MakeTemplateCoarseNoWarp (jni/PatchFinder.cc:130-142) + MakeTemplateSums ----
      ca = (int)irx; cb = (int)iry;
      const int bord = HALF + 1;
      if (!(ca >= bord && cb >= bord && ca < wl - bord && cb < hl - bord)) { if (alive) why = 4; alive = false; }
    }
    GROW_STAMP(0);   // geometry
    int tsum = 0, tsumsq = 0;
    if (alive) {
      int sa = 0, sq = 0;
      for (int q = lane; q < NPIX; q += 64) {
        const int y = q / PS, x = q - y * PS;
        const int v = img_src[(size_t)(cb - HALF + y) * ip + (ca - HALF + x)];
        tmpl[q] = (uint8_t)v; sa += v; sq += v * v;
      }
      tsum = wsum_i(sa); tsumsq = wsum_i(sq);
    }
    __builtin_amdgcn_wave_barrier();
    GRow<PS> trow;
    _Pragma("unroll") for (int q = 0; q < (PS + 3) / 4; q++) trow.w[q] = 0u;
    if (alive && sub < PS) trow = grow_load_row_lds<PS>(tmpl + sub * PS);
    GROW_STAMP(1);   // template
    // ---- the target keyframe's corners near the epipolar line, :622-641: filter 64 at a time, score the survivors in order ----
    int nBest = -1, nBestZMSSD = tp.max_ssd + 1;
    if (alive) {
      const double dMaxDistDiff = tp.one_pixel_dist * (4.0 + 1.0 * nLevelScale), dMaxDistSq = dMaxDistDiff * dMaxDistDiff;
      for (int base = 0; base < ntc; base += 64) {
        bool ok = false;
        uint32_t cv = 0;
        if (base + lane < ntc) {
          cv = tcorners[base + lane];
          const double v2Im[2] = {timp[2 * (base + lane)], timp[2 * (base + lane) + 1]};   // vImplaneCorners (:612-620), k_target_implane
          const double dDistDiff = dNormDist - (v2Im[0] * normal0 + v2Im[1] * normal1);
          const double len = v2Im[0] * along0 + v2Im[1] * along1;
          ok = !(dDistDiff * dDistDiff > dMaxDistSq) && !(len < dMinLen) && !(len > dMaxLen);
        }
        unsigned long long bm = __ballot(ok);
        while (bm) {                                                 // NPW survivors per step, in list order
          const int rank = __popcll(bm & ((1ull << lane) - 1ull));
          const bool mine = ((bm >> lane) & 1ull) && rank < NPW;
          if (mine) sel[rank] = make_int2((int)cv, base + lane);
          const int nsel = min(NPW, (int)__popcll(bm));
          bm &= ~__ballot(mine);
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_wave_barrier();
          const int2 pick = sel[grp < nsel ? grp : 0];
          const int cx = pick.x & 0xFFFF, cy = (unsigned)pick.x >> 16;
          int ssd = tp.max_ssd + 1;
          const bool inb = cx >= HALF && cy >= HALF && cx < wl - HALF && cy < hl - HALF;
          {
            int sA = 0, sQ = 0, sX = 0;
            if (inb && grp < nsel && sub < PS) {
              const GRow<PS> ir = grow_load_row<PS>(img_tgt + (size_t)(cy - HALF + sub) * ip + (cx - HALF));
              _Pragma("unroll") for (int q = 0; q < (PS + 3) / 4; q++) {
                sA = (int)__builtin_amdgcn_udot4(ir.w[q], 0x01010101u, (unsigned)sA, false);
                sQ = (int)__builtin_amdgcn_udot4(ir.w[q], ir.w[q], (unsigned)sQ, false);
                sX = (int)__builtin_amdgcn_udot4(ir.w[q], trow.w[q], (unsigned)sX, false);
              }
            }
            sA = grow_grp_sum<G>(sA); sQ = grow_grp_sum<G>(sQ); sX = grow_grp_sum<G>(sX);
            const int SA = tsum, SB = sA;
            if (inb) ssd = ((2 * SA * SB - SA * SA - SB * SB) / NPIX + sQ + tsumsq - 2 * sX);   // ZMSSDAtPoint :352-380
          }
          _Pragma("unroll") for (int g = 0; g < NPW; g++) {          // first strict minimum in list order (:630-637)
            const int sg = __shfl(ssd, g * G), ig = __shfl(pick.y, g * G);
            if (g < nsel && sg < nBestZMSSD) { nBest = ig; nBestZMSSD = sg; }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      if (nBest == -1) { alive = false; why = 5; }
    }
    GROW_STAMP(2);   // corner filter + ZMSSD
    // ---- MakeSubPixTemplate + SetSubPixPos + IterateSubPixToConvergence(kTarget, 10), :658-664 ----
    double sub0 = 0, sub1 = 0;
    if (alive) {
      const uint32_t bc = tcorners[nBest];
      sub0 = level_zero_pos((double)(bc & 0xFFFF), nLevel); sub1 = level_zero_pos((double)(bc >> 16), nLevel);
      const bool converged = wave_subpix<PS>(tmpl, img_tgt, ip, wl, hl, nLevel, 10, lane, sub0, sub1, sh_slab[wave]);
      if (!converged) { alive = false; why = 6; }
    }
    GROW_STAMP(3);   // sub-pixel
    // ---- triangulation and the new point's patch vectors, :666-702 (lane 0) ----
    if (alive && lane == 0) {
      double uA[2], uB[2], pB[3], pw[3];
      cam_unproject(tp.cam, root0, root1, uA);
      cam_unproject(tp.cam, sub0, sub1, uB);
      const Pose tinv = pose_inverse(Ttgt);
      reproject_point(pose_mul(Tsrc, tinv), uA, uB, pB);
      pose_xform(tinv, pB, pw);
      double cen[3], rgt[3], dwn[3];
      unit_ray(tp.cam, root0, root1, cen); unit_ray(tp.cam, root0 + nLevelScale, root1, rgt); unit_ray(tp.cam, root0, root1 + nLevelScale, dwn);
      double pc[3];
      pose_xform(Tsrc, pw, pc);                                        // RefreshPixelVectors, normal (0, 0, -1)
      const double hgt = fabs(-pc[2]), rc = fabs(-cen[2]), rr = fabs(-rgt[2]), rd = fabs(-dwn[2]);
      double dr[3], dd[3];
      for (int i = 0; i < 3; i++) { const double cop = cen[i] * hgt / rc; dr[i] = rgt[i] * hgt / rr - cop; dd[i] = dwn[i] * hgt / rd - cop; }
      EpiResult& r = res[wave];
      rot_inv(Tsrc, dr, r.right); rot_inv(Tsrc, dd, r.down);
      for (int i = 0; i < 3; i++) r.pos[i] = pw[i];
      r.root[0] = root0; r.root[1] = root1; r.sub[0] = sub0; r.sub[1] = sub1; r.irx = ca; r.iry = cb;
      r.ok = 1;
    }
    GROW_STAMP(4);   // triangulation
    if (lane == 0 && ci < ncand && why) a.cand_score[nLevel][(size_t)s * a.cap[nLevel] + ci] = -(double)why;
    __syncthreads();
    GROW_STAMP(5);   // waiting for the other wavefronts of the chunk
    // ---- ordered commit: mMap.vpPoints.push_back + the two measurements, :692-701 ----
    if (threadIdx.x == 0) {
      for (int wv = 0; wv < GROW_WAVES; wv++) {
        if (!res[wv].ok) continue;
        const int pid = st->n_points;
        if (pid >= P) { if (c0 + wv < ncand) a.cand_score[nLevel][(size_t)s * a.cap[nLevel] + c0 + wv] = -7.0; continue; }   // map capacity: stop growing
        const EpiResult& r = res[wv];
        MapPointDev mp;
        for (int i = 0; i < 3; i++) { mp.pos[i] = r.pos[i]; mp.right[i] = r.right[i]; mp.down[i] = r.down[i]; }
        mp.src_kf = ksrc; mp.src_level = nLevel; mp.irx = r.irx; mp.iry = r.iry; mp.bad = 0; mp.n_in = 0; mp.n_out = 0; mp.n_meas_kfs = 2;
        m.pts[(size_t)s * P + pid] = mp;
        TrackData td;
        for (int i = 0; i < 3; i++) td.cam[i] = 0;
        for (int i = 0; i < 2; i++) { td.image[i] = 0; td.vfound[i] = 0; }
        for (int i = 0; i < 4; i++) { td.derivs[i] = 0; td.warp_inv[i] = 0; td.last_warp[i] = 0; }
        td.sqrt_inv_noise = 0; td.tsum = 0; td.tsumsq = 0;
        td.last_warp[0] = 9999.9; td.last_warp[3] = 9999.9;               // jni/PatchFinder.cc:23
        m.td[(size_t)s * P + pid] = td;
        m.pt_level[(size_t)s * P + pid] = -1; m.pt_flags[(size_t)s * P + pid] = 0;
        MeasDev mm;
        mm.valid = 1; mm.level = (signed char)nLevel; mm.subpix = 1; mm.pad = 0;
        for (int k = 0; k <= ksrc; k++) {                                 // the new column of the measurement table
          MeasDev z; z.root[0] = 0; z.root[1] = 0; z.valid = 0; z.level = 0; z.subpix = 0; z.source = 0; z.pad = 0;
          m.kf_meas[((size_t)s * K + k) * P + pid] = z;
        }
        mm.source = 2 /* SRC_ROOT */; mm.root[0] = r.root[0]; mm.root[1] = r.root[1];
        m.kf_meas[((size_t)s * K + ksrc) * P + pid] = mm;
        mm.source = 4 /* SRC_EPIPOLAR */; mm.root[0] = r.sub[0]; mm.root[1] = r.sub[1];
        m.kf_meas[((size_t)s * K + ktgt) * P + pid] = mm;
        m.cur_meas[(size_t)s * P + pid].valid = 0;
        st->n_points = pid + 1;
      }
    }
    __syncthreads();
    GROW_STAMP(6);   // ordered commit
  }
}

// MapMaker::ReFind_Common (jni/MapMaker.cc:967-1036) by one wavefront: try to measure map point `pid` in keyframe `k`.
// `rowlut` = the row look-up table of keyframe k's corner lists when k is the current frame (the new keyframe), else null (the
// rows are found by binary search in the stored list).  RefindCache is the function-static PatchFinder of the reference as far
// as it can matter: MakeTemplateCoarseCont (jni/PatchFinder.cc:79-125) keeps the template of the previous call when that call
// was for the same point and the warp moved by less than 0.07, and TemplateBad() then still says what the last generation said.
struct RefindCache { bool have_last, prev_bad; double last_warp[4]; int tsum, tsumsq; };

DEVFN void never_retry_set(const MapDev& m, const TrackParams& tp, int s, int pid, int k, int lane) {
  if (m.never_retry && lane == 0) atomicOr(&m.never_retry[((size_t)s * tp.max_points + pid) * 2 + (k >> 6)], 1ull << (k & 63));
}

template <int PS>
DEVFN bool refind_common(const MapDev& m, const TrackParams& tp, const GrowArgs& a, int s, int k, int pid, const int* rowlut,
                         uint8_t* tmpl, double* slab, RefindCache& cache, int lane) {
  constexpr int HALF = PS / 2;
  const int K = tp.max_keyframes, P = tp.max_points;
  MapPointDev& p = m.pts[(size_t)s * P + pid];
  MeasDev& cell = m.kf_meas[((size_t)s * K + k) * P + pid];
  if (cell.valid) return false;                                      // sMeasurementKFs.count(&k), :971
  if (m.never_retry && ((m.never_retry[((size_t)s * P + pid) * 2 + (k >> 6)] >> (k & 63)) & 1ull)) return false;   // sNeverRetryKFs, :972
  const Pose Tk = m.kf_pose[(size_t)s * K + k];
  double c[3];
  pose_xform(Tk, p.pos, c);
  if (c[2] < 0.001) { never_retry_set(m, tp, s, pid, k, lane); return false; }                                     // :979
  const double ip0 = c[0] / c[2], ip1 = c[1] / c[2];
  if (ip0 * ip0 + ip1 * ip1 > tp.cam.largest_radius * tp.cam.largest_radius) { never_retry_set(m, tp, s, pid, k, lane); return false; }   // :985
  const CamProj pr = cam_project(tp.cam, ip0, ip1);
  if (pr.invalid) { never_retry_set(m, tp, s, pid, k, lane); return false; }                                       // :991
  if (pr.im[0] < 0 || pr.im[1] < 0 || pr.im[0] > a.w[0] || pr.im[1] > a.h[0]) { never_retry_set(m, tp, s, pid, k, lane); return false; }   // :996
  double d[4];
  cam_derivs(tp.cam, pr, d);
  // CalcSearchLevelAndWarpMatrix, jni/PatchFinder.cc:31-68
  const double ooz = 1.0 / c[2];
  double mr[3], md[3];
  pose_rot(Tk, p.right, mr);
  pose_rot(Tk, p.down, md);
  const double r0 = mr[0] - c[0] * mr[2] * ooz, r1 = mr[1] - c[1] * mr[2] * ooz;
  const double d0 = md[0] - c[0] * md[2] * ooz, d1 = md[1] - c[1] * md[2] * ooz;
  const double wi[4] = {(d[0] * r0 + d[1] * r1) * ooz, (d[0] * d0 + d[1] * d1) * ooz, (d[2] * r0 + d[3] * r1) * ooz, (d[2] * d0 + d[3] * d1) * ooz};
  double det = wi[0] * wi[3] - wi[1] * wi[2];
  int level = 0;
  while (det > 3 && level < NLEV - 1) { level++; det *= 0.25; }
  const bool bad_scale = det > 3 || det < 0.25;                      // mbTemplateBad = true (:62-65); a regenerated template replaces the verdict
  const int scale = 1 << level;
  REFIND_STAMP(10);  // projection, derivatives, warp
  // MakeTemplateCoarseCont, :79-125: transform_image with the accumulated stepping of jni/vision/ImageHandler.cpp:21-113
  double inv[4];
  inv2(wi, inv);
  const double m2[4] = {inv[0] * scale, inv[1] * scale, inv[2] * scale, inv[3] * scale};
  bool refresh = !cache.have_last;
  for (int i = 0; !refresh && i < 2; i++) {
    const double dx = m2[i] - cache.last_warp[i], dy = m2[2 + i] - cache.last_warp[2 + i];
    if (dx * dx + dy * dy > 0.07 * 0.07) refresh = true;
  }
  bool bad;
  if (refresh) {
    int nOutside = 0, sum = 0, sumsq = 0;
    const int sl = p.src_level;
    const uint8_t* src = m.kf_img[sl] + ((size_t)s * K + p.src_kf) * a.kf_stride[sl];
    const int sp = a.kf_pitch[sl], iw = a.w[sl], ih = a.h[sl];
    const double across[2] = {m2[0], m2[2]}, down[2] = {m2[1], m2[3]};
    const double px0 = (double)p.irx - (m2[0] * HALF + m2[1] * HALF), py0 = (double)p.iry - (m2[2] * HALF + m2[3] * HALF);
    const double cr[2] = {down[0] - PS * across[0], down[1] - PS * across[1]};
    // one template pixel per lane (two for 11x11): the lane walks the accumulated sample position to its pixel with exactly
    // the additions transform_image makes (whole rows with their carriage return, then steps along the row)
    const float x_bound = (float)(iw - 1), y_bound = (float)(ih - 1);
    for (int q = lane; q < PS * PS; q += 64) {
      const int r = q / PS, j = q - r * PS;
      double x = px0, y = py0;
#pragma unroll 1
      for (int i = 0; i < r; i++) {
#pragma unroll
        for (int jj = 0; jj < PS; jj++) { x += across[0]; y += across[1]; }
        x += cr[0]; y += cr[1];
      }
#pragma unroll 1
      for (int jj = 0; jj < j; jj++) { x += across[0]; y += across[1]; }
      int v = 0;
      if (0 <= x && 0 <= y && x < x_bound && y < y_bound) {
        const int lx = (int)x, ly = (int)y;
        x -= lx; y -= ly;
        const uint8_t* q0 = src + (size_t)ly * sp + lx;
        v = (uint8_t)((1 - y) * ((1 - x) * q0[0] + x * q0[1]) + y * ((1 - x) * q0[sp] + x * q0[sp + 1]));
      } else nOutside++;
      tmpl[q] = (uint8_t)v;
      sum += v; sumsq += v * v;
    }
    nOutside = wsum_i(nOutside);
    cache.tsum = wsum_i(sum); cache.tsumsq = wsum_i(sumsq);
    __builtin_amdgcn_wave_barrier();
    bad = nOutside != 0;
    cache.have_last = true;
    for (int i = 0; i < 4; i++) cache.last_warp[i] = m2[i];
  } else bad = bad_scale ? true : cache.prev_bad;
  cache.prev_bad = bad;
  const int tsum = cache.tsum, tsumsq = cache.tsumsq;
  REFIND_STAMP(11);  // template
  if (bad) { never_retry_set(m, tp, s, pid, k, lane); return false; }                                              // TemplateBad, :1004
  // FindPatchCoarse(v2Image, k, 4), jni/PatchFinder.cc:170-235
  const int wl = a.w[level], hl = a.h[level], ip = a.kf_pitch[level];
  const uint8_t* img = m.kf_img[level] + ((size_t)s * K + k) * a.kf_stride[level];
  const uint32_t* corners = m.kf_corners[level] + ((size_t)s * K + k) * tp.kcap[level];
  const int nc = m.kf_ncorners[((size_t)s * K + k) * NLEV + level];
  const double irx = pr.im[0] / scale, iry = pr.im[1] / scale;
  const unsigned nRange = (4u + scale - 1) / scale;
  int nTop = (int)(iry - nRange);
  const int nBottomPlusOne = (int)(iry + nRange + 1);
  const int nLeft = (int)(irx - nRange), nRight = (int)(irx + nRange);
  if (nTop < 0) nTop = 0;
  if (nTop >= hl || nBottomPlusOne <= 0) { never_retry_set(m, tp, s, pid, k, lane); return false; }
  int i0, i1;
  if (rowlut) {
    // the new keyframe's corner list is the current frame's (k_copy_kf_corners), so the frame's row look-up table serves
    // (two loads instead of two binary searches of dependent global loads; indices clamped to the stored list)
    const int* lut = a.rowlut[level] + (size_t)s * (hl + 1);
    i0 = lut[nTop]; i1 = nBottomPlusOne >= hl ? nc : lut[nBottomPlusOne];
    if (i0 > nc) i0 = nc;
    if (i1 > nc) i1 = nc;
  } else {
    // Level::vCornerRowLUT of a stored keyframe = first corner of a row: lower bound of (row << 16) in the raster-ordered list
    auto lower = [&](int row) { int lo = 0, hi = nc; const uint32_t key = (uint32_t)row << 16; while (lo < hi) { const int mid = (lo + hi) >> 1; if (corners[mid] < key) lo = mid + 1; else hi = mid; } return lo; };
    i0 = lower(nTop); i1 = nBottomPlusOne >= hl ? nc : lower(nBottomPlusOne);
  }
  int bestx = -1, besty = -1, nBest = tp.max_ssd + 1;
  for (int base = i0; base < i1; base += 64) {
    bool ok = false;
    uint32_t cv = 0;
    if (base + lane < i1) {
      cv = corners[base + lane];
      const int cx = cv & 0xFFFF, cy = cv >> 16;
      const double dx = irx - cx, dy = iry - cy;
      ok = !(cx < nLeft || cx > nRight) && !(dx * dx + dy * dy > (double)(nRange * nRange));
    }
    unsigned long long bm = __ballot(ok);
    while (bm) {
      const int kk = __ffsll((long long)bm) - 1;
      bm &= bm - 1;
      const uint32_t cc = __shfl(cv, kk);
      const int cx = cc & 0xFFFF, cy = cc >> 16;
      const int ssd = wave_zmssd<PS>(tmpl, img, ip, wl, hl, cx, cy, tsum, tsumsq, tp.max_ssd, lane);
      if (ssd < nBest) { bestx = cx; besty = cy; nBest = ssd; }
    }
  }
  REFIND_STAMP(12);  // corner search + ZMSSD
  if (!(nBest < tp.max_ssd)) { never_retry_set(m, tp, s, pid, k, lane); return false; }                            // :1010
  double sub0 = level_zero_pos((double)bestx, level), sub1 = level_zero_pos((double)besty, level);
  if (level > 0) wave_subpix<PS>(tmpl, img, ip, wl, hl, level, 8, lane, sub0, sub1, slab);   // :1020-1024, convergence not looked at
  __builtin_amdgcn_wave_barrier();
  REFIND_STAMP(13);  // sub-pixel
  if (lane == 0) {                                                 // :1016-1034
    MeasDev mm;
    mm.valid = 1; mm.level = (signed char)level; mm.subpix = level > 0; mm.source = 1 /* SRC_REFIND */; mm.pad = 0;
    mm.root[0] = sub0; mm.root[1] = sub1;
    cell = mm;
    atomicAdd(&p.n_meas_kfs, 1);
  }
  return true;
}

// MapMaker::ReFindInSingleKeyFrame(new keyframe), jni/MapMaker.cc:1040-1056 (vslam_params.grow_map bit 1).
// One wavefront per map point, the points of a stream strided over the grid; a point writes only its own cell of the new
// keyframe's measurement row, so no ordering is needed.  ReFind_Common's function-static PatchFinder never sees the same
// point twice in a row here, so its template cache never hits and every template is warped afresh.
template <int PS>
__global__ __launch_bounds__(GROW_THREADS) void k_refind(MapDev m, TrackParams tp, GrowArgs a) {
  const int s = blockIdx.y;
  TrackerState* st = &m.st[s];
  if (!st->kf_pending) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int P = tp.max_points;
  const int ksrc = st->n_kf;
  __shared__ uint8_t sh_tmpl[GROW_WAVES][128];
  __shared__ double sh_slab[GROW_WAVES][3 * (PS - 2) * (PS - 2)];
  const int npts = st->n_points;
#ifdef VSLAM_BA_PROF
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_grow_prof[15] = clock64();
#endif
  for (int pid = blockIdx.x * GROW_WAVES + wave; pid < npts; pid += gridDim.x * GROW_WAVES) {
    REFIND_STAMP(8);   // previous iteration's tail / skipped points
    if (m.pts[(size_t)s * P + pid].bad) continue;                    // the trash list, jni/Map.cc:16-27
    REFIND_STAMP(9);   // point and cell loads
    RefindCache cache; cache.have_last = false; cache.prev_bad = false;
    refind_common<PS>(m, tp, a, s, ksrc, pid, a.rowlut[0] /* non-null: k is the current frame, its row LUTs serve */, sh_tmpl[wave], sh_slab[wave], cache, lane);
  }
}

// The idle jobs of MapMaker::run that re-find measurements (vslam_params.idle_iterations), gated per stream on device:
// mode 0  ReFindNewlyMade (jni/MapMaker.cc:1061-1081) once BundleAdjustRecent has converged: every point of the new queue (the
//         points AddPointEpipolar made since the last pass) against every keyframe, one wavefront per point walking the
//         keyframes in order (the PatchFinder's template cache lives across the keyframes of one point);
// mode 1  ReFindFromFailureQueue (:1083-1096) when the gate kernel drew it: one wavefront per queued (keyframe, point).  The
//         reference sorts the queue by address first; the entries do not interact (different points, or the same point in
//         different keyframes whose cells and never-retry bits are disjoint), so no order is imposed here.
template <int PS>
__global__ __launch_bounds__(GROW_THREADS) void k_refind_idle(MapDev m, TrackParams tp, GrowArgs a, int mode) {
  const int s = blockIdx.y;
  TrackerState* st = &m.st[s];
  if (!st->map_good) return;
  if (mode == 0 ? !st->ba_converged_recent : !st->idle_do_fail) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int P = tp.max_points;
  __shared__ uint8_t sh_tmpl[GROW_WAVES][128];
  __shared__ double sh_slab[GROW_WAVES][3 * (PS - 2) * (PS - 2)];
  const int nk = st->n_kf;
  int nfound = 0;
  if (mode == 0) {
    for (int pid = st->newq_head + blockIdx.x * GROW_WAVES + wave; pid < st->n_points; pid += gridDim.x * GROW_WAVES) {
      if (m.pts[(size_t)s * P + pid].bad) continue;                  // :1070-1073
      RefindCache cache; cache.have_last = false; cache.prev_bad = false;
      for (int k = 0; k < nk; k++) nfound += refind_common<PS>(m, tp, a, s, k, pid, nullptr, sh_tmpl[wave], sh_slab[wave], cache, lane) ? 1 : 0;
    }
  } else {
    const int2* fq = m.fq + (size_t)s * tp.fq_cap;
    const int n = st->fq_n < tp.fq_cap ? st->fq_n : tp.fq_cap;
    for (int e = blockIdx.x * GROW_WAVES + wave; e < n; e += gridDim.x * GROW_WAVES) {
      RefindCache cache; cache.have_last = false; cache.prev_bad = false;
      nfound += refind_common<PS>(m, tp, a, s, fq[e].x, fq[e].y, nullptr, sh_tmpl[wave], sh_slab[wave], cache, lane) ? 1 : 0;
    }
  }
  if (lane == 0 && nfound) atomicAdd(mode == 0 ? &st->n_refound_new : &st->n_refound_failed, nfound);
}

// bookkeeping around the idle re-find jobs, one lane per stream: what 0 before the failure-queue job draws the job (the reference's
// rand() % 20 == 0, :112, as "every 20th time the condition is evaluated"); 1 after ReFindNewlyMade empties the new queue;
// 2 after ReFindFromFailureQueue empties the failure queue.
__global__ void k_idle_gate(MapDev m, int S, int what) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  TrackerState* st = &m.st[s];
  if (!st->map_good) return;
  if (what == 0) {
    int go = 0;
    if (st->ba_converged_recent && st->ba_converged_full) { go = (st->idle_count % 20) == 0; st->idle_count++; }
    st->idle_do_fail = go;
  } else if (what == 1) {
    if (st->ba_converged_recent) st->newq_head = st->n_points;
  } else if (st->idle_do_fail) { st->fq_n = 0; st->idle_do_fail = 0; }
}

static bool keeps_kf_corners(const vslam_system* sys) { return sys->p.grow_map != 0 || sys->p.idle_iterations != 0; }

int grow_alloc(vslam_system* sys) {
  sys->map.never_retry = nullptr; sys->map.fq = nullptr;
  if (!keeps_kf_corners(sys)) return VSLAM_OK;
  if (sys->p.idle_iterations != 0) {
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, (size_t)sys->S * sys->p.max_points * 2 * sizeof(unsigned long long) + 64));
    HIPCHK(hipMemsetAsync(q, 0, (size_t)sys->S * sys->p.max_points * 2 * sizeof(unsigned long long) + 64, sys->stream));
    sys->allocs.push_back(q); sys->map.never_retry = (unsigned long long*)q;
    q = nullptr;
    HIPCHK(hipMalloc(&q, (size_t)sys->S * sys->tp.fq_cap * sizeof(int2) + 64));
    sys->allocs.push_back(q); sys->map.fq = (int2*)q;
  }
  const size_t S = sys->S, K = sys->p.max_keyframes;
  for (int l = 0; l < NLEV; l++) {
    void* ptr = nullptr;
    HIPCHK(hipMalloc(&ptr, S * K * (size_t)sys->tp.kcap[l] * sizeof(uint32_t) + 64));
    sys->allocs.push_back(ptr);
    sys->map.kf_corners[l] = (uint32_t*)ptr;
  }
  void* ptr = nullptr;
  HIPCHK(hipMalloc(&ptr, S * K * NLEV * sizeof(int) + 64));
  HIPCHK(hipMemsetAsync(ptr, 0, S * K * NLEV * sizeof(int) + 64, sys->stream));
  sys->allocs.push_back(ptr);
  sys->map.kf_ncorners = (int*)ptr;
  ptr = nullptr;
  HIPCHK(hipMalloc(&ptr, S * (size_t)sys->tp.kcap[0] * 2 * sizeof(double) + 64));
  sys->allocs.push_back(ptr);
  sys->grow_implane = (double*)ptr;
  return VSLAM_OK;
}

static void grow_args(vslam_system* sys, GrowArgs& a);

int grow_copy_corners(vslam_system* sys) {
  if (!keeps_kf_corners(sys)) return VSLAM_OK;
  const LevelGeom* g = sys->geom;
  hipLaunchKernelGGL(k_copy_kf_corners, dim3(NLEV, sys->S), dim3(256), 0, sys->stream, sys->map, sys->tp, sys->fr.corners[0], sys->fr.corners[1],
                     sys->fr.corners[2], sys->fr.corners[3], sys->fr.ncorners, g[0].cap, g[1].cap, g[2].cap, g[3].cap);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

int grow_levels(vslam_system* sys, const int* order, int n) {
  GrowArgs a;
  grow_args(sys, a);
  for (int i = 0; i < n; i++) {
    const int r = fe_thin_new_keyframe(sys, order[i]);
    if (r) return r;
    hipLaunchKernelGGL(k_target_implane, dim3(8, sys->S), dim3(256), 0, sys->stream, sys->map, sys->tp, a, order[i]);
    if (sys->tp.P == 8) hipLaunchKernelGGL(k_epipolar<8>, dim3(sys->S), dim3(GROW_THREADS), 0, sys->stream, sys->map, sys->tp, a, order[i]);
    else hipLaunchKernelGGL(k_epipolar<11>, dim3(sys->S), dim3(GROW_THREADS), 0, sys->stream, sys->map, sys->tp, a, order[i]);
  }
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

int grow_on_keyframe(vslam_system* sys) {
  if (!keeps_kf_corners(sys)) return VSLAM_OK;
  int r = grow_copy_corners(sys);
  if (r) return r;
  GrowArgs a;
  grow_args(sys, a);
  if (sys->p.grow_map & 2) {                                             // ReFindInSingleKeyFrame(*pK), jni/MapMaker.cc:497
    if (sys->tp.P == 8) hipLaunchKernelGGL(k_refind<8>, dim3(REFIND_BLOCKS, sys->S), dim3(GROW_THREADS), 0, sys->stream, sys->map, sys->tp, a);
    else hipLaunchKernelGGL(k_refind<11>, dim3(REFIND_BLOCKS, sys->S), dim3(GROW_THREADS), 0, sys->stream, sys->map, sys->tp, a);
  }
  if (!(sys->p.grow_map & 1)) { HIPCHK(hipGetLastError()); return VSLAM_OK; }
  r = fe_keyframe_rest_gated(sys);                                       // pK->MakeKeyFrame_Rest(), jni/MapMaker.cc:488
  if (r) return r;
  const int order[NLEV] = {3, 0, 1, 2};                                   // AddSomeMapPoints(3); (0); (1); (2), :498-501
  return grow_levels(sys, order, NLEV);
}

static void grow_args(vslam_system* sys, GrowArgs& a) {
  const LevelGeom* g = sys->geom;
  for (int l = 0; l < NLEV; l++) {
    a.cand[l] = sys->cand[l]; a.cand_score[l] = sys->cand_score[l]; a.cap[l] = g[l].cap; a.w[l] = g[l].w; a.h[l] = g[l].h; a.kf_pitch[l] = g[l].pitch;
    a.kf_stride[l] = (size_t)g[l].pitch * g[l].h; a.rowlut[l] = sys->fr.rowlut[l];
  }
  a.ncand = sys->ncand;
  a.tgt_implane = sys->grow_implane; a.tgt_cap = sys->tp.kcap[0];
}

// ReFindNewlyMade (mode 0) / ReFindFromFailureQueue (mode 1) of the idle pass, with the queue bookkeeping around them
int grow_idle_refind(vslam_system* sys, int mode) {
  GrowArgs a;
  grow_args(sys, a);
  const dim3 gs((sys->S + 63) / 64), bs(64);
  if (mode == 1) hipLaunchKernelGGL(k_idle_gate, gs, bs, 0, sys->stream, sys->map, sys->S, 0);
  if (sys->tp.P == 8) hipLaunchKernelGGL(k_refind_idle<8>, dim3(REFIND_BLOCKS, sys->S), dim3(GROW_THREADS), 0, sys->stream, sys->map, sys->tp, a, mode);
  else hipLaunchKernelGGL(k_refind_idle<11>, dim3(REFIND_BLOCKS, sys->S), dim3(GROW_THREADS), 0, sys->stream, sys->map, sys->tp, a, mode);
  hipLaunchKernelGGL(k_idle_gate, gs, bs, 0, sys->stream, sys->map, sys->S, mode == 0 ? 1 : 2);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

extern "C" int vslam_get_keyframe_corners(vslam_system* sys, int stream, int keyframe, int level, uint32_t* corners, int cap, int* n) {
  if (!sys || stream < 0 || stream >= sys->S || keyframe < 0 || keyframe >= sys->p.max_keyframes || level < 0 || level >= NLEV) { vslam_set_error("get_keyframe_corners: bad argument"); return VSLAM_E_INVALID; }
  if (!keeps_kf_corners(sys)) { vslam_set_error("get_keyframe_corners: keyframe corner lists are only kept with grow_map or idle_iterations"); return VSLAM_E_STATE; }
  HIPCHK(hipStreamSynchronize(sys->stream));
  const size_t slot = (size_t)stream * sys->p.max_keyframes + keyframe;
  int cnt = 0;
  HIPCHK(hipMemcpy(&cnt, sys->map.kf_ncorners + slot * NLEV + level, sizeof(int), hipMemcpyDeviceToHost));
  if (n) *n = cnt;
  const int m = cnt < cap ? cnt : cap;
  if (corners && m > 0) HIPCHK(hipMemcpy(corners, sys->map.kf_corners[level] + slot * sys->tp.kcap[level], (size_t)m * 4, hipMemcpyDeviceToHost));
  return VSLAM_OK;
}

#ifdef VSLAM_BA_PROF
extern "C" int vslam_debug_grow_prof(unsigned long long* out16, int reset) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_grow_prof), sizeof(unsigned long long) * 16));
  if (reset) { unsigned long long z[16] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_grow_prof), z, sizeof(z))); }
  return VSLAM_OK;
}
#endif
