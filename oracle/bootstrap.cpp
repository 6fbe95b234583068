// ORACLE (test infrastructure only -- see ptam_oracle.h).  The map bootstrap: the tracker's trail
// tracking (jni/Tracker.cc:247-346), HomographyInit::Compute (jni/HomographyInit.cc:43-71), MapMaker::InitFromStereo
// (jni/MapMaker.cc:204-376), CalcPlaneAligner / ApplyGlobalTransformationToMap (:1104-1231, :440-449), as sequential loops in
// the reference's order.  The mathematics of HomographyInit and CalcPlaneAligner is the oracle's own restatement in homography.cpp
// (independent of the product's csrc/bootstrap_math.h; only the definition of the random draws is the same on both sides -- see
// that file's header): PARITY UNPINNED (no fixture of the reference covers it).
#include <algorithm>
#include <cstring>
#include "ptam_system.hpp"

namespace orc {

V3 unit_ray(const Camera& cam, double ix, double iy);
void refresh_pixel_vectors(MapPoint& p, const KeyFrame& k, const V3& center, const V3& one_right, const V3& one_down);

// ATANCamera::UnProject followed by GetProjectionDerivs (jni/ATANCamera.cc:149-164, 198-231): the derivatives at the state
// the un-projection leaves
static void unproject_with_derivs(const Camera& c, double ix, double iy, double out[2], double jac[4]) {
  const double dx = (ix - c.center[0]) * c.inv_focal[0], dy = (iy - c.center[1]) * c.inv_focal[1];
  const double dist_r = sqrt(dx * dx + dy * dy);
  const double rr = c.invrtrans(dist_r);
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

// ---- Tracker::TrackForInitialMap and the trails, jni/Tracker.cc:247-346 -------------------------------------------------------
void System::TrackForInitialMap() {
  const int max_ssd = 100000;                                                     // MiniPatchMaxSSD, :249
  if (init_stage == 0) {
    if (spacebar) { spacebar = false; TrailTrackingStart(); init_stage = 1; }
    return;
  }
  if (init_stage == 1) {
    const int good = TrailTrackingAdvance(max_ssd);
    if (good < 10) { trails.clear(); init_stage = 0; return; }                    // Reset(), :266-269
    if (spacebar) {
      spacebar = false;
      std::vector<std::array<int, 4>> matches;
      for (auto& t : trails) matches.push_back({t.init[0], t.init[1], t.cur[0], t.cur[1]});
      init_ok = InitFromStereo(first_kf, cur, matches);
      init_stage = 2;
    }
  }
}

void System::TrailTrackingStart() {
  // :290-318.  The functor of the sort compares with >, on the NEGATED scores: the lowest Shi-Tomasi scores come first (the
  // reference's behaviour, kept); std::sort leaves the order of equal scores open -- here they keep the candidate order.
  make_keyframe_rest_nonmax(cur, p.nonmax_barrier, (p.quirks & ORC_Q_NONMAX_RIGHT_NEIGHBOUR) != 0);
  make_keyframe_rest_candidates(cur, 70.0);
  first_kf = cur;
  std::vector<std::pair<double, uint32_t>> v;
  const int w = cur.w[0], h = cur.h[0];
  auto inb = [&](int x, int y) { return x >= 4 && y >= 4 && x < w - 4 && y < h - 4; };
  for (size_t i = 0; i < cur.cand[0].size(); i++) {
    const int x = cur.cand[0][i] & 0xFFFF, y = cur.cand[0][i] >> 16;
    if (!inb(x, y)) continue;
    v.push_back({-1.0 * cur.cand_score[0][i], cur.cand[0][i]});
  }
  std::stable_sort(v.begin(), v.end(), [](const std::pair<double, uint32_t>& a, const std::pair<double, uint32_t>& b) { return a.first > b.first; });
  trails.clear();
  int to_add = 1000;
  for (size_t i = 0; i < v.size() && to_add > 0; i++) {
    Trail t;
    const int x = v[i].second & 0xFFFF, y = v[i].second >> 16;
    orc_minipatch_sample(cur.im[0].data(), w, h, w, x, y, t.patch);
    t.init[0] = t.cur[0] = x; t.init[1] = t.cur[1] = y;
    trails.push_back(t);
    to_add--;
  }
  prev_kf = first_kf;
}

int System::TrailTrackingAdvance(int max_ssd) {
  // :321-346: forward search at the FAST corners, then the backward ("married matches") check against the previous frame
  int good = 0;
  const int w = cur.w[0], h = cur.h[0];
  std::vector<Trail> keep;
  for (auto& t : trails) {
    const int start[2] = {t.cur[0], t.cur[1]};
    int end[2] = {start[0], start[1]};
    bool found = orc_minipatch_find(t.patch, cur.im[0].data(), w, h, w, cur.corners[0].data(), (int)cur.corners[0].size(), 10, max_ssd, end) != 0;
    if (found) {
      uint8_t back[81];
      orc_minipatch_sample(cur.im[0].data(), w, h, w, end[0], end[1], back);
      int bp[2] = {end[0], end[1]};
      found = orc_minipatch_find(back, prev_kf.im[0].data(), w, h, w, prev_kf.corners[0].data(), (int)prev_kf.corners[0].size(), 10, max_ssd, bp) != 0;
      const int dx = bp[0] - start[0], dy = bp[1] - start[1];
      if (dx * dx + dy * dy > 2) found = false;
      t.cur[0] = end[0]; t.cur[1] = end[1];
      good++;                                                                     // counted before the backward check, as there
    }
    if (found) keep.push_back(t);
  }
  trails.swap(keep);
  prev_kf = cur;
  return good;
}

// ---- MapMaker::InitFromStereo, jni/MapMaker.cc:204-376 -----------------------------------------------------------------------
void System::RefreshSceneDepth(KeyFrame& k) {
  // :1236-1252
  double sum = 0.0, sumsq = 0.0; int n = 0;
  for (auto& it : k.meas) { const V3 c = xform(k.pose, pts[it.first]->pos); sum += c[2]; sumsq += c[2] * c[2]; n++; }
  k.depth_mean = sum / n;
  k.depth_sigma = sqrt((sumsq / n) - (k.depth_mean) * (k.depth_mean));
}

bool System::InitFromStereo(const KeyFrame& kF, const KeyFrame& kS, const std::vector<std::array<int, 4>>& trail) {
  std::vector<HMatch> vm;
  for (auto& t : trail) {                                                        // :210-229: the derivatives are those at the SECOND position
    HMatch m; double j0[4];
    unproject_with_derivs(camera, t[0], t[1], m.first, j0);
    unproject_with_derivs(camera, t[2], t[3], m.second, m.jac);
    vm.push_back(m);
  }
  SE3 se3;
  if (!homography_init_compute(vm, 5.0, boot_seed, se3, &n_hom_inliers)) return false;   // :233-240
  const double mag = sqrt(se3.t[0] * se3.t[0] + se3.t[1] * se3.t[1] + se3.t[2] * se3.t[2]);
  if (mag == 0) return false;                                                    // :243-248
  for (int i = 0; i < 3; i++) se3.t[i] *= p.wiggle_scale / mag;                  // :250
  KeyFrame* pkFirst = new KeyFrame(kF); KeyFrame* pkSecond = new KeyFrame(kS);
  pkFirst->fixed = true; pkFirst->pose = SE3(); pkFirst->meas.clear();
  pkSecond->fixed = false; pkSecond->pose = se3; pkSecond->meas.clear();
  kfs.push_back(pkFirst); kfs.push_back(pkSecond);                               // (pushed at :339-340; the indices are needed below)
  Finder f; f.P = p.patch_size;
  f.max_ssd = f.P * f.P * 625;                                                   // PatchFinder's default mnMaxSSD (jni/PatchFinder.cc:16), unused by the sub-pixel steps
  for (size_t i = 0; i < trail.size(); i++) {
    const int cx = trail[i][0], cy = trail[i][1];
    // MakeTemplateCoarseNoWarp (jni/PatchFinder.cc:130-142) at level 0 of the first keyframe
    const int P = f.P, half = P / 2;
    if (!(cx >= half + 1 && cy >= half + 1 && cx < pkFirst->w[0] - (half + 1) && cy < pkFirst->h[0] - (half + 1))) continue;   // mbTemplateBad -> the sub-pixel steps fail
    f.level = 0; f.bad = false;
    f.tmpl.resize(P * P);
    int sum = 0, sumsq = 0;
    for (int r = 0; r < P; r++) for (int c = 0; c < P; c++) { const int b = pkFirst->im[0][(size_t)(cy - half + r) * pkFirst->w[0] + (cx - half + c)]; f.tmpl[r * P + c] = (uint8_t)b; sum += b; sumsq += b * b; }
    f.tsum = sum; f.tsumsq = sumsq;
    f.coarse[0] = 0; f.coarse[1] = 0;
    finder_make_subpix(f);
    f.subpix[0] = trail[i][2]; f.subpix[1] = trail[i][3];                        // SetSubPixPos, :300
    if (!finder_iterate_subpix_to_convergence(f, *pkSecond, 10)) continue;       // :301-304
    double uB[2];
    camera.unproject(f.subpix[0], f.subpix[1], uB);
    const V3 wp = reproject_point(se3, uB, vm[i].first);                         // :309-311
    if (wp[2] < 0.0) continue;
    MapPoint* mp = new MapPoint();
    mp->pos = wp; mp->boot = true; mp->src_kf = 0; mp->src_level = 0; mp->irx = cx; mp->iry = cy;
    mp->finder.P = p.patch_size; mp->finder.max_ssd = f.max_ssd;
    // :271-292: the "right" neighbour is taken one pixel DOWN (0, 1) and the "down" neighbour one pixel RIGHT (1, 0) -- the reference's
    refresh_pixel_vectors(*mp, *pkFirst, unit_ray(camera, cx, cy), unit_ray(camera, cx + 0, cy + 1), unit_ray(camera, cx + 1, cy + 0));
    pts.push_back(mp);
    const int pid = (int)pts.size() - 1;
    Measurement m1; m1.level = 0; m1.source = SRC_ROOT; m1.root[0] = cx; m1.root[1] = cy; m1.subpix = true;
    pkFirst->meas[pid] = m1; mp->meas_kfs.insert(0);
    Measurement m2; m2.level = 0; m2.source = SRC_TRAIL; m2.root[0] = f.subpix[0]; m2.root[1] = f.subpix[1]; m2.subpix = true;
    pkSecond->meas[pid] = m2; mp->meas_kfs.insert(1);
  }
  n_init_points = (int)pts.size();
  for (KeyFrame* k : {pkFirst, pkSecond}) {                                      // MakeKeyFrame_Rest, :341-342
    make_keyframe_rest_nonmax(*k, p.nonmax_barrier, (p.quirks & ORC_Q_NONMAX_RIGHT_NEIGHBOUR) != 0);
    make_keyframe_rest_candidates(*k, 70.0);
  }
  map_good = true;                                                               // (BundleAdjust needs nothing of it; set for the helpers' guards)
  for (int i = 0; i < 5; i++) { BundleAdjustAll(); HandleBadPoints(); }          // :344-345
  RefreshSceneDepth(*pkFirst); RefreshSceneDepth(*pkSecond);                     // :349-350
  wiggle_depth_norm = p.wiggle_scale / pkFirst->depth_mean;
  // AddSomeMapPoints works from the newest keyframe (ksrc = size - 1 = the second one), :353-356
  AddSomeMapPoints(0); AddSomeMapPoints(3); AddSomeMapPoints(1); AddSomeMapPoints(2);
  ba_converged_full = false; ba_converged_recent = false;
  int guard = 0;
  while (!ba_converged_full && guard++ < 50) { BundleAdjustAll(); HandleBadPoints(); }   // :361-365 (bounded here)
  ApplyGlobalTransformationToMap(CalcPlaneAligner());                            // :368
  pose = pkSecond->pose; start_pose = pose;                                      // se3TrackerPose, :370
  return true;
}

// ---- CalcPlaneAligner / ApplyGlobalTransformationToMap ------------------------------------------------------------------------
SE3 System::CalcPlaneAligner() {
  std::vector<V3> pos;                                                           // every point of the map (bad ones live in the trash there)
  for (auto q : pts) pos.push_back(q->pos);
  SE3 T;
  calc_plane_aligner(pos, boot_seed + 1u, T);                                    // identity with fewer than ten points (:1107-1110)
  return T;
}

void System::ApplyGlobalTransformationToMap(const SE3& new_from_old) {
  // :440-449; RefreshPixelVectors (jni/MapPoint.cc:4-29) re-derives the pixel vectors from the (transformed) source keyframe
  const SE3 inv = inverse(new_from_old);
  for (auto k : kfs) k->pose = mul(k->pose, inv);
  for (auto q : pts) {
    q->pos = xform(new_from_old, q->pos);
    const KeyFrame& k = *kfs[q->src_kf];
    const int s = 1 << q->src_level;
    const double cx = level_zero_pos((double)q->irx, q->src_level), cy = level_zero_pos((double)q->iry, q->src_level);
    if (q->src_kf == 0 && q->boot) refresh_pixel_vectors(*q, k, unit_ray(camera, cx, cy), unit_ray(camera, cx, cy + 1), unit_ray(camera, cx + 1, cy));
    else refresh_pixel_vectors(*q, k, unit_ray(camera, cx, cy), unit_ray(camera, cx + s, cy), unit_ray(camera, cx, cy + s));
  }
}

}  // namespace orc

extern "C" int orc_homography_init(const double* m8, int n, double max_pixel_error, unsigned seed, double out12[12], int* n_inliers) {
  std::vector<orc::HMatch> m((size_t)n);
  for (int i = 0; i < n; i++) { for (int k = 0; k < 2; k++) { m[i].first[k] = m8[8 * i + k]; m[i].second[k] = m8[8 * i + 2 + k]; } for (int k = 0; k < 4; k++) m[i].jac[k] = m8[8 * i + 4 + k]; }
  orc::SE3 T;
  if (!orc::homography_init_compute(m, max_pixel_error, seed, T, n_inliers)) return 0;
  for (int i = 0; i < 9; i++) out12[i] = T.R[i];
  for (int i = 0; i < 3; i++) out12[9 + i] = T.t[i];
  return 1;
}

extern "C" int orc_calc_plane_aligner(const double* pos3, int n, unsigned seed, double out12[12]) {
  std::vector<orc::V3> pos((size_t)n);
  for (int i = 0; i < n; i++) pos[i] = orc::v3(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]);
  orc::SE3 T;
  if (!orc::calc_plane_aligner(pos, seed, T)) return 0;
  for (int i = 0; i < 9; i++) out12[i] = T.R[i];
  for (int i = 0; i < 3; i++) out12[9 + i] = T.t[i];
  return 1;
}
