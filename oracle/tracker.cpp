// ORACLE (test infrastructure only).  jni/Tracker.cc + jni/TrackerData.h + jni/KeyFrame.cc restated.
// Differences from the reference that are part of the build's contract:
//   * random_shuffle (jni/Tracker.cc:397,525) is the identity permutation;
//   * SmallBlurryImage / CalcSBIRotation (a "next" row) is off: gvnUseSBI = 0 path of ApplyMotionModel;
//   * the relocaliser is out of scope: a lost tracker keeps its pose and reports quality BAD.
#include "ptam_system.hpp"

namespace orc {

void make_keyframe_lite(KeyFrame& k, const uint8_t* gray, int w, int h, int stride, const int thr[4]) {
  // jni/KeyFrame.cc:5-51
  for (int l = 0; l < 4; l++) { k.w[l] = w >> l; k.h[l] = h >> l; k.im[l].assign((size_t)k.w[l] * k.h[l], 0); k.maxcorners[l].clear(); }
  for (int y = 0; y < h; y++) memcpy(k.im[0].data() + (size_t)y * w, gray + (size_t)y * stride, w);
  for (int l = 0; l < 4; l++) {
    if (l) orc_halfsample(k.im[l - 1].data(), k.w[l - 1], k.h[l - 1], k.w[l - 1], k.im[l].data(), k.w[l]);
    const int cap = k.w[l] * k.h[l];
    k.corners[l].resize(cap);
    const int n = orc_fast10(k.im[l].data(), k.w[l], k.h[l], k.w[l], thr[l], k.corners[l].data(), cap);
    k.corners[l].resize(n);
    k.lut[l].resize(k.h[l]);
    orc_row_lut(k.corners[l].data(), n, k.h[l], k.lut[l].data());
  }
}

void make_keyframe_rest_nonmax(KeyFrame& k, int barrier, bool quirk) {
  // jni/KeyFrame.cc:53-63 (non-max part of MakeKeyFrame_Rest; Shi-Tomasi candidates are a "next" row)
  for (int l = 0; l < 4; l++) {
    const int n = (int)k.corners[l].size();
    std::vector<int> sc(n);
    orc_fast_score(k.im[l].data(), k.w[l], k.h[l], k.w[l], k.corners[l].data(), n, barrier, sc.data());
    k.maxcorners[l].resize(n ? n : 1);
    const int m = orc_nonmax(k.corners[l].data(), sc.data(), n, quirk, k.maxcorners[l].data());
    k.maxcorners[l].resize(m);
  }
}

System::System(const Params& pp) : p(pp) {
  camera.init(p.cam, p.width, p.height, (p.quirks & ORC_Q_CAM_INT_RADIUS) != 0);
  for (int i = 0; i < 4; i++) attempted[i] = found[i] = 0;
  cur.depth_mean = 1.0; cur.depth_sigma = 1.0;   // jni/Tracker.cc:53-54
}
System::~System() { for (auto k : kfs) delete k; for (auto q : pts) delete q; delete pending; }

int System::AddKeyFrameRaw(const double pose12[12], bool fixed, const uint8_t* gray, int stride, double dmean, double dsigma) {
  KeyFrame* k = new KeyFrame;
  for (int i = 0; i < 9; i++) k->pose.R[i] = pose12[i];
  for (int i = 0; i < 3; i++) k->pose.t[i] = pose12[9 + i];
  k->fixed = fixed; k->depth_mean = dmean; k->depth_sigma = dsigma;
  make_keyframe_lite(*k, gray, p.width, p.height, stride, p.thr);
  make_keyframe_rest_nonmax(*k, p.nonmax_barrier, (p.quirks & ORC_Q_NONMAX_RIGHT_NEIGHBOUR) != 0);
  kfs.push_back(k);
  return (int)kfs.size() - 1;
}

int System::AddPointRaw(const double pos[3], int src_kf, int src_level, int irx, int iry, const double right[3], const double down[3]) {
  MapPoint* m = new MapPoint;
  m->pos = v3(pos[0], pos[1], pos[2]);
  m->src_kf = src_kf; m->src_level = src_level; m->irx = irx; m->iry = iry;
  m->pix_right = v3(right[0], right[1], right[2]); m->pix_down = v3(down[0], down[1], down[2]);
  m->finder.P = p.patch_size;
  m->finder.max_ssd = p.patch_size * p.patch_size * 500;   // jni/PatchFinder.cc:19-20
  m->finder.tmpl.assign(p.patch_size * p.patch_size, 0);
  pts.push_back(m);
  return (int)pts.size() - 1;
}

void System::AddMeasRaw(int kf, int pt, int level, const double root[2], bool subpix, int source) {
  Measurement m; m.level = level; m.subpix = subpix; m.root[0] = root[0]; m.root[1] = root[1]; m.source = source;
  kfs[kf]->meas[pt] = m;
  pts[pt]->meas_kfs.insert(kf);
}

void System::SetMapGood() {
  map_good = true;
  wiggle_depth_norm = p.wiggle_scale / kfs[0]->depth_mean;   // jni/MapMaker.cc:353
}

// ---- TrackerData (jni/TrackerData.h) ---------------------------------------------------------------------------
static void td_project(MapPoint& td, const SE3& pose, const Camera& cam, Camera::Proj& pr, bool& projected) {
  // TrackerData::Project, :69-87
  projected = false;
  td.in_image = td.pot_visible = false;
  td.cam = xform(pose, td.pos);
  if (td.cam[2] < 0.001) return;
  td.implane[0] = td.cam[0] / td.cam[2]; td.implane[1] = td.cam[1] / td.cam[2];
  if (td.implane[0] * td.implane[0] + td.implane[1] * td.implane[1] > cam.largest_radius * cam.largest_radius) return;
  pr = cam.project(td.implane[0], td.implane[1]);
  projected = true;
  td.image[0] = pr.im[0]; td.image[1] = pr.im[1];
  if (pr.invalid) return;
  if (td.image[0] < 0 || td.image[1] < 0 || td.image[0] > cam.size[0] || td.image[1] > cam.size[1]) return;
  td.in_image = true;
}

// TrackerData::ProjectAndDerivs, :98-102.  GetDerivsUnsafe reads the camera's cached last projection.  When Project()
// returns before calling Cam.Project (point behind the camera / beyond the largest radius) the reference would read the
// cache left by whichever point was projected before it (order-dependent artefact); the build keeps the point's own
// previous derivatives instead (deliberate deviation).
static void td_project_and_derivs(MapPoint& td, const SE3& pose, const Camera& cam) {
  Camera::Proj pr; bool projected;
  td_project(td, pose, cam, pr, projected);
  if (td.found && projected) cam.derivs(pr, td.derivs);
}

static void td_calc_jacobian(MapPoint& td) {
  // TrackerData::CalcJacobian, :107-122
  const double ooz = 1.0 / td.cam[2];
  const double pos[4] = {td.cam[0], td.cam[1], td.cam[2], 1.0};
  for (int m = 0; m < 6; m++) {
    double mot[4];
    generator_field(m, pos, mot);
    const double f0 = (mot[0] - td.cam[0] * mot[2] * ooz) * ooz;
    const double f1 = (mot[1] - td.cam[1] * mot[2] * ooz) * ooz;
    td.jac[0 * 6 + m] = td.derivs[0] * f0 + td.derivs[1] * f1;
    td.jac[1 * 6 + m] = td.derivs[2] * f0 + td.derivs[3] * f1;
  }
}

static void td_linear_update(MapPoint& td, const double v6[6]) {
  // TrackerData::LinearUpdate, :125-131
  double a = 0, b = 0;
  for (int m = 0; m < 6; m++) { a += td.jac[m] * v6[m]; b += td.jac[6 + m] * v6[m]; }
  td.image[0] += a; td.image[1] += b;
}

// ---- Tracker -----------------------------------------------------------------------------------------------------
void System::TrackFrame(const uint8_t* gray, int stride) {
  // jni/Tracker.cc:76-146
  FrameBegin(gray, stride);
  if (tracked_this_frame) TrackMap();
  FrameEnd();
}

// TrackFrame in pieces (the stage entry points of the C ABI, vslam_patch_search / vslam_pose_update, are compared against these):
// FrameBegin = :76-105 + ApplyMotionModel; TrackMap = SearchStage(0), PoseStage(0), SearchStage(1), PoseStage(1);
// FrameEnd = UpdateMotionModel, AssessTrackingQuality and the keyframe decision (:107-132).
void System::FrameBegin(const uint8_t* gray, int stride) {
  kf_added_this_frame = false;
  if (pending && --pending->countdown == 0) { ApplyBundle(*pending); delete pending; pending = nullptr; HandleBadPoints(); }   // deferred map-maker results
  cur.meas.clear();
  make_keyframe_lite(cur, gray, p.width, p.height, stride, p.thr);
  if (p.use_sbi) {                           // :86-97: the small images of the rotation estimator
    if (!have_sbi) {
      sbi_make(sbi_this, cur.im[3].data(), cur.w[3], cur.h[3], 0.75);
      sbi_last = sbi_this;
      have_sbi = true;
    } else {
      sbi_last = sbi_this;
      sbi_make(sbi_this, cur.im[3].data(), cur.w[3], cur.h[3], 0.75);
    }
  }
  frame++;
  if (Tracking()) {
    if (p.use_sbi) calc_sbi_rotation(sbi_this, sbi_last, camera, (p.quirks & ORC_Q_CAM_INT_RADIUS) != 0, sbi_rot, &sbi_score);   // :104-105
    ApplyMotionModel();
  }
  tracked_this_frame = Tracking();
  if (!map_good) TrackForInitialMap();         // jni/Tracker.cc:144-145: no map yet, try to make one
}

void System::FrameEnd() {
  if (tracked_this_frame) TrackerFrameEnd();
  for (int i = 0; i < p.idle_iterations; i++) IdleIteration();   // the map-maker's idle jobs, a fixed number of run() iterations per frame
}

void System::TrackerFrameEnd() {
  UpdateMotionModel();
  AssessTrackingQuality();
  if (quality == 2 && NeedNewKeyFrame() && frame - last_kf_dropped > p.min_frames_between_kf) {   // :128-132
    AddKeyFrame();                     // Tracker::AddNewKeyFrame :823-827
    last_kf_dropped = frame;
    kf_added_this_frame = true;
  }
}

void System::ApplyMotionModel() {
  // jni/Tracker.cc:781-798
  start_pose = pose;
  double v[6];
  for (int i = 0; i < 6; i++) v[i] = velocity[i];
  if (p.use_sbi) { v[0] = 0.0; v[1] = 0.0; v[3] = sbi_rot[3]; v[4] = sbi_rot[4]; v[5] = sbi_rot[5]; }   // :788-794
  pose = mul(se3_exp(v), start_pose);
}

void System::UpdateMotionModel() {
  // jni/Tracker.cc:802-820
  const SE3 nfo = mul(pose, inverse(start_pose));
  double motion[6];
  se3_ln(nfo, motion);
  for (int i = 0; i < 6; i++) velocity[i] = 0.9 * (0.5 * motion[i] + 0.5 * velocity[i]);
  double v[6];
  for (int i = 0; i < 6; i++) v[i] = velocity[i];
  for (int i = 0; i < 3; i++) v[i] *= 1.0 / cur.depth_mean;
  double s = 0; for (int i = 0; i < 6; i++) s += v[i] * v[i];
  msd_vel = sqrt(s);
}

void System::AssessTrackingQuality() {
  // jni/Tracker.cc:832-878
  int ta = 0, tf = 0, la = 0, lf = 0;
  for (int i = 0; i < 4; i++) { ta += attempted[i]; tf += found[i]; if (i >= 2) { la += attempted[i]; lf += found[i]; } }
  if (tf == 0 || ta == 0) quality = 0;
  else {
    const double dTotal = (double)tf / ta;
    const double dLarge = la > 10 ? (double)lf / la : dTotal;
    if (dTotal > 0.3) quality = 2;
    else if (dLarge < 0.13) quality = 0;
    else quality = 1;
  }
  if (quality == 1) {  // IsDistanceToNearestKeyFrameExcessive, jni/MapMaker.cc:1098-1101
    double best = 9999999999.9;
    for (auto k : kfs) best = std::min(best, KeyFrameLinearDist(cur.pose, k->pose));
    if (best > p.wiggle_scale * 10.0) quality = 0;
  }
  if (quality == 0) lost_frames++; else lost_frames = 0;
}

int System::SearchForPoints(std::vector<int>& vTD, int nRange, int nSubPixIts) {
  // jni/Tracker.cc:629-674
  int nFound = 0;
  for (int idx : vTD) {
    MapPoint& TD = *pts[idx];
    Finder& F = TD.finder;
    finder_make_template(F, TD, *kfs[TD.src_kf]);
    if (F.bad) { TD.in_image = TD.pot_visible = TD.found = false; continue; }
    attempted[F.level]++;
    const long before = F.n_zmssd;
    const bool bFound = finder_find_coarse(F, TD.image, cur, (unsigned)nRange);
    n_zmssd += F.n_zmssd - before;
    TD.searched = true;
    if (!bFound) { TD.found = false; continue; }
    TD.found = true;
    TD.sqrt_inv_noise = 1.0 / level_scale(F.level);
    nFound++;
    found[F.level]++;
    if (nSubPixIts > 0) {
      TD.did_subpix = true;
      finder_make_subpix(F);
      if (!finder_iterate_subpix_to_convergence(F, cur, nSubPixIts)) {
        TD.found = false; nFound--; found[F.level]--;
        continue;
      }
      TD.vfound[0] = F.subpix[0]; TD.vfound[1] = F.subpix[1];
    } else {
      TD.vfound[0] = F.coarse[0]; TD.vfound[1] = F.coarse[1];
      TD.did_subpix = false;
    }
  }
  return nFound;
}

void System::CalcPoseUpdate(const std::vector<int>& vTD, double dOverrideSigma, bool bMarkOutliers, double out[6]) {
  // jni/Tracker.cc:683-774 (Tukey)
  std::vector<double> e2;
  for (int idx : vTD) {
    MapPoint& TD = *pts[idx];
    if (!TD.found) continue;
    TD.err_cov[0] = (TD.vfound[0] - TD.image[0]) * TD.sqrt_inv_noise;
    TD.err_cov[1] = (TD.vfound[1] - TD.image[1]) * TD.sqrt_inv_noise;
    e2.push_back(TD.err_cov[0] * TD.err_cov[0] + TD.err_cov[1] * TD.err_cov[1]);
  }
  if (e2.empty()) { for (int i = 0; i < 6; i++) out[i] = 0; return; }
  const double sigma2 = dOverrideSigma > 0 ? dOverrideSigma : find_sigma_squared(EST_TUKEY, e2);
  WLS6 wls;
  wls.add_prior(p.wls_prior);
  const bool qint = (p.quirks & ORC_Q_POSE_INT_RESIDUAL) != 0;
  for (int idx : vTD) {
    MapPoint& TD = *pts[idx];
    if (!TD.found) continue;
    const double es = TD.err_cov[0] * TD.err_cov[0] + TD.err_cov[1] * TD.err_cov[1];
    const double wgt = weight(EST_TUKEY, es, sigma2);
    if (wgt == 0.0) { if (bMarkOutliers) TD.n_outlier++; continue; }
    else if (bMarkOutliers) TD.n_inlier++;
    double j1[6], j2[6];
    for (int m = 0; m < 6; m++) { j1[m] = TD.sqrt_inv_noise * TD.jac[m]; j2[m] = TD.sqrt_inv_noise * TD.jac[6 + m]; }
    // :766-767 casts the residual to int (quirk #6); PTAM passes the double
    wls.add_mJ(qint ? (double)(int)TD.err_cov[0] : TD.err_cov[0], j1, wgt);
    wls.add_mJ(qint ? (double)(int)TD.err_cov[1] : TD.err_cov[1], j2, wgt);
  }
  wls.compute(out);
}

void System::TrackMap() {
  // jni/Tracker.cc:358-626
  SearchStage(0); PoseStage(0); SearchStage(1); PoseStage(1);
}

void System::SearchStage(int stage) {
  if (stage == 0) {
    for (int i = 0; i < 4; i++) attempted[i] = found[i] = 0;
    for (int l = 0; l < 4; l++) tm_pvs[l].clear();
    std::vector<int>* avPVS = tm_pvs;
    for (size_t i = 0; i < pts.size(); i++) {         // :369-392
      MapPoint& TD = *pts[i];
      if (TD.bad) continue;                           // bad points live in the trash list (jni/Map.cc:16-27)
      Camera::Proj pr; bool projected;
      td_project(TD, pose, camera, pr, projected);
      if (!TD.in_image) continue;
      camera.derivs(pr, TD.derivs);                   // GetDerivsUnsafe
      TD.search_level = finder_calc_level_and_warp(TD.finder, TD, pose, TD.derivs);
      if (TD.search_level == -1) continue;
      TD.searched = false; TD.found = false;
      avPVS[TD.search_level].push_back((int)i);
    }
    // :396-397 random_shuffle -> identity permutation
    tm_next.clear(); tm_iter.clear();
    std::vector<int>& vNext = tm_next;
    unsigned nCoarseMax = p.coarse_max, nCoarseRange = p.coarse_range;
    did_coarse = false;
    tm_coarse_tried = false; tm_coarse_found = 0;
    bool bTryCoarse = true;
    if (p.coarse_disabled || msd_vel < p.coarse_min_vel || nCoarseMax == 0) bTryCoarse = false;
    if (just_recovered) { bTryCoarse = true; nCoarseMax *= 2; nCoarseRange *= 2; just_recovered = false; }
    if (bTryCoarse && avPVS[3].size() + avPVS[2].size() > (unsigned)p.coarse_min) {   // :437-491
      if (avPVS[3].size() <= nCoarseMax) { vNext = avPVS[3]; avPVS[3].clear(); }
      else {
        for (unsigned i = 0; i < nCoarseMax; i++) vNext.push_back(avPVS[3][i]);
        avPVS[3].erase(avPVS[3].begin(), avPVS[3].begin() + nCoarseMax);
      }
      if (vNext.size() < nCoarseMax) {
        const unsigned more = nCoarseMax - vNext.size();
        if (avPVS[2].size() <= more) { vNext = avPVS[2]; avPVS[2].clear(); }   // :454-456 replaces, not appends (PTAM bug kept)
        else {
          for (unsigned i = 0; i < more; i++) vNext.push_back(avPVS[2][i]);
          avPVS[2].erase(avPVS[2].begin(), avPVS[2].begin() + more);
        }
      }
      tm_coarse_found = SearchForPoints(vNext, nCoarseRange, p.coarse_subpix_its);
      tm_iter = vNext;
      tm_coarse_tried = true;
    }
    return;
  }
  std::vector<int>* avPVS = tm_pvs;
  std::vector<int>& vNext = tm_next;
  std::vector<int>& vIter = tm_iter;
  const int nFineRange = did_coarse ? 5 : 10;   // :495-497
  {
    const int l = 3;                      // :501-508
    for (int i : avPVS[l]) td_project_and_derivs(*pts[i], pose, camera);
    SearchForPoints(avPVS[l], nFineRange, p.fine_subpix_its);
    for (int i : avPVS[l]) vIter.push_back(i);
  }
  vNext.clear();
  for (int l = 2; l >= 0; l--) for (int i : avPVS[l]) vNext.push_back(i);
  int nFinePatchesToUse = p.max_patches - (int)vIter.size();   // :518-526
  if (nFinePatchesToUse < 0) nFinePatchesToUse = 0;
  if ((int)vNext.size() > nFinePatchesToUse) vNext.resize(nFinePatchesToUse);
  if (did_coarse) for (int i : vNext) td_project_and_derivs(*pts[i], pose, camera);
  SearchForPoints(vNext, nFineRange, 0);
  for (int i : vNext) vIter.push_back(i);
}

void System::PoseStage(int stage) {
  std::vector<int>& vIter = tm_iter;
  if (stage == 0) {
    if (!tm_coarse_tried || tm_coarse_found < (unsigned)p.coarse_min) return;   // :465
    did_coarse = true;
    for (int iter = 0; iter < 10; iter++) {
      if (iter != 0) for (int i : vIter) if (pts[i]->found) td_project_and_derivs(*pts[i], pose, camera);
      for (int i : vIter) if (pts[i]->found) td_calc_jacobian(*pts[i]);
      double up[6];
      CalcPoseUpdate(vIter, iter > 5 ? 1.0 : 0.0, false, up);
      pose = mul(se3_exp(up), pose);
    }
    return;
  }
  double last_up[6] = {0, 0, 0, 0, 0, 0};
  for (int iter = 0; iter < 10; iter++) {   // :543-577
    const bool nonlinear = (iter == 0 || iter == 4 || iter == 9);
    if (iter != 0) {
      if (nonlinear) { for (int i : vIter) if (pts[i]->found) td_project_and_derivs(*pts[i], pose, camera); }
      else { for (int i : vIter) if (pts[i]->found) td_linear_update(*pts[i], last_up); }
    }
    if (nonlinear) for (int i : vIter) if (pts[i]->found) td_calc_jacobian(*pts[i]);
    double up[6];
    CalcPoseUpdate(vIter, iter > 5 ? 16.0 : 0.0, iter == 9, up);
    pose = mul(se3_exp(up), pose);
    for (int i = 0; i < 6; i++) last_up[i] = up[i];
  }
  cur.pose = pose;                          // :594
  cur.meas.clear();                         // :597-607
  for (int i : vIter) {
    MapPoint& TD = *pts[i];
    if (!TD.found) continue;
    Measurement m; m.root[0] = TD.vfound[0]; m.root[1] = TD.vfound[1]; m.level = TD.search_level; m.subpix = TD.did_subpix; m.source = SRC_TRACKER;
    cur.meas[i] = m;
  }
  {                                         // :610-625
    double dSum = 0, dSumSq = 0; int nNum = 0;
    for (int i : vIter) if (pts[i]->found) { const double z = pts[i]->cam[2]; dSum += z; dSumSq += z * z; nNum++; }
    if (nNum > 20) {
      cur.depth_mean = dSum / nNum;
      cur.depth_sigma = sqrt((dSumSq / nNum) - (cur.depth_mean) * (cur.depth_mean));
    }
  }
  iteration_set = vIter;
}

}  // namespace orc
