// ORACLE (test infrastructure only).  BA-driver part of jni/MapMaker.cc restated
// (AddKeyFrame :470-506, KeyFrameLinearDist/NClosest/Closest :705-758, NeedNewKeyFrame :761-773,
//  BundleAdjustAll :776-798, BundleAdjustRecent :801-851, BundleAdjust :854-960).
// The reference orders its std::set<KeyFrame*> / std::map<MapPoint*,...> by heap address; the build defines the
// order as insertion index.  The map-maker runs synchronously (the reference's thread is disabled,
// jni/MapMaker.cc:56).  Map growth (ReFindInSingleKeyFrame, AddSomeMapPoints) is in mapgrow.cpp, behind Params::grow_map.
#include "ptam_system.hpp"

namespace orc {

double System::KeyFrameLinearDist(const SE3& a, const SE3& b) {
  // :705-712
  const SE3 ia = inverse(a), ib = inverse(b);
  const double d[3] = {ib.t[0] - ia.t[0], ib.t[1] - ia.t[1], ib.t[2] - ia.t[2]};
  return sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}

bool System::NeedNewKeyFrame() {
  // :761-773 with ClosestKeyFrame :737-758
  double best = 9999999999.9;
  for (auto k : kfs) { const double d = KeyFrameLinearDist(cur.pose, k->pose); if (d < best) best = d; }
  double dDist = best;
  dDist *= (1.0 / cur.depth_mean);
  return dDist > p.max_kf_dist_wiggle_mult * wiggle_depth_norm;
}

void System::AddKeyFrame() {
  // MapMaker::AddKeyFrame :470-478 (deep copy) + AddKeyFrameFromTopOfQueue :481-506 run inline
  KeyFrame* pK = new KeyFrame(cur);
  make_keyframe_rest_nonmax(*pK, p.nonmax_barrier, (p.quirks & ORC_Q_NONMAX_RIGHT_NEIGHBOUR) != 0);   // :488
  kfs.push_back(pK);
  const int kidx = (int)kfs.size() - 1;
  for (auto& it : pK->meas) { pts[it.first]->meas_kfs.insert(kidx); it.second.source = SRC_TRACKER; }   // :491-494
  if (p.grow_map & 2) n_refound = ReFindInSingleKeyFrame(kidx);                                            // :497
  if (p.grow_map & 1) {                                                                                   // AddSomeMapPoints, :498-501
    make_keyframe_rest_candidates(*pK, 70.0);                                                              // rest of MakeKeyFrame_Rest, :488
    n_points_added = AddSomeMapPoints(3) + AddSomeMapPoints(0) + AddSomeMapPoints(1) + AddSomeMapPoints(2);
  }
  ba_converged_full = false; ba_converged_recent = false;                                                   // :504-505
  // MapMaker::run :98-99: local bundle adjustment takes priority once the queue is empty
  defer_ba = true;
  const int r = BundleAdjustRecent();
  defer_ba = false;
  if (!pending) { last_ba_accepted = r; HandleBadPoints(); }                                                  // run() :117
}

void System::HandleBadPoints() {
  // :140-164.  Bad points stay in the arrays flagged `bad` (the reference moves them to a trash list, jni/Map.cc:16-27).
  for (auto q : pts) if (q->n_outlier > 20 && q->n_outlier > q->n_inlier) q->bad = true;
  for (size_t i = 0; i < pts.size(); i++) if (pts[i]->bad) for (auto k : kfs) k->meas.erase((int)i);
}

int System::BundleAdjustAll() {
  // :776-798
  std::vector<int> adj, fixed, points;
  for (size_t i = 0; i < kfs.size(); i++) (kfs[i]->fixed ? fixed : adj).push_back((int)i);
  for (size_t i = 0; i < pts.size(); i++) if (!pts[i]->bad) points.push_back((int)i);
  return BundleAdjust(adj, fixed, points, false);
}

int System::BundleAdjustRecent() {
  // :801-851
  if ((int)kfs.size() < p.ba_min_keyframes) { ba_converged_recent = true; return -2; }
  std::set<int> sAdjust;
  const int newest = (int)kfs.size() - 1;
  sAdjust.insert(newest);
  std::vector<std::pair<double, int>> v;                                  // NClosestKeyFrames :714-735
  for (int i = 0; i < (int)kfs.size(); i++) if (i != newest) v.push_back(std::make_pair(KeyFrameLinearDist(kfs[newest]->pose, kfs[i]->pose), i));
  unsigned N = p.ba_window - 1;
  if (N > v.size()) N = v.size();
  std::partial_sort(v.begin(), v.begin() + N, v.end());
  for (unsigned i = 0; i < N; i++) if (!kfs[v[i].second]->fixed) sAdjust.insert(v[i].second);
  std::set<int> sPoints;
  for (int k : sAdjust) for (auto& it : kfs[k]->meas) sPoints.insert(it.first);
  std::vector<int> fixed;
  for (int i = 0; i < (int)kfs.size(); i++) {
    if (sAdjust.count(i)) continue;
    bool inc = false;
    for (auto& it : kfs[i]->meas) if (sPoints.count(it.first)) { inc = true; break; }
    if (inc) fixed.push_back(i);
  }
  return BundleAdjust(std::vector<int>(sAdjust.begin(), sAdjust.end()), fixed, std::vector<int>(sPoints.begin(), sPoints.end()), true);
}

int System::BundleAdjust(const std::vector<int>& adj, const std::vector<int>& fixed, const std::vector<int>& points, bool recent) {
  // :854-960.  The Bundle is assembled from the map and computed now; its results are written back by ApplyBundle --
  // immediately, or (defer_ba, tracker-driven keyframes with ba_delay_frames = D > 0) at the start of the D-th following
  // frame, which models the reference's map-maker thread finishing a little later than the tracker (jni/MapMaker.cc:80-123).
  PendingBA* pb = new PendingBA;
  Bundle& b = pb->b;
  b.camera = camera;
  b.max_iterations = p.ba_max_iterations; b.convergence_limit = p.ba_convergence_limit; b.min_sigma = p.ba_min_tukey_sigma;
  std::map<int, int> view_id, point_id;
  for (int k : adj) { view_id[k] = b.AddCamera(kfs[k]->pose, kfs[k]->fixed); pb->id_view.push_back(k); }
  for (int k : fixed) { view_id[k] = b.AddCamera(kfs[k]->pose, true); pb->id_view.push_back(k); }
  for (int q : points) { point_id[q] = b.AddPoint(pts[q]->pos); pb->id_point.push_back(q); }
  for (int k = 0; k < (int)kfs.size(); k++) {                                                      // :888-902
    if (!view_id.count(k)) continue;
    for (auto& it : kfs[k]->meas) {
      if (!point_id.count(it.first)) continue;
      const int s = level_scale(it.second.level);
      b.AddMeas(view_id[k], point_id[it.first], it.second.root, (double)(s * s));
    }
  }
  abort_flag = false;
  pb->recent = recent;
  pb->accepted = b.Compute(&abort_flag);
  const int acc = pb->accepted;
  if (defer_ba && p.ba_delay_frames > 0) { pb->countdown = p.ba_delay_frames; delete pending; pending = pb; return acc; }
  ApplyBundle(*pb);
  delete pb;
  return acc;
}

void System::ApplyBundle(PendingBA& pb) {
  Bundle& b = pb.b;
  const int nAccepted = pb.accepted;
  const bool recent = pb.recent;
  n_ba_trials += b.n_trials;
  last_ba_accepted = nAccepted;
  if (nAccepted < 0) return;             // reference: mbResetRequested (:913); here surfaced to the caller
  if (nAccepted > 0) {                   // :918-929
    for (size_t i = 0; i < pb.id_point.size(); i++) pts[pb.id_point[i]]->pos = b.pts[i].pos;
    for (size_t i = 0; i < pb.id_view.size(); i++) kfs[pb.id_view[i]]->pose = b.cams[i].pose;
    if (recent) ba_converged_recent = false;
    ba_converged_full = false;
  }
  if (b.converged) { ba_converged_recent = true; if (!recent) ba_converged_full = true; }   // :931-935
  for (auto& pc : b.outlier_meas) {                                                                 // :941-959
    const int pp = pb.id_point[pc.first], pk = pb.id_view[pc.second];
    Measurement& m = kfs[pk]->meas[pp];
    if ((int)pts[pp]->meas_kfs.size() <= 2 || m.source == SRC_ROOT) pts[pp]->bad = true;
    else {
      if (m.source == SRC_TRACKER || m.source == SRC_EPIPOLAR) failure_queue.push_back(std::make_pair(pk, pp));
      else pts[pp]->never_retry.insert(pk);
      kfs[pk]->meas.erase(pp);
      pts[pp]->meas_kfs.erase(pk);
    }
  }
}

}  // namespace orc
