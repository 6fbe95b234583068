// ORACLE (test infrastructure only).  Self-test of the CPU restatement and of the host side of the synthetic feeder, built with
// -fsanitize=address,undefined (make -C oracle sanitize) and run by tests/test_sanitizers.py: a seeded scene, a ground-truth
// map built the way visualslam_android_amd/feeder.py builds it (grid-thinned maximal FAST corners back-projected by
// vslam_feeder_make_point, measurements by vslam_feeder_project), then Tracker::TrackFrame + MapMaker::AddKeyFrame with map
// growth, the asynchronous map-maker model, the stand-alone Bundle and the front-end pieces.  Exit code 0 = no finding.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/vslam_feeder.h"
#include "ptam_system.hpp"

using namespace orc;

static const double kCam[5] = {0.841906, 1.10893, 0.505171, 0.470265, -0.0133843};

int main() {
  const int W = 320, H = 240, NKF = 8, NFRAMES = 46;
  vslam_feeder* f = nullptr;
  if (vslam_feeder_create(W, H, kCam, 77, 2, &f)) return 2;
  Params p{};
  p.width = W; p.height = H; p.patch_size = 8;
  p.thr[0] = 10; p.thr[1] = 15; p.thr[2] = 15; p.thr[3] = 10; p.nonmax_barrier = 10;
  p.max_patches = 1000; p.coarse_min = 20; p.coarse_max = 60; p.coarse_range = 30; p.coarse_subpix_its = 8; p.coarse_disabled = 0;
  p.coarse_min_vel = 0.006; p.fine_subpix_its = 8; p.wls_prior = 100.0; p.min_frames_between_kf = 20;
  p.max_kf_dist_wiggle_mult = 0.2; p.wiggle_scale = 0.1; p.ba_max_iterations = 20; p.ba_convergence_limit = 1e-6;
  p.ba_min_tukey_sigma = 0.4; p.ba_window = 5; p.ba_min_keyframes = 8;
  for (int i = 0; i < 5; i++) p.cam[i] = kCam[i];
  p.quirks = 0; p.ba_delay_frames = 3; p.use_sbi = 1; p.grow_map = 3;
  System sys(p);
  // ---- ground-truth map: NKF keyframes at frames -20 NKF .. -20 ----
  std::vector<std::vector<double>> poses(NKF, std::vector<double>(12));
  std::vector<uint8_t> img((size_t)W * H);
  const int per_level[4] = {120, 50, 20, 8};
  for (int k = 0; k < NKF; k++) {
    vslam_feeder_pose(f, -20.0 * (NKF - k), poses[k].data());
    vslam_feeder_render_pose(f, poses[k].data(), 1000 + k, img.data(), W);
    sys.AddKeyFrameRaw(poses[k].data(), k == 0, img.data(), W, 1.0, 0.1);
  }
  int npts = 0, nmeas = 0;
  for (int k = 0; k < NKF; k++) {
    const KeyFrame& kf = *sys.kfs[k];
    for (int l = 0; l < 4; l++) {
      const int lw = W >> l, lh = H >> l, s = 1 << l;
      int taken = 0, cell = 12;
      std::vector<char> used((size_t)(lw / cell + 1) * (lh / cell + 1), 0);
      for (uint32_t c : kf.maxcorners[l]) {
        const int x = c & 0xFFFF, y = c >> 16;
        if (x < 10 || y < 10 || x >= lw - 10 || y >= lh - 10 || taken >= per_level[l]) continue;
        char& u = used[(size_t)(y / cell) * (lw / cell + 1) + x / cell];
        if (u) continue;
        u = 1;
        double pos[3], right[3], down[3];
        if (vslam_feeder_make_point(f, poses[k].data(), l, x, y, pos, right, down) != 0) continue;
        const int pid = sys.AddPointRaw(pos, k, l, x, y, right, down);
        const double root[2] = {(x + 0.5) * s - 0.5, (y + 0.5) * s - 0.5};
        sys.AddMeasRaw(k, pid, l, root, true, SRC_ROOT);
        npts++; taken++; nmeas++;
        for (int k2 = 0; k2 < NKF; k2++) {
          if (k2 == k) continue;
          double im[2], depth;
          if (vslam_feeder_project(f, poses[k2].data(), pos, 12 * s, im, &depth) == 1) { sys.AddMeasRaw(k2, pid, l, im, true, SRC_TRACKER); nmeas++; }
        }
      }
    }
  }
  sys.SetMapGood();
  std::vector<double> start(12);
  vslam_feeder_pose(f, -1.0, start.data());
  for (int i = 0; i < 9; i++) sys.pose.R[i] = start[i];
  for (int i = 0; i < 3; i++) sys.pose.t[i] = start[9 + i];
  // ---- the sequence ----
  std::vector<uint8_t> frames((size_t)NFRAMES * W * H);
  vslam_feeder_render(f, 0, NFRAMES, frames.data(), W, (size_t)W * H, 2);
  int good = 0, kfs = 0;
  for (int t = 0; t < NFRAMES; t++) {
    sys.TrackFrame(frames.data() + (size_t)t * W * H, W);
    good += sys.quality == 2;
    kfs += sys.kf_added_this_frame;
  }
  std::printf("map %d points %d measurements; %d/%d frames good, %d keyframes added, %zu points at the end, %ld LM trials\n", npts, nmeas, good, NFRAMES, kfs,
              sys.pts.size(), sys.n_ba_trials);
  if (good < NFRAMES - 2 || kfs < 2 || (int)sys.pts.size() <= npts) { std::fprintf(stderr, "selftest: the tracker did not follow the sequence\n"); vslam_feeder_destroy(f); return 3; }
  sys.BundleAdjustAll(); sys.HandleBadPoints();
  // stand-alone Bundle on the adjusted map's first cameras
  {
    Bundle b; b.camera = sys.camera;
    for (int k = 0; k < 4; k++) b.AddCamera(sys.kfs[k]->pose, k == 0);
    std::map<int, int> pid;
    for (int k = 0; k < 4; k++)
      for (auto& it : sys.kfs[k]->meas) {
        if (!pid.count(it.first)) pid[it.first] = b.AddPoint(sys.pts[it.first]->pos);
        b.AddMeas(k, pid[it.first], it.second.root, (double)((1 << it.second.level) * (1 << it.second.level)));
      }
    bool abort_flag = false;
    const int acc = b.Compute(&abort_flag);
    std::printf("stand-alone bundle: %d accepted, %ld trials, %zu outlier measurements\n", acc, b.n_trials, b.outlier_meas.size());
    if (acc < 0) return 4;
  }
  vslam_feeder_destroy(f);
  return 0;
}
