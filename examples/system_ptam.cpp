// The reference's JNI glue (jni/jni_part.cpp:18-75, class SystemPTAM) rebuilt on include/vslam/ptam.h with the Android
// camera replaced by the synthetic feeder: create ATANCamera, Map, MapMaker, Tracker; feed frames to TrackFrame.
// Usage: system_ptam [n_frames]        a ground-truth map is uploaded, then n_frames are tracked
//        system_ptam [n_frames] boot   no map: the screen is touched at frames 0 and 12 and the tracker makes its own
//                                      (trail tracking + MapMaker::InitFromStereo on the device), then keeps tracking
// (needs an MI355X; prints the tracker's user message per frame)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../include/vslam/ptam.h"
#include "../include/vslam_feeder.h"

class SystemPTAM {   // same members and construction order as jni/jni_part.cpp:20-46
 public:
  Map* mpMap; MapMaker* mpMapMaker; Tracker* mpTracker; ATANCamera* mpCamera;
  SystemPTAM(int w, int h, bool bootstrap = false) {
    mpCamera = new ATANCamera("Camera");
    mpMap = new Map;
    mpMapMaker = new MapMaker(*mpMap, *mpCamera);
    mpTracker = new Tracker(w, h, *mpCamera, *mpMap, *mpMapMaker, bootstrap);
  }
  ~SystemPTAM() { delete mpTracker; delete mpMapMaker; delete mpMap; delete mpCamera; }
  void onTouchScreen() { mpTracker->mbUserPressedSpacebar = true; }                 // :49-51
  void update(cv::Mat& bw, cv::Mat& rgb) { mpTracker->TrackFrame(bw, rgb, true); }  // :59-71
};

int main(int argc, char** argv) {
  const int W = 640, H = 480, n = argc > 1 ? atoi(argv[1]) : 5;
  const double cam[5] = {0.841906, 1.10893, 0.505171, 0.470265, -0.0133843};
  vslam_feeder* f = nullptr;
  if (vslam_feeder_create(W, H, cam, 1234, 2, &f)) return 2;
  if (argc > 2 && std::string(argv[2]) == "boot") {               // the reference's own start: no map, two touches (jni/Tracker.cc:247-288)
    SystemPTAM boot(W, H, true);
    cv::Mat bw(H, W, CV_8UC1), rgb(H, W, CV_8UC4);
    for (int t = 0; t < n; t++) {
      double p[12];
      vslam_feeder_pose(f, t, p);
      vslam_feeder_render_pose(f, p, 100 + t, bw.data, bw.step);
      if (t == 0 || t == 12) boot.onTouchScreen();
      boot.update(bw, rgb);
      int info[6];
      vslam_get_init_info(boot.mpMap->sys, 0, info);
      printf("frame %d: stage %d trails %d | %s\n", t, info[0], info[1], boot.mpTracker->GetMessageForUser().c_str());
    }
    KeyFrame k0, k1; std::vector<std::pair<MapMaker::ImageRef, MapMaker::ImageRef>> none; mySE3 T;
    const bool ok = boot.mpMapMaker->InitFromStereo(k0, k1, none, T);   // the map exists: reports it
    printf("InitFromStereo: %s, camera at t = (%.3f %.3f %.3f)\n", ok ? "map made" : "no map", T.t[0], T.t[1], T.t[2]);
    vslam_feeder_destroy(f);
    return ok ? 0 : 1;
  }
  if (argc > 2 && std::string(argv[2]) == "stereo") {             // a caller that owns the keyframes and the matches: MapMaker::InitFromStereo itself
    // the matches: the trails of a first system after 12 frames (any matcher would do)
    std::vector<std::pair<MapMaker::ImageRef, MapMaker::ImageRef>> matches;
    cv::Mat first(H, W, CV_8UC1), second(H, W, CV_8UC1), rgb(H, W, CV_8UC4);
    {
      SystemPTAM a(W, H, true);
      cv::Mat bw(H, W, CV_8UC1);
      for (int t = 0; t < 12; t++) {
        double p[12];
        vslam_feeder_pose(f, t, p);
        vslam_feeder_render_pose(f, p, 100 + t, bw.data, bw.step);
        if (t == 0) { a.onTouchScreen(); bw.copyTo(first); }
        if (t == 11) bw.copyTo(second);
        a.update(bw, rgb);
      }
      std::vector<int> tr(4 * 1000); int nt = 0;
      vslam_get_trails(a.mpMap->sys, 0, tr.data(), 1000, &nt);
      for (int i = 0; i < nt; i++) matches.push_back({{tr[4 * i], tr[4 * i + 1]}, {tr[4 * i + 2], tr[4 * i + 3]}});
    }
    SystemPTAM b(W, H, true);
    KeyFrame kF, kS; kF.sys = kS.sys = b.mpMap->sys;
    first.copyTo(kF.im0); second.copyTo(kS.im0);
    mySE3 T;
    const bool ok = b.mpMapMaker->InitFromStereo(kF, kS, matches, T);
    printf("InitFromStereo(kFirst, kSecond, %d matches): %s, camera at t = (%.3f %.3f %.3f)\n", (int)matches.size(), ok ? "map made" : "no map", T.t[0], T.t[1], T.t[2]);
    int good = 0;
    cv::Mat bw(H, W, CV_8UC1);
    for (int t = 12; t < 12 + n; t++) {                             // and the tracker follows the map it was handed
      double p[12];
      vslam_feeder_pose(f, t, p);
      vslam_feeder_render_pose(f, p, 100 + t, bw.data, bw.step);
      b.update(bw, rgb);
      const std::string msg = b.mpTracker->GetMessageForUser();
      if (msg.find("quality good") != std::string::npos) good++;
      if (t == 12 + n - 1) printf("frame %d: %s\n", t, msg.c_str());
    }
    vslam_feeder_destroy(f);
    return ok && good == n ? 0 : 1;
  }
  SystemPTAM sys(W, H);
  vslam_system* dev = sys.mpMap->sys;
  // ground-truth map from one source keyframe (the bootstrap InitFromStereo is out of scope): every 7th maximal corner
  double kfpose[12];
  vslam_feeder_pose(f, -20, kfpose);
  std::vector<uint8_t> img((size_t)W * H);
  vslam_feeder_render_pose(f, kfpose, 1, img.data(), W);
  if (vslam_map_add_keyframe(dev, 0, kfpose, 1, img.data(), W, 1.0, 0.1) < 0) { fprintf(stderr, "%s\n", vslam_last_error()); return 1; }
  vslam_make_keyframe_lite(dev, img.data(), W, 0, 0);
  vslam_fast_nonmax(dev);
  int npts = 0;
  for (int l = 0; l < 4; l++) {
    std::vector<uint32_t> c(100000); int nc = 0;
    vslam_read_max_corners(dev, 0, l, c.data(), nullptr, (int)c.size(), &nc);
    for (int i = 0; i < nc; i += 3) {
      const int x = c[i] & 0xFFFF, y = c[i] >> 16;
      if (x < 10 || y < 10 || x >= (W >> l) - 10 || y >= (H >> l) - 10) continue;
      double pos[3], r[3], d[3];
      if (vslam_feeder_make_point(f, kfpose, l, x, y, pos, r, d)) continue;
      const int p = vslam_map_add_point(dev, 0, pos, 0, l, x, y, r, d);
      if (p < 0) break;
      const double root[2] = {(x + 0.5) * (1 << l) - 0.5, (y + 0.5) * (1 << l) - 0.5};
      vslam_map_add_measurement(dev, 0, 0, p, l, root, 1, 2);
      npts++;
    }
  }
  vslam_map_set_good(dev, 0);
  double start[12];
  vslam_feeder_pose(f, -1, start);
  vslam_set_pose(dev, 0, start);
  printf("map: 1 keyframe, %d points\n", npts);
  cv::Mat bw(H, W, CV_8UC1), rgb(H, W, CV_8UC4);
  for (int t = 0; t < n; t++) {
    double p[12];
    vslam_feeder_pose(f, t, p);
    vslam_feeder_render_pose(f, p, 100 + t, bw.data, bw.step);
    sys.update(bw, rgb);
    const mySE3 T = sys.mpTracker->GetCurrentPose();
    double err = 0;
    for (int i = 0; i < 3; i++) { const double e = T.t[i] - p[9 + i]; err = e * e > err ? e * e : err; }
    printf("frame %d: %s | |t - t_true|_max = %.2e\n", t, sys.mpTracker->GetMessageForUser().c_str(), err > 0 ? __builtin_sqrt(err) : 0.0);
  }
  vslam_feeder_destroy(f);
  return 0;
}
