// ptam.h -- C++ mirror of the reference's hot-path classes over the C ABI (vslam_c.h): same class and member names as
// jni/Tracker.h, jni/MapMaker.h, jni/KeyFrame.h, jni/Bundle.h, jni/Map.h, jni/ATANCamera.h, jni/RT.h so the body of
// jni/jni_part.cpp (class SystemPTAM, :18-75) compiles against it unchanged.  Header only; every method forwards to
// libvslam_hip.so -- no algorithm lives here.
//
// OpenCV/Eigen are not required: cv::Mat below is the minimal compatible view the hot path needs (data, step, rows,
// cols); define VSLAM_HAVE_OPENCV before including this header to use the real one.
#pragma once
#include <cstdint>
#include <cstring>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "../vslam_c.h"

#ifndef VSLAM_HAVE_OPENCV
#define CV_8UC1 0
#define CV_8UC4 24
namespace cv {
struct Mat {  // minimal stand-in for cv::Mat as used by jni/KeyFrame.cc:12-23 (8-bit images)
  int rows = 0, cols = 0, channels_ = 1;
  size_t step = 0;
  unsigned char* data = nullptr;
  std::vector<unsigned char> own;
  Mat() {}
  Mat(int r, int c, int type, void* ext = nullptr, size_t st = 0) { init(r, c, type, ext, st); }
  void init(int r, int c, int type, void* ext, size_t st) {
    rows = r; cols = c; channels_ = type == CV_8UC4 ? 4 : 1;
    step = st ? st : (size_t)c * channels_;
    if (ext) data = (unsigned char*)ext; else { own.assign((size_t)r * step, 0); data = own.data(); }
  }
  void create(int r, int c, int type) { if (r != rows || c != cols || !data) init(r, c, type, nullptr, 0); }
  int channels() const { return channels_; }
  int type() const { return channels_ == 4 ? CV_8UC4 : CV_8UC1; }
  template <class T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
  template <class T> T* ptr(int y) { return (T*)(data + (size_t)y * step); }
  void copyTo(Mat& o) const { o.create(rows, cols, type()); for (int y = 0; y < rows; y++) memcpy(o.data + (size_t)y * o.step, data + (size_t)y * step, (size_t)cols * channels_); }
};
}  // namespace cv
#endif

namespace vslam_detail {
inline void check(int rc) { if (rc < 0) throw std::runtime_error(std::string("vslam: ") + vslam_last_error()); }
}

// jni/RT.h:247-312: camera-from-world rigid transform; R row-major.
struct mySE3 {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double t[3] = {0, 0, 0};
  const double* get_translation() const { return t; }
  const double* get_rotation() const { return R; }
};

// jni/ATANCamera.h:55-165: only the parameter block travels; projection runs on device.
class ATANCamera {
 public:
  explicit ATANCamera(const std::string& = "Camera") {}          // jni/ATANCamera.cc:6-29 (hard-coded parameters)
  double params[5] = {0.841906, 1.10893, 0.505171, 0.470265, -0.0133843};
};

// jni/Map.h:21-34.  Owns the device system once the Tracker has created it.
struct Map {
  vslam_system* sys = nullptr;
  ~Map() { if (sys) vslam_destroy(sys); }
  bool IsGood() const { if (!sys) return false; vslam_track_state s; vslam_detail::check(vslam_get_state(sys, 0, &s)); return s.n_keyframes > 0 && s.n_points > 0; }
};

// jni/KeyFrame.h:73-97: the tracker's current frame lives on device; this handle reads it back.
struct KeyFrame {
  vslam_system* sys = nullptr;
  mySE3 se3CfromW;
  cv::Mat im0;                                                      // aLevels[0].im: the host copy MapMaker::InitFromStereo hands to the device
  void MakeKeyFrame_Lite(cv::Mat& im, cv::Mat& /*imColor*/) {     // jni/KeyFrame.cc:5-51
    im.copyTo(im0);                                                 // :12
    vslam_detail::check(vslam_make_keyframe_lite(sys, im.data, im.step, 0, 0));
    vslam_detail::check(vslam_synchronize(sys));
  }
  // jni/KeyFrame.cc:53-95: fast_nonmax + Shi-Tomasi candidates (gvdCandidateMinSTScore = 70); the SmallBlurryImage
  // of the relocaliser is not built.
  void MakeKeyFrame_Rest() { vslam_detail::check(vslam_make_keyframe_rest(sys, 70.0)); }
  struct Candidate { int x, y; double dSTScore; };                  // jni/KeyFrame.h:36-43 (irLevelPos, dSTScore)
  std::vector<Candidate> Candidates(int level) const {              // Level::vCandidates
    int n = 0;
    vslam_detail::check(vslam_read_candidates(sys, 0, level, nullptr, nullptr, 0, &n));
    std::vector<uint32_t> p(n > 0 ? n : 1); std::vector<double> sc(n > 0 ? n : 1);
    vslam_detail::check(vslam_read_candidates(sys, 0, level, p.data(), sc.data(), (int)p.size(), &n));
    std::vector<Candidate> out(n);
    for (int i = 0; i < n; i++) out[i] = {(int)(p[i] & 0xFFFF), (int)(p[i] >> 16), sc[i]};
    return out;
  }
  std::vector<std::pair<int, int>> Corners(int level) const {     // Level::vCorners
    std::vector<uint32_t> c(1 << 20); int n = 0;
    vslam_detail::check(vslam_read_corners(sys, 0, level, c.data(), (int)c.size(), &n));
    std::vector<std::pair<int, int>> out(n);
    for (int i = 0; i < n; i++) out[i] = {(int)(c[i] & 0xFFFF), (int)(c[i] >> 16)};
    return out;
  }
};

// jni/MapMaker.h:33-48 (hot-path members).  The map-maker runs synchronously on device (the reference's thread is
// disabled, jni/MapMaker.cc:56): AddKeyFrame = enqueue + AddKeyFrameFromTopOfQueue + BundleAdjustRecent.
class MapMaker {
 public:
  MapMaker(Map& m, const ATANCamera& cam) : mMap(m), mCamera(cam) {}
  void AddKeyFrame(KeyFrame&) { vslam_detail::check(vslam_add_keyframe(mMap.sys, 0)); }              // :470-478
  void RequestReset() {}
  bool ResetDone() { return true; }                                                                    // no spin (survey fact #4)
  int QueueSize() { return 0; }
  bool NeedNewKeyFrame(KeyFrame&) { int v = 0; vslam_detail::check(vslam_need_new_keyframe(mMap.sys, 0, &v)); return v != 0; }              // :761-773 (the tracker's current frame)
  bool IsDistanceToNearestKeyFrameExcessive(KeyFrame&) { int v = 0; vslam_detail::check(vslam_distance_to_nearest_keyframe_excessive(mMap.sys, 0, &v)); return v != 0; }   // :1098-1101
  // :204-376, the reference's signature (ImageRef = a pair of ints): the two keyframes' images (KeyFrame::im0, kept by MakeKeyFrame_Lite)
  // and the matches go to the device, which makes the map from them (vslam_init_from_stereo); se3TrackerPose as there.  A Tracker
  // built with bBootstrap makes the same call by itself on the second spacebar press (the trails never leave the device); once the
  // map exists this reports it (true and the tracker's pose) without touching it.
  typedef std::pair<int, int> ImageRef;
  bool InitFromStereo(KeyFrame& kFirst, KeyFrame& kSecond, std::vector<std::pair<ImageRef, ImageRef>>& vMatches, mySE3& se3TrackerPose) {
    int info[6];
    vslam_detail::check(vslam_get_init_info(mMap.sys, 0, info));
    if (!info[5]) {
      if (!kFirst.im0.data || !kSecond.im0.data) return false;
      std::vector<int> m(4 * vMatches.size());
      for (size_t i = 0; i < vMatches.size(); i++) { m[4 * i] = vMatches[i].first.first; m[4 * i + 1] = vMatches[i].first.second; m[4 * i + 2] = vMatches[i].second.first; m[4 * i + 3] = vMatches[i].second.second; }
      double q[12];
      const int rc = vslam_init_from_stereo(mMap.sys, kFirst.im0.data, kSecond.im0.data, kFirst.im0.step, (int)vMatches.size(), m.data(), q);
      vslam_detail::check(rc);
      if (rc != 1) return false;
    }
    vslam_track_state s; vslam_detail::check(vslam_get_state(mMap.sys, 0, &s));
    memcpy(se3TrackerPose.R, s.pose, sizeof(se3TrackerPose.R)); memcpy(se3TrackerPose.t, s.pose + 9, sizeof(se3TrackerPose.t));
    return true;
  }
  void BundleAdjustRecent() { vslam_detail::check(vslam_bundle_adjust_recent(mMap.sys)); }             // :801-851
  void BundleAdjustAll() { vslam_detail::check(vslam_bundle_adjust_all(mMap.sys)); }                   // :776-798
  // :393-422, all four levels of the current candidate lists against keyframe nKeyFrame's measurements (< 0: the tracker's)
  void ThinCandidates(int nKeyFrame = -1) { vslam_detail::check(vslam_thin_candidates(mMap.sys, nKeyFrame)); }
  // :1254-1286 "SaveMap": map.dump and keyframes/<i>.info below `dir` (the reference writes to the working directory)
  void GUICommandHandler(const std::string& sCommand, const std::string& dir = ".") {
    if (sCommand == "SaveMap") vslam_detail::check(vslam_save_map(mMap.sys, 0, dir.c_str()));
    else throw std::runtime_error("MapMaker::GUICommandHandler: unhandled command " + sCommand);
  }
 protected:
  Map& mMap;
  ATANCamera mCamera;
};

// jni/Tracker.h:43-150 (public surface).
class Tracker {
 public:
  // bBootstrap: no map is uploaded, the tracker makes its own like the reference's (spacebar, spacebar: jni/Tracker.cc:247-288), and the
  // map-maker grows it on every keyframe (AddKeyFrameFromTopOfQueue, jni/MapMaker.cc:481-506)
  Tracker(int width, int heigth, const ATANCamera& c, Map& m, MapMaker& mm, bool bBootstrap = false) : mMap(m), mMapMaker(mm) {   // jni/Tracker.cc:16-40
    vslam_params p;
    vslam_detail::check(vslam_default_params(&p, width, heigth, 1));
    if (bBootstrap) { p.bootstrap = 1; p.grow_map = 3; }
    for (int i = 0; i < 5; i++) p.cam[i] = c.params[i];
    vslam_detail::check(vslam_create(&p, &mMap.sys));
    mCurrentKF.sys = mMap.sys;
  }
  // jni/Tracker.cc:76-146.  imageColor is only drawn on in the reference; it is ignored here.
  void TrackFrame(cv::Mat& imFrame, cv::Mat& /*imageColor*/, bool /*bDraw*/) {
    if (mbUserPressedSpacebar) { mbUserPressedSpacebar = false; vslam_touch(mMap.sys); }
    vslam_detail::check(vslam_update(mMap.sys, imFrame.data, imFrame.step, 0));
  }
  mySE3 GetCurrentPose() {                                                                              // jni/Tracker.h:58
    vslam_track_state s; vslam_detail::check(vslam_get_state(mMap.sys, 0, &s));
    mySE3 T; memcpy(T.R, s.pose, sizeof(T.R)); memcpy(T.t, s.pose + 9, sizeof(T.t));
    return T;
  }
  std::string GetMessageForUser() { char b[512]; vslam_detail::check(vslam_get_message(mMap.sys, 0, b, sizeof(b))); return b; }   // :880-883
  void Reset() {}                                                                                        // no map-maker spin
  bool mbUserPressedSpacebar = false, mbUserPressedReset = false;                                        // jni/Tracker.h:138-139
 protected:
  KeyFrame mCurrentKF;
  Map& mMap;
  MapMaker& mMapMaker;
};

// jni/Bundle.h:107-160 (public surface), one problem per object.
class Bundle {
 public:
  explicit Bundle(const ATANCamera& TCam, int width = 640, int height = 480, int max_cameras = 64, int max_points = 4096, int max_meas = 65536) {
    vslam_params p;
    vslam_detail::check(vslam_default_params(&p, width, height, 1));
    for (int i = 0; i < 5; i++) p.cam[i] = TCam.params[i];
    vslam_detail::check(vslam_bundle_create(&p, 1, max_cameras, max_points, max_meas, &b));
  }
  ~Bundle() { vslam_bundle_destroy(b); }
  int AddCamera(const mySE3& se3CamFromWorld, bool bFixed) { double q[12]; memcpy(q, se3CamFromWorld.R, 72); memcpy(q + 9, se3CamFromWorld.t, 24); int r = vslam_bundle_add_camera(b, 0, q, bFixed); vslam_detail::check(r); return r; }
  int AddPoint(const double v3Pos[3]) { int r = vslam_bundle_add_point(b, 0, v3Pos); vslam_detail::check(r); return r; }
  void AddMeas(int nCam, int nPoint, const double v2Pos[2], double dSigmaSquared) { vslam_detail::check(vslam_bundle_add_meas(b, 0, nCam, nPoint, v2Pos, dSigmaSquared)); }
  int Compute(bool* /*pbAbortSignal*/) {                                                                // jni/Bundle.cc:136-178
    vslam_detail::check(vslam_bundle_compute(b)); vslam_detail::check(vslam_bundle_synchronize(b));
    int acc = 0; vslam_detail::check(vslam_bundle_get_result(b, 0, &acc, &conv, nullptr, nullptr, nullptr)); return acc;
  }
  bool Converged() { return conv != 0; }
  void GetPoint(int n, double out[3]) { vslam_detail::check(vslam_bundle_get_point(b, 0, n, out)); }
  mySE3 GetCamera(int n) { double q[12]; vslam_detail::check(vslam_bundle_get_camera(b, 0, n, q)); mySE3 T; memcpy(T.R, q, 72); memcpy(T.t, q + 9, 24); return T; }
  std::vector<std::pair<int, int>> GetOutlierMeasurements() {
    std::vector<int> pc(2 * 65536); int n = vslam_bundle_get_outlier_meas(b, 0, pc.data(), 65536); vslam_detail::check(n);
    std::vector<std::pair<int, int>> out; for (int i = 0; i < n; i++) out.push_back({pc[2 * i], pc[2 * i + 1]}); return out;
  }
  std::set<int> GetOutliers() {
    std::vector<int> idx(65536); int n = vslam_bundle_get_outlier_points(b, 0, idx.data(), 65536); vslam_detail::check(n);
    return std::set<int>(idx.begin(), idx.begin() + n);
  }
 private:
  vslam_bundle* b = nullptr;
  int conv = 0;
};
