// ORACLE (test infrastructure only -- see ptam_oracle.h).  CPU restatement of the PTAM tracking and
// mapping hot path: PatchFinder, TrackerData, Tracker::TrackFrame/TrackMap, Bundle, MapMaker's BA driver.
// Scalar double precision, one function per reference function, file:line cited at each.
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "ptam_math.hpp"
#include "ptam_oracle.h"

namespace orc {

enum { SRC_TRACKER = 0, SRC_REFIND = 1, SRC_ROOT = 2, SRC_TRAIL = 3, SRC_EPIPOLAR = 4 };  // jni/KeyFrame.h:50

struct Measurement {  // jni/KeyFrame.h:46-51
  int level; bool subpix; double root[2]; int source;
};

struct KeyFrame {     // jni/KeyFrame.h:73-97 (+ Level :55-69)
  SE3 pose; bool fixed = false;
  int w[4], h[4];
  std::vector<uint8_t> im[4];
  std::vector<uint32_t> corners[4];
  std::vector<int> lut[4];
  std::vector<uint32_t> maxcorners[4];
  std::vector<uint32_t> cand[4]; std::vector<double> cand_score[4];   // Level::vCandidates (irLevelPos packed, dSTScore)
  std::map<int, Measurement> meas;   // keyed by map-point index (reference: by MapPoint*, address order)
  double depth_mean = 0, depth_sigma = 0;
};

// PatchFinder state that persists per map point (jni/PatchFinder.h:96-128)
struct Finder {
  int P = 11;                 // mnPatchSize
  int max_ssd = 0;            // mnMaxSSD
  std::vector<uint8_t> tmpl;  // mimTemplate
  int tsum = 0, tsumsq = 0;
  double warp_inv[4] = {0, 0, 0, 0};   // mm2WarpInverse
  int level = 0;              // mnSearchLevel
  double hinv[9];             // mm3HInv
  std::vector<double> jac[2]; // mimJacs  [(x-1)*(P-2) + (y-1)]
  double subpix[2]; double mean_diff = 0;
  double coarse[2];
  bool found = false, bad = false;
  bool have_last = false;     // mpLastTemplateMapPoint == &p
  double last_warp[4] = {9999.9, 0, 0, 9999.9};   // :23
  long n_zmssd = 0;           // statistics: ZMSSD evaluations (K of SURVEY 8(d))
};

struct MapPoint {     // jni/MapPoint.h:22-69 + TrackerData (jni/TrackerData.h:36-66)
  V3 pos; bool bad = false;
  bool boot = false;             // made by InitFromStereo: its pixel vectors use the reference's swapped neighbours (jni/MapMaker.cc:271-292)
  int src_kf = 0, src_level = 0; int irx = 0, iry = 0;
  V3 pix_right, pix_down;
  int n_outlier = 0, n_inlier = 0;
  std::set<int> meas_kfs;        // MapMakerData::sMeasurementKFs
  std::set<int> never_retry;     // MapMakerData::sNeverRetryKFs
  // TrackerData
  Finder finder;
  V3 cam; double implane[2] = {0, 0}; double image[2] = {0, 0}; double derivs[4] = {0, 0, 0, 0};
  bool in_image = false, pot_visible = false;
  int search_level = 0; bool searched = false, found = false, did_subpix = false;
  double vfound[2] = {0, 0}; double sqrt_inv_noise = 0;
  double err_cov[2] = {0, 0}; double jac[12];
};

struct Params {
  int width, height, patch_size;
  int thr[4]; int nonmax_barrier;
  int max_patches; int coarse_min, coarse_max, coarse_range, coarse_subpix_its, coarse_disabled;
  double coarse_min_vel; int fine_subpix_its; double wls_prior;
  int min_frames_between_kf; double max_kf_dist_wiggle_mult, wiggle_scale;
  int ba_max_iterations; double ba_convergence_limit, ba_min_tukey_sigma; int ba_window, ba_min_keyframes;
  double cam[5]; int quirks;
  int ba_delay_frames = 0;   // 0: results applied at once; D > 0: applied at the start of the D-th following frame
  int use_sbi = 0;           // gvnUseSBI, jni/Tracker.cc:88 (reference: 1)
  int idle_iterations = 0;   // iterations of MapMaker::run's idle jobs (jni/MapMaker.cc:94-117) after every frame; 0: one BundleAdjustRecent per keyframe only
  int grow_map = 0;          // bit 0: AddSomeMapPoints (jni/MapMaker.cc:498-501), bit 1: ReFindInSingleKeyFrame (:497); 0: only the tracker's measurements
};

// SmallBlurryImage (jni/SmallBlurryImage.h): mimSmall, mimTemplate (zero-mean, blurred), mimImageJacs (x, y interleaved)
struct SBI { int w = 0, h = 0; std::vector<uint8_t> small; std::vector<float> tmpl, jacs; bool made_jacs = false; };
void sbi_make(SBI& s, const uint8_t* level3, int w3, int h3, double blur);
void sbi_make_jacs(SBI& s);
void calc_sbi_rotation(const SBI& cur, SBI& last, const Camera& cam, bool quirk_int_radius, double out6[6], double* score);

// ---- Bundle (jni/Bundle.{h,cc}) --------------------------------------------------------------------------------
struct BCamera { bool fixed; SE3 pose, pose_new; double U[36]; double ea[6]; int start_row; };
struct BPoint { V3 pos, pos_new; double V[9]; double eb[3]; double Vinv[9]; int n_meas = 0, n_outliers = 0; std::set<int> cams;
                std::vector<std::pair<int,int>> script; };
struct BMeas { int p, c; bool bad = false; bool erased = false; double found[2]; double eps[2]; double A[12]; double B[6]; double W[18];
               double sqrt_inv_noise; V3 cam; double err2; double derivs[4]; };

struct Bundle {
  Camera camera;
  std::vector<BCamera> cams; std::vector<BPoint> pts; std::vector<BMeas> meas;  // meas in AddMeas order; erased flagged
  std::vector<std::vector<int>> lut;  // [cam][point] -> meas index or -1
  std::vector<std::pair<int,int>> outlier_meas;
  int n_cams_to_update = 0, next_start_row = 0;
  double sigma2 = 0, lambda = 0, lambda_factor = 0;
  bool converged = false, hit_max = false; int counter = 0, accepted = 0;
  int max_iterations = 20; double convergence_limit = 1e-6; double min_sigma = 0.4;
  long n_trials = 0;
  int AddCamera(const SE3& pose, bool fixed);
  int AddPoint(V3 pos);
  void AddMeas(int cam, int point, const double pos[2], double sigma2);
  int Compute(const bool* abort);
  bool Do_LM_Step(const bool* abort);
  double FindNewError();
  void ProjectAndFindSquaredError(BMeas& m);
  std::set<int> GetOutliers() const;
};

// ---- the whole single-stream system: Map + MapMaker (BA driver part) + Tracker --------------------------------
struct System {
  Params p; Camera camera;
  std::vector<KeyFrame*> kfs; std::vector<MapPoint*> pts;
  bool map_good = false;
  double wiggle_depth_norm = 0;   // mdWiggleScaleDepthNormalized
  bool ba_converged_recent = true, ba_converged_full = true;
  std::vector<std::pair<int,int>> failure_queue;
  // tracker
  KeyFrame cur;
  SE3 pose, start_pose; double velocity[6] = {0, 0, 0, 0, 0, 0};
  SBI sbi_this, sbi_last; bool have_sbi = false; double sbi_rot[6] = {0, 0, 0, 0, 0, 0}; double sbi_score = 0;   // mpSBIThisFrame / mpSBILastFrame / mv6SBIRot
  double msd_vel = 0; bool did_coarse = false; bool just_recovered = false;
  int frame = 0, last_kf_dropped = -20, lost_frames = 0; int quality = 2;  // 0 BAD 1 DODGY 2 GOOD
  int attempted[4], found[4];
  bool kf_added_this_frame = false;
  long n_zmssd = 0, n_ba_trials = 0; int last_ba_accepted = -2;
  std::vector<int> iteration_set;

  explicit System(const Params& pp);
  ~System();
  int AddKeyFrameRaw(const double pose12[12], bool fixed, const uint8_t* gray, int stride, double dmean, double dsigma);
  int AddPointRaw(const double pos[3], int src_kf, int src_level, int irx, int iry, const double right[3], const double down[3]);
  void AddMeasRaw(int kf, int pt, int level, const double root[2], bool subpix, int source);
  void SetMapGood();
  // Tracker (jni/Tracker.cc)
  void TrackFrame(const uint8_t* gray, int stride);
  void TrackMap();
  // TrackFrame in stages (same statements, same order): FrameBegin; SearchStage(0); PoseStage(0); SearchStage(1); PoseStage(1); FrameEnd
  void FrameBegin(const uint8_t* gray, int stride); void FrameEnd(); void TrackerFrameEnd(); void SearchStage(int stage); void PoseStage(int stage);
  bool Tracking() const { return map_good && lost_frames < 3; }
  std::vector<int> tm_pvs[4], tm_next, tm_iter; bool tm_coarse_tried = false; unsigned tm_coarse_found = 0; bool tracked_this_frame = false;
  int SearchForPoints(std::vector<int>& v, int range, int subpix_its);
  void CalcPoseUpdate(const std::vector<int>& v, double override_sigma, bool mark_outliers, double out[6]);
  void ApplyMotionModel(); void UpdateMotionModel(); void AssessTrackingQuality();
  // MapMaker (jni/MapMaker.cc)
  void AddKeyFrame();            // AddKeyFrame + AddKeyFrameFromTopOfQueue + one BundleAdjustRecent
  bool NeedNewKeyFrame();
  double KeyFrameLinearDist(const SE3& a, const SE3& b);
  int BundleAdjustRecent(); int BundleAdjustAll(); void HandleBadPoints();
  // map growth (mapgrow.cpp)
  void ThinCandidates(KeyFrame& k, int level); int ClosestKeyFrame(int kidx);
  bool AddPointEpipolar(int ksrc, int ktgt, int level, int candidate); int AddSomeMapPoints(int level);
  bool ReFind_Common(int kidx, int pi); int ReFindInSingleKeyFrame(int kidx);
  // the idle jobs of MapMaker::run (jni/MapMaker.cc:94-117)
  std::vector<int> new_queue;      // mqNewQueue: points AddPointEpipolar made, waiting for ReFindNewlyMade
  long idle_count = 0;             // evaluations of the lowest-priority job's condition (stands in for rand() % 20 == 0, :112)
  int n_refound_new = 0, n_refound_failed = 0, n_ba_all = 0, n_ba_recent_idle = 0;
  void ReFindNewlyMade(); void ReFindFromFailureQueue(); void IdleIteration(); void IdleJob(int job);
  Finder refinder; int refind_last_point = -1;     // ReFind_Common's static PatchFinder and its mpLastTemplateMapPoint
  int n_points_added = 0, n_refound = 0;
  std::vector<int> grow_log;   // per AddPointEpipolar call: level, packed candidate position, stage at which it gave up (0 = point added)
  int BundleAdjust(const std::vector<int>& adj, const std::vector<int>& fixed, const std::vector<int>& points, bool recent);
  // a finished Bundle whose results are still to be written to the map (asynchronous map-maker model, see mapmaker.cpp)
  struct PendingBA { Bundle b; std::vector<int> id_view, id_point; bool recent = true; int accepted = 0; int countdown = -1; };
  PendingBA* pending = nullptr;
  void ApplyBundle(PendingBA& pb);
  bool defer_ba = false;
  bool abort_flag = false;
  // map bootstrap (bootstrap.cpp): Tracker::TrackForInitialMap and the trails (jni/Tracker.cc:247-346), MapMaker::InitFromStereo (jni/MapMaker.cc:204-376)
  struct Trail { uint8_t patch[81]; int init[2], cur[2]; };
  int init_stage = 0;            // TRAIL_TRACKING_NOT_STARTED / STARTED / COMPLETE
  bool spacebar = false;         // mbUserPressedSpacebar
  bool init_ok = false; int n_hom_inliers = 0, n_init_points = 0;
  unsigned boot_seed = 1;        // stands in for the reference's rand() state
  std::vector<Trail> trails; KeyFrame first_kf, prev_kf;
  void TrackForInitialMap(); void TrailTrackingStart(); int TrailTrackingAdvance(int max_ssd);
  bool InitFromStereo(const KeyFrame& kF, const KeyFrame& kS, const std::vector<std::array<int, 4>>& trail_matches);
  void RefreshSceneDepth(KeyFrame& k);
  SE3 CalcPlaneAligner(); void ApplyGlobalTransformationToMap(const SE3& new_from_old);
};

// map bootstrap mathematics (homography.cpp): HomographyMatch / HomographyDecomposition of jni/HomographyInit.h, HomographyInit::Compute, CalcPlaneAligner
struct HMatch { double first[2], second[2], jac[4]; };           // v2CamPlaneFirst, v2CamPlaneSecond, m2PixelProjectionJac (row-major)
struct HDecomposition { double Rp[9], Tp[3], n[3], d; double R[9], t[3]; int score; };
bool homography_init_compute(const std::vector<HMatch>& m, double max_pixel_error, unsigned seed, SE3& second_from_first, int* n_inliers);
bool calc_plane_aligner(const std::vector<V3>& points, unsigned seed, SE3& aligner);
void make_keyframe_lite(KeyFrame& k, const uint8_t* gray, int w, int h, int stride, const int thr[4]);
void make_keyframe_rest_nonmax(KeyFrame& k, int barrier, bool quirk);
void make_keyframe_rest_candidates(KeyFrame& k, double min_score);
V3 reproject_point(const SE3& AfromB, const double v2A[2], const double v2B[2]);
void svd4_smallest_right_vector(const double A[16], double out[4]);

// PatchFinder pieces exposed for unit tests
int transform_image(const uint8_t* in, int iw, int ih, int istride, uint8_t* out, int P, const double M[4],
                    const double inOrig[2], const double outOrig[2]);
int finder_calc_level_and_warp(Finder& f, const MapPoint& p, const SE3& pose, const double derivs[4]);
void finder_make_template(Finder& f, const MapPoint& p, const KeyFrame& src);
bool finder_find_coarse(Finder& f, const double irPos[2], const KeyFrame& kf, unsigned range);
int finder_zmssd(const Finder& f, const uint8_t* img, int w, int h, int stride, int icol, int irow);
void finder_make_subpix(Finder& f);
bool finder_iterate_subpix_to_convergence(Finder& f, const KeyFrame& kf, int max_its);

}  // namespace orc
