/*
 * ptam_oracle.h -- C API of the CPU ORACLE for the PTAM tracking + local-BA hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (visualslam_android_amd/csrc, libvslam_hip.so) never links or calls anything here.
 *
 * Every function is a scalar, double-precision restatement of one reference function and
 * cites the reference file:line it follows (paths relative to the reference checkout,
 * ahcorde/visualSLAM_Android).
 *
 * Parity status (see DESIGN.md "Oracle pinning"):
 *   - FAST-10 segment test: pinned -- exhaustive equivalence with the reference's decision
 *     tree (jni/vision/cvfast.cpp:6124-9235) over all 3^16 ring states, by a tree-walk of the
 *     reference text (oracle/pin_fast_tree.py; not a build of the reference).
 *   - Tukey/Cauchy/Huber M-estimators: pinned against jni/MEstimator.h compiled verbatim
 *     (oracle/Makefile -> oracle/_ref/libref_mestimator.so).
 *   - Everything else: the reference has no tests, fixtures or golden vectors and cannot be
 *     built here (needs OpenCV 2.4 + Eigen 3) => "parity unpinned" against the reference,
 *     pinned only by analytic known-answer tests (tests/test_oracle_*.py).
 */
#ifndef PTAM_ORACLE_H
#define PTAM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LEVELS 4

/* reference-quirk flags */
#define ORC_Q_CAM_INT_RADIUS         1  /* jni/ATANCamera.cc:70-82 */
#define ORC_Q_POSE_INT_RESIDUAL      2  /* jni/Tracker.cc:766-767  */
#define ORC_Q_NONMAX_RIGHT_NEIGHBOUR 4  /* jni/vision/cvfast.cpp:9282-9285 */

/* ---- frame front-end ------------------------------------------------------------------ */

/* cv::resize(prev, lev, size/2) call at jni/KeyFrame.cc:20-23, restated as the exact 2:1
 * area filter (a+b+c+d+2)>>2 (third-party arithmetic: parity unpinned). dst is (w/2)x(h/2). */
void orc_halfsample(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);

/* cvCornerFast_10, jni/vision/cvfast.cpp:6088-9241.  corners packed x | y<<16, raster order.
 * Returns the total number of corners (may exceed cap; only the first cap are written). */
int orc_fast10(const uint8_t* img, int w, int h, int stride, int threshold,
               uint32_t* corners, int cap);

/* row LUT, jni/KeyFrame.cc:43-49: lut[y] = first index with corner.y >= y, y in [0,h). */
void orc_row_lut(const uint32_t* corners, int n, int h, int* lut);

/* compute_fast_score_old, jni/vision/cvfast.cpp:9337-9393 */
void orc_fast_score(const uint8_t* img, int w, int h, int stride, const uint32_t* corners, int n,
                    int barrier, int* scores);

/* nonmax_suppression, jni/vision/cvfast.cpp:9243-9335. quirk!=0 reproduces the "check right"
 * bug (corners[i-1] row test; skipped for i==0 where the reference reads out of bounds). */
int orc_nonmax(const uint32_t* corners, const int* scores, int n, int quirk, uint32_t* out);

/* FindShiTomasiScoreAtPoint, jni/vision/ImageHandler.cpp:124-155 */
double orc_shi_tomasi(const uint8_t* img, int stride, int nsize, int px, int py);

/* KeyFrame::MakeKeyFrame_Rest candidate loop, jni/KeyFrame.cc:66-95: maximal corners inside the border whose Shi-Tomasi
 * score (half-window 3) exceeds min_score, in raster order.  Returns the count (may exceed cap). */
int orc_candidates(const uint8_t* img, int w, int h, int stride, const uint32_t* maxcorners, int n, double min_score,
                   int border, uint32_t* out_pos, double* out_score, int cap);
/* MapMaker::ThinCandidates, jni/MapMaker.cc:393-422: drop candidates closer than 10 px (level pixels) to a measurement of
 * the keyframe at the same level or one level up.  out arrays hold n entries.  Returns the number kept. */
int orc_thin_candidates(const uint32_t* pos, const double* score, int n, int level, const double* meas_root,
                        const int* meas_level, int n_meas, uint32_t* out_pos, double* out_score);

/* KeyFrame::MakeKeyFrame_Lite, jni/KeyFrame.cc:5-51: 4-level pyramid + FAST-10 + row LUT.
 * lvl_img[l] must hold (w>>l)*(h>>l) bytes (tight pitch); corners[l] cap entries;
 * lut[l] (h>>l) ints.  Returns 0. */
int orc_make_keyframe_lite(const uint8_t* gray, int w, int h, int stride, const int thr[ORC_LEVELS],
                           uint8_t* const lvl_img[ORC_LEVELS], uint32_t* const corners[ORC_LEVELS],
                           int cap, int ncorners[ORC_LEVELS], int* const lut[ORC_LEVELS]);

/* MiniPatch (jni/MiniPatch.cc): 9x9 raw SSD search at FAST corners inside a +-range box */
int orc_minipatch_sample(const uint8_t* img, int w, int h, int stride, int x, int y, uint8_t* patch /* 81 */);
int orc_minipatch_find(const uint8_t* patch, const uint8_t* img, int w, int h, int stride, const uint32_t* corners,
                       int n, int range, int max_ssd, int pos[2]);

/* ---- whole-path oracle: one sequence (stream) ------------------------------------------------ */

typedef struct orc_params {           /* mirrors the hot-path fields of vslam_params */
  int width, height, patch_size;
  int thr[4]; int nonmax_barrier;
  int max_patches; int coarse_min, coarse_max, coarse_range, coarse_subpix_its, coarse_disabled;
  double coarse_min_vel; int fine_subpix_its; double wls_prior;
  int min_frames_between_kf; double max_kf_dist_wiggle_mult, wiggle_scale;
  int ba_max_iterations; double ba_convergence_limit, ba_min_tukey_sigma; int ba_window, ba_min_keyframes;
  double cam[5]; int quirks;
  int ba_delay_frames;
  int use_sbi;                        /* gvnUseSBI, jni/Tracker.cc:88 */
  int grow_map;                       /* AddSomeMapPoints on every new keyframe, jni/MapMaker.cc:498-501 */
  int idle_iterations;                /* iterations of MapMaker::run's idle jobs per frame (jni/MapMaker.cc:94-117) */
} orc_params;

typedef struct orc_track_state {      /* same fields as vslam_track_state */
  double pose[12];                    /* R row-major (9) + t (3): camera-from-world, jni/Tracker.h:58 */
  double velocity[6];
  double msd_velocity, depth_mean, depth_sigma;
  int attempted[4], found[4];         /* manMeasAttempted/Found, jni/Tracker.h:121-122 */
  int quality, lost_frames, frame, did_coarse;
  int kf_added, n_keyframes, n_points, ba_accepted;
  long long n_zmssd, n_ba_trials;
} orc_track_state;

void* orc_sys_create(const orc_params* p);
void orc_sys_destroy(void* sys);
int orc_sys_add_keyframe(void* sys, const double pose12[12], int fixed, const uint8_t* gray, int stride, double dmean, double dsigma);
int orc_sys_add_point(void* sys, const double pos[3], int src_kf, int src_level, int irx, int iry, const double right[3], const double down[3]);
void orc_sys_add_meas(void* sys, int kf, int pt, int level, const double root[2], int subpix, int source);
void orc_sys_set_map_good(void* sys);
void orc_sys_set_pose(void* sys, const double pose12[12]);
void orc_sys_set_velocity(void* sys, const double v6[6]);
void orc_sys_track_frame(void* sys, const uint8_t* gray, int stride);   /* Tracker::TrackFrame, jni/Tracker.cc:76-146 */
/* the same in stages: frame_begin (MakeKeyFrame_Lite, motion model); search_stage(0) = PVS + coarse selection + SearchForPoints (:369-461);
 * pose_stage(0) = the coarse Gauss-Newton iterations (:463-490); search_stage(1) = fine selection + SearchForPoints (:493-535);
 * pose_stage(1) = the fine iterations + measurement export + scene depth (:543-625); frame_end = motion model, quality, keyframe (:107-132) */
void orc_sys_frame_begin(void* sys, const uint8_t* gray, int stride);
void orc_sys_search_stage(void* sys, int stage);
void orc_sys_pose_stage(void* sys, int stage);
void orc_sys_frame_end(void* sys);
/* map bootstrap (bootstrap.cpp): the next frame of a system without a map starts the trails / runs InitFromStereo (jni/Tracker.cc:247-288) */
void orc_sys_set_last_keyframe_dropped(void* sys, int frame);   /* Tracker::mnLastKeyFrameDropped (jni/Tracker.h), -20 at start */
void orc_sys_press_spacebar(void* sys);
void orc_sys_set_boot_seed(void* sys, unsigned seed);
int orc_sys_init_from_stereo(void* sys, const uint8_t* first, const uint8_t* second, int stride, const int* matches4, int n, double pose12[12]);   /* MapMaker::InitFromStereo with the caller's frames and matches */
void orc_sys_get_init_info(void* sys, int out[6]);   /* stage, trails, InitFromStereo succeeded, homography inliers, points after the stereo pass, map good */
int orc_sys_get_trails(void* sys, int* out4 /* initial x, y, current x, y */, int cap);
/* HomographyInit::Compute (jni/HomographyInit.cc:43-71) on n matches given as 8 doubles each: first (z = 1 plane), second, d pixel / d plane (2x2 row-major) */
int orc_homography_init(const double* matches8, int n, double max_pixel_error, unsigned seed, double out12[12], int* n_inliers);
/* MapMaker::CalcPlaneAligner (jni/MapMaker.cc:1104-1231) on n points */
int orc_calc_plane_aligner(const double* pos3, int n, unsigned seed, double out12[12]);
void orc_sys_idle_job(void* sys, int job);   /* one of them: 0 idle BundleAdjustRecent, 1 ReFindNewlyMade, 2 BundleAdjustAll, 3 ReFindFromFailureQueue */
void orc_sys_idle_iteration(void* sys);        /* one pass through the idle jobs of MapMaker::run, jni/MapMaker.cc:94-117 */
void orc_sys_get_idle_stats(void* sys, int out[6]);   /* points re-found by ReFindNewlyMade / ReFindFromFailureQueue, BundleAdjustAll / idle BundleAdjustRecent calls, queue lengths */
void orc_sys_get_state(void* sys, orc_track_state* out);
/* per map point: found flag, searched flag, search level, did-subpix, found position (L0), projected position */
int orc_sys_get_point_tracks(void* sys, int* found, int* searched, int* level, int* subpix, double* vfound, double* image, int cap);
int orc_sys_get_points(void* sys, double* pos3, int* bad, int* n_in, int* n_out, int cap);
void orc_sys_get_keyframe_pose(void* sys, int kf, double pose12[12]);
int orc_sys_get_keyframe_meas(void* sys, int kf, int* pt, int* level, double* root, int* source, int cap);
int orc_sys_get_template(void* sys, int pt, uint8_t* tmpl, int* sum, int* sumsq, int* bad);
/* every AddPointEpipolar call so far: (level, packed candidate position, stage at which it gave up; 0 = point added) */
int orc_sys_get_grow_log(void* sys, int* out3, int cap);
int orc_sys_bundle_adjust_recent(void* sys);   /* MapMaker::BundleAdjustRecent, jni/MapMaker.cc:801-851 */
int orc_sys_bundle_adjust_all(void* sys);      /* MapMaker::BundleAdjustAll,    jni/MapMaker.cc:776-798 */

/* ---- SmallBlurryImage rotation prior (jni/SmallBlurryImage.cc, jni/Tracker.cc:885-893); third-party arithmetic restated, see sbi.cpp */
/* MakeFromKF on a level-3 image: small image (w3/2 x h3/2) and the zero-mean blurred template; returns w | h << 16 */
int orc_sbi_make(const uint8_t* level3, int w3, int h3, double blur, uint8_t* small_out, float* tmpl_out);
/* CalcSBIRotation between the level-3 images of this and the last frame: ln(SE3fromSE2(IteratePosRelToTarget(6))) */
void orc_sbi_rotation(const uint8_t* cur_l3, const uint8_t* last_l3, int w3, int h3, double blur, const double cam5[5],
                      int quirks, double out6[6], double* score);

/* MapMaker::ReprojectPoint (jni/MapMaker.cc:174-200): point in frame B from its z = 1 projections in A and B; AfromB 12 doubles */
void orc_reproject_point(const double AfromB12[12], const double v2A[2], const double v2B[2], double out3[3]);

/* ---- stand-alone Bundle (jni/Bundle.h:111-121) --------------------------------------------------- */
void* orc_ba_create(const double cam5[5], int width, int height, int quirks, int max_iterations, double convergence_limit, double min_sigma);
void orc_ba_destroy(void* ba);
int orc_ba_add_camera(void* ba, const double pose12[12], int fixed);
int orc_ba_add_point(void* ba, const double pos[3]);
void orc_ba_add_meas(void* ba, int cam, int point, const double pos[2], double sigma_squared);
int orc_ba_compute(void* ba);                  /* Bundle::Compute: accepted iterations or <0 */
void orc_ba_get_camera(void* ba, int n, double pose12[12]);
void orc_ba_get_point(void* ba, int n, double pos[3]);
int orc_ba_converged(void* ba);
int orc_ba_get_outlier_meas(void* ba, int* pc_pairs, int cap);   /* (p, c) pairs */
int orc_ba_get_outlier_points(void* ba, int* idx, int cap);
void orc_ba_get_stats(void* ba, double* sigma2, double* lambda, long long* trials);

/* ---- substrate unit functions -------------------------------------------------------------------- */
void orc_se3_exp(const double mu[6], double pose12[12]);
void orc_se3_ln(const double pose12[12], double mu[6]);
void orc_cam_project(const double cam5[5], int w, int h, int quirks, double cx, double cy, double im[2], double derivs[4], int* invalid, double* largest_radius);
void orc_cam_unproject(const double cam5[5], int w, int h, double ix, double iy, double out[2]);
double orc_find_sigma_squared(int est, const double* v, int n);
double orc_weight(int est, double e2, double s2);
double orc_sqrt_weight(int est, double e2, double s2);
double orc_objective(int est, double e2, double s2);
int orc_transform_image(const uint8_t* in, int iw, int ih, int istride, uint8_t* out, int P, const double M[4], const double inOrig[2], const double outOrig[2]);
int orc_zmssd(const uint8_t* tmpl, int P, const uint8_t* img, int w, int h, int stride, int icol, int irow);

#ifdef __cplusplus
}
#endif
#endif
