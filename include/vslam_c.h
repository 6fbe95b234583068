/*
 * vslam_c.h -- C ABI of libvslam_hip.so: the MI355X (gfx950) implementation of the PTAM
 * tracking + local bundle-adjustment hot path of ahcorde/visualSLAM_Android.
 *
 * The only C ABI the reference has is its JNI surface (jni/jni_part.cpp:84-145):
 *   native_createTest / native_disposeTest / native_touchScreen / native_update(gray, rgba).
 * vslam_create / vslam_destroy / vslam_touch / vslam_update replace those one for one
 * (INTEGRATION.md shows the JNI stub that binds them).  The remaining entry points expose the
 * stages behind Tracker::TrackFrame (jni/Tracker.cc:76-146) and MapMaker::AddKeyFrame
 * (jni/MapMaker.cc:470-478) individually for parity tests and benchmarks; each cites the
 * reference function it replaces.
 *
 * Conventions: plain pointers and sizes only; every call returns 0 on success or a negative
 * VSLAM_E_* code (never aborts; the reference has no error convention, jni_part.cpp:144
 * returns constant 0).  One vslam_system holds n_streams independent sequences that are
 * processed in lock-step by batched kernels on one HIP stream; calls on one system must be
 * serialised by the caller, different systems are independent.  Inputs are borrowed for the
 * duration of the call; outputs go to caller-provided buffers.
 */
#ifndef VSLAM_C_H
#define VSLAM_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSLAM_LEVELS 4 /* jni/KeyFrame.h:33 */

#define VSLAM_OK 0
#define VSLAM_E_INVALID (-1)  /* bad argument */
#define VSLAM_E_HIP (-2)      /* HIP runtime error, see vslam_last_error() */
#define VSLAM_E_CAPACITY (-3) /* a fixed-capacity device buffer overflowed */
#define VSLAM_E_STATE (-4)    /* call not valid in the current state */

/* reference-quirk switches. Default 0 = PTAM-intended semantics. */
#define VSLAM_Q_CAM_INT_RADIUS 1         /* jni/ATANCamera.cc:70-82  */
#define VSLAM_Q_POSE_INT_RESIDUAL 2      /* jni/Tracker.cc:766-767   */
#define VSLAM_Q_NONMAX_RIGHT_NEIGHBOUR 4 /* jni/vision/cvfast.cpp:9282-9285 */

typedef struct vslam_system vslam_system;

/* Every tunable the reference hard-codes on the hot path. */
typedef struct vslam_params {
  int width, height;               /* jni/jni_part.cpp:41 (800x480 there) */
  int n_streams;                   /* independent sequences batched on this GPU */
  int fast_threshold[VSLAM_LEVELS];/* jni/KeyFrame.cc:32-39: 10,15,15,10 */
  int nonmax_barrier;              /* jni/KeyFrame.cc:63: 10 */
  int patch_size;                  /* jni/PatchFinder.h:48: 11 (BASELINE configs: 8) */
  int max_corners[VSLAM_LEVELS];   /* device capacity per stream and level */
  int max_points;                  /* map-point capacity per stream */
  int max_keyframes;               /* keyframe capacity per stream */
  int max_patches_per_frame;       /* jni/Tracker.cc:518: 1000 */
  int coarse_min, coarse_max;      /* jni/Tracker.cc:405-406: 20, 60 */
  int coarse_range;                /* jni/Tracker.cc:407: 30 */
  int coarse_subpix_its;           /* jni/Tracker.cc:408: 8 */
  int coarse_disabled;             /* jni/Tracker.cc:409: 0 */
  double coarse_min_vel;           /* jni/Tracker.cc:410: 0.006 */
  int fine_subpix_its;             /* jni/Tracker.cc:505: 8 (level 3 only) */
  double wls_prior;                /* jni/Tracker.cc:734: 100 */
  int use_sbi;                     /* jni/Tracker.cc:88 gvnUseSBI (reference 1): SmallBlurryImage rotation prior in the motion model; default 0 */
  int min_frames_between_kf;       /* jni/Tracker.cc:128: 20 */
  double max_kf_dist_wiggle_mult;  /* jni/MapMaker.cc:768: 0.2 */
  double wiggle_scale;             /* jni/MapMaker.cc:57: 0.1 */
  int ba_max_iterations;           /* jni/Bundle.cc:65: 20 */
  double ba_convergence_limit;     /* jni/Bundle.cc:66: 1e-6 */
  double ba_min_tukey_sigma;       /* jni/Bundle.cc:224: 0.4 */
  int ba_window;                   /* jni/MapMaker.cc:812-820: newest + 4 nearest = 5 */
  int ba_min_keyframes;            /* jni/MapMaker.cc:803: 8 */
  double cam[5];                   /* jni/ATANCamera.cc:20-24 normalised fx fy cx cy w */
  int quirks;                      /* VSLAM_Q_* bit set */
  int device;                      /* HIP device ordinal */
  int ba_delay_frames;             /* 0: a keyframe's bundle adjustment is finished before the next frame (synchronous
                                      map-maker); D > 0: it runs on its own HIP stream beside the next frames and its results
                                      are applied at the start of the D-th following frame (the reference's map-maker is a
                                      second thread whose results also arrive "a few frames later", jni/MapMaker.cc:80-123) */
  int grow_map;                    /* bit flags, the rest of MapMaker::AddKeyFrameFromTopOfQueue (jni/MapMaker.cc:481-506) on every new keyframe:
                                      1: MakeKeyFrame_Rest's candidates, ThinCandidates and AddSomeMapPoints (epipolar search +
                                         triangulation, :498-501, 525-703), so the map gains points;
                                      2: ReFindInSingleKeyFrame (:497, 967-1056), so the keyframe also gains measurements of points the
                                         tracker did not measure in it;
                                      3: both, the reference's behaviour; 0 (default): only the tracker's measurements are stored */
  int ba_batch_frames;             /* asynchronous map-maker only (ba_delay_frames > 0): Bundle::Compute of the keyframes of this many consecutive
                                      frames is launched together (1 .. ba_delay_frames; 0 = 1).  Independent sequences ask for keyframes in
                                      different frames; one launch per frame would carry only a few problems.  Does not change any result: a
                                      keyframe's adjustment is still applied ba_delay_frames frames after the keyframe */
  int idle_iterations;             /* MapMaker::run's idle jobs (jni/MapMaker.cc:94-117) after every frame, this many passes: BundleAdjustRecent until it
                                      has converged, ReFindNewlyMade (:1061-1081), BundleAdjustAll until it has converged (:776-798), every 20th pass
                                      ReFindFromFailureQueue (:1083-1096; the reference draws rand() % 20), HandleBadPoints.  0 (default): one
                                      BundleAdjustRecent per keyframe only.  -1: none after a frame, but the failure queue and never-retry sets are kept so that
                                      vslam_mapmaker_idle_job can run the jobs on request.  Needs the synchronous map-maker (ba_delay_frames = 0) */
  int bootstrap;                   /* 1: a stream without a map runs Tracker::TrackForInitialMap (jni/Tracker.cc:247-288): vslam_press_spacebar starts the
                                      trails on the next frame, a second press runs MapMaker::InitFromStereo (HomographyInit, the stereo points,
                                      AddSomeMapPoints, BundleAdjustAll, CalcPlaneAligner) on the frame that consumes it.  Needs grow_map != 0
                                      (keyframe corner lists) and the synchronous map-maker for the streams being initialised.  0 (default): maps are uploaded */
  int ba_sum_order;                /* 0 (default): Bundle::Compute sums U, V, the reduced camera system and the objectives as parallel partial sums and
                                      matrix-core products (results within ~1e-10 of the reference's loops).  1: every sum in the order of the
                                      reference's loops (jni/Bundle.cc:241-321, 362-470, 537-561) -- same results as the reference's sequential
                                      code bit for bit, several times slower: the parity mode */
} vslam_params;

const char* vslam_last_error(void);
int vslam_default_params(vslam_params* p, int width, int height, int n_streams);

/* native_createTest / native_disposeTest (jni/jni_part.cpp:109-118) */
int vslam_create(const vslam_params* p, vslam_system** out);
int vslam_destroy(vslam_system* sys);
int vslam_synchronize(vslam_system* sys);
/* Self-test hook: the library's own transcendentals (csrc/vslam_libm.h -- the camera model's atan / tan, jni/ATANCamera.h:136-150,
 * and the sin / cos / asin / acos of SE3 exp and ln, jni/RT.h:134-214, 318-383) over n arguments, on the device or
 * (on_host != 0) as compiled for the host.  fn: 0 sin, 1 cos, 2 tan, 3 atan, 4 asin, 5 acos, 6 sqrt, 7 reciprocal.  Both sides
 * must return the same bits: one ulp of difference in a projection flips template pixels (jni/vision/ImageHandler.cpp:12-19). */
int vslam_eval_transcendental(int fn, int n, const double* x, double* y, int on_host);

/* ---- frame front-end ------------------------------------------------------------------- */

/* KeyFrame::MakeKeyFrame_Lite (jni/KeyFrame.cc:5-51) for all n_streams frames at once:
 * 4-level pyramid, FAST-10 per level, raster-ordered corner lists and row LUTs, all on device.
 * gray: n_streams images, image s at gray + s*stream_stride, rows row_stride bytes apart;
 * on_device != 0 means gray is device memory (borrowed until the next front-end call; with vslam_params.bootstrap until the
 * frame AFTER the next has been enqueued and this one's work has finished: the trail tracker reads the previous frame's level 0
 * in place, jni/Tracker.cc:294-346).  Asynchronous on the system's stream. */
int vslam_make_keyframe_lite(vslam_system* sys, const uint8_t* gray, size_t row_stride,
                             size_t stream_stride, int on_device);

/* fast_nonmax (jni/vision/cvfast.cpp:9395-9400) on the current frame of every stream, all
 * levels: compute_fast_score_old + nonmax_suppression -> vMaxCorners. */
int vslam_fast_nonmax(vslam_system* sys);

/* read-back of the current frame (synchronises) */
int vslam_read_level_image(vslam_system* sys, int stream, int level, uint8_t* dst, size_t dst_stride);
int vslam_read_corners(vslam_system* sys, int stream, int level, uint32_t* corners /* x | y<<16 */,
                       int cap, int* n);
int vslam_read_row_lut(vslam_system* sys, int stream, int level, int* lut /* height>>level ints */);
int vslam_read_max_corners(vslam_system* sys, int stream, int level, uint32_t* corners, int* scores,
                           int cap, int* n);

/* KeyFrame::MakeKeyFrame_Rest (jni/KeyFrame.cc:53-95) for the current frame of every stream: fast_nonmax on the four levels,
 * then the candidate list -- maximal corners inside the 10-px border whose Shi-Tomasi score (half-window 3,
 * jni/vision/ImageHandler.cpp:124-155) exceeds min_shi_tomasi_score (reference: 70, :57).  The SmallBlurryImage part of the
 * function belongs to the relocaliser and is not built. */
int vslam_make_keyframe_rest(vslam_system* sys, double min_shi_tomasi_score);
/* MapMaker::ThinCandidates (jni/MapMaker.cc:393-422) on all four levels of the current candidate lists, against the
 * measurements of `keyframe` of each stream (keyframe < 0: the tracker's measurements of the current frame, i.e. what
 * MapMaker::AddKeyFrame copies into the new keyframe). */
int vslam_thin_candidates(vslam_system* sys, int keyframe);
/* Candidate::irLevelPos (packed x | y<<16) and dSTScore of one level, raster order; *n = count (may exceed cap). */
int vslam_read_candidates(vslam_system* sys, int stream, int level, uint32_t* pos, double* score, int cap, int* n);
/* grow_map = 1: Level::vCorners (packed x | y << 16, raster order) of a stored keyframe -- the epipolar search's target list.
 * At most 16384 / 8192 / 4096 / 2048 corners per level are kept (a longer list is cut in raster order); *n = stored count. */
int vslam_get_keyframe_corners(vslam_system* sys, int stream, int keyframe, int level, uint32_t* corners, int cap, int* n);
/* use_sbi = 1: the SmallBlurryImage of the current frame (jni/SmallBlurryImage.cc:20-55: (w/16) x (h/16) u8 image and its
 * zero-mean blurred fp32 template) and rot8 = { mv6SBIRot[6] (jni/Tracker.cc:885-893), final ESM score, 0 }. */
int vslam_read_sbi(vslam_system* sys, int stream, uint8_t* small_img, float* tmpl, double rot8[8]);

/* ---- MiniPatch (jni/MiniPatch.cc), the primitives of the reference's trail tracking ------------- */
/* MiniPatch::SampleFromImage (:71-83): 9x9 patches around n integer positions of the current frame's level 0
 * of `stream`; ok[i] = 0 where the patch would leave the image (the reference asserts). Synchronous. */
int vslam_minipatch_sample(vslam_system* sys, int stream, int n, const int* pos_xy, uint8_t* patches /* n*81 */, int* ok);
/* MiniPatch::FindPatch (:35-68): for each patch, best raw SSD at the FAST corners inside the +-range box around
 * pos_xy[i]; on success pos_xy[i] is replaced by the corner and found[i] = 1 (SSD < max_ssd, jni/Tracker.cc:226). */
int vslam_minipatch_find(vslam_system* sys, int stream, int n, const uint8_t* patches, int* pos_xy, int range, int max_ssd,
                         int* found);

/* ---- map (jni/Map.h:21-34) ---------------------------------------------------------------- */
/* The reference fills its map through InitFromStereo / AddPointEpipolar (bootstrap and map growth: out of
 * scope / "next" rows); the feeder supplies a ground-truth map through these instead. */

/* KeyFrame (jni/KeyFrame.h:73-97): pose12 = R row-major (9) + t (3), camera-from-world; gray is host memory.
 * Builds the keyframe's pyramid on device.  Returns the keyframe index (>= 0) or a negative error. */
int vslam_map_add_keyframe(vslam_system* sys, int stream, const double pose12[12], int fixed, const uint8_t* gray,
                           size_t row_stride, double depth_mean, double depth_sigma);
/* the same for n keyframes of the stream at once: pose12 n x 12, fixed n, gray n images image_stride bytes apart, depth_mean_sigma n x 2.
 * Returns the index of the first one. */
int vslam_map_add_keyframes(vslam_system* sys, int stream, int n, const double* pose12, const int* fixed, const uint8_t* gray, size_t row_stride,
                            size_t image_stride, const double* depth_mean_sigma);
/* MapPoint (jni/MapPoint.h:22-69). Returns the point index or a negative error. */
int vslam_map_add_point(vslam_system* sys, int stream, const double pos[3], int src_keyframe, int src_level,
                        int ir_x, int ir_y, const double pixel_right_w[3], const double pixel_down_w[3]);
/* the same for n points at once (pos, pixel_right_w, pixel_down_w: 3n doubles; ir_xy: 2n ints). Returns the new point count. */
int vslam_map_add_points(vslam_system* sys, int stream, int n, const double* pos, const int* src_keyframe, const int* src_level,
                         const int* ir_xy, const double* pixel_right_w, const double* pixel_down_w);
/* Measurement (jni/KeyFrame.h:46-51): source 0 TRACKER 1 REFIND 2 ROOT 3 TRAIL 4 EPIPOLAR */
int vslam_map_add_measurement(vslam_system* sys, int stream, int keyframe, int point, int level,
                              const double root_pos[2], int subpix, int source);
/* the same for n measurements at once (arrays of n; root_pos 2n doubles) */
int vslam_map_add_measurements(vslam_system* sys, int stream, int n, const int* keyframe, const int* point,
                               const int* level, const double* root_pos, const int* subpix, const int* source);
int vslam_map_set_good(vslam_system* sys, int stream);           /* Map::bGood (jni/Map.h:33) */
/* map editing after the upload (a host-side map-maker, a loaded map, tests that re-synchronise the map to a reference):
 * MapPoint::v3WorldPos of the points [first, first + n) (pos3: 3n doubles) and KeyFrame::se3CfromW of one keyframe */
int vslam_map_set_point_positions(vslam_system* sys, int stream, int first, int n, const double* pos3);
int vslam_map_set_keyframe_pose(vslam_system* sys, int stream, int keyframe, const double pose12[12]);
int vslam_set_pose(vslam_system* sys, int stream, const double pose12[12]);
int vslam_set_velocity(vslam_system* sys, int stream, const double v6[6]);
/* Tracker::mnLastKeyFrameDropped (jni/Tracker.h:118; -20 after Reset, jni/Tracker.cc:59): frame number of the stream's last keyframe;
 * the tracker asks for the next one more than min_frames_between_kf frames later (jni/Tracker.cc:128) */
int vslam_set_last_keyframe_dropped(vslam_system* sys, int stream, int frame);

/* ---- tracking ------------------------------------------------------------------------------ */

typedef struct vslam_track_state {
  double pose[12];                 /* Tracker::GetCurrentPose, jni/Tracker.h:58 */
  double velocity[6];              /* mv6CameraVelocity */
  double msd_velocity, depth_mean, depth_sigma;
  int attempted[VSLAM_LEVELS], found[VSLAM_LEVELS];   /* manMeasAttempted / manMeasFound */
  int quality;                     /* 0 BAD, 1 DODGY, 2 GOOD (jni/Tracker.h:123) */
  int lost_frames, frame, did_coarse;
  int kf_added, n_keyframes, n_points, ba_accepted;
  long long n_zmssd, n_ba_trials;  /* counters: ZMSSD evaluations, LM trials */
} vslam_track_state;

/* Tracker::TrackFrame (jni/Tracker.cc:76-146) for one frame of every stream: MakeKeyFrame_Lite, motion
 * model, TrackMap, quality assessment, and -- when the tracker asks for a keyframe -- MapMaker::AddKeyFrame
 * (jni/MapMaker.cc:470-506) followed by one BundleAdjustRecent (jni/MapMaker.cc:801-851), all on device.
 * Same image arguments as vslam_make_keyframe_lite.  Asynchronous. */
int vslam_track_frame(vslam_system* sys, const uint8_t* gray, size_t row_stride, size_t stream_stride,
                      int on_device);
/* Tracker::TrackFrame stage by stage (per-kernel entry points for parity tests and benchmarks).  After
 * vslam_make_keyframe_lite, on the current frame of every stream:
 *   vslam_patch_search(sys, 0)  Tracker::ApplyMotionModel (jni/Tracker.cc:781-798), the potentially visible set with
 *                               TrackerData::Project / PatchFinder::CalcSearchLevelAndWarpMatrix (:369-392), the coarse selection
 *                               (:399-461) and Tracker::SearchForPoints (:629-674: MakeTemplateCoarseCont, FindPatchCoarse,
 *                               sub-pixel iterations) of the coarse set;
 *   vslam_pose_update(sys, 0)   the ten coarse Gauss-Newton iterations (:463-490): CalcJacobian, Tracker::CalcPoseUpdate (:683-774);
 *   vslam_patch_search(sys, 1)  the fine selection (:493-535) and SearchForPoints of the level-3 points and of the rest;
 *   vslam_pose_update(sys, 1)   the ten fine iterations (:543-577), measurement export and scene depth (:594-625),
 *                               UpdateMotionModel (:802-820), AssessTrackingQuality (:832-878), the keyframe decision (:128-132);
 *   vslam_finish_frame(sys)     MapMaker::AddKeyFrame + BundleAdjustRecent for the streams whose tracker asked for a keyframe.
 * vslam_track_frame is exactly this sequence.  All asynchronous; the tracker state between two stages is read with
 * vslam_get_state / vslam_get_point_tracks / vslam_get_template (pose = the tracker's current estimate) . */
int vslam_patch_search(vslam_system* sys, int stage);
int vslam_pose_update(vslam_system* sys, int stage);
int vslam_finish_frame(vslam_system* sys);
/* the JNI-equivalent per-frame entry: native_update (jni/jni_part.cpp:132-145): host gray image of stream 0..n-1,
 * synchronous; native_touchScreen (:120-124) = the spacebar of every stream: with vslam_params.bootstrap it starts the trails / runs
 * InitFromStereo on the next frame of the streams that have no map, otherwise (and once a map exists) it has no effect. */
int vslam_update(vslam_system* sys, const uint8_t* gray, size_t row_stride, size_t stream_stride);
int vslam_touch(vslam_system* sys);

int vslam_get_state(vslam_system* sys, int stream, vslam_track_state* out);
/* the same for the streams [first, first + n): out[n], one device-to-host copy */
int vslam_get_states(vslam_system* sys, int first, int n, vslam_track_state* out);
/* MapMaker::NeedNewKeyFrame (jni/MapMaker.cc:761-773: distance to the closest keyframe, scaled by the scene depth, against
 * max_kf_dist_wiggle_mult * mdWiggleScaleDepthNormalized) and MapMaker::IsDistanceToNearestKeyFrameExcessive (:1098-1101: > 10 * wiggle
 * scale) for the stream's current pose.  The tracker takes both decisions on device; these are the reference's public members. */
int vslam_need_new_keyframe(vslam_system* sys, int stream, int* need);
int vslam_distance_to_nearest_keyframe_excessive(vslam_system* sys, int stream, int* excessive);
/* Tracker::GetMessageForUser (jni/Tracker.cc:880-883): "Tracking Map, quality good. Found: a/b ... Map: nP, nKF" */
int vslam_get_message(vslam_system* sys, int stream, char* buf, size_t cap);
/* per map point of a stream (arrays of n_points): TrackerData::bFound/bSearched/nSearchLevel/bDidSubPix,
 * v2Found (L0), v2Image.  Any pointer may be NULL. Returns n_points. */
int vslam_get_point_tracks(vslam_system* sys, int stream, int* found, int* searched, int* level, int* subpix,
                           double* vfound, double* image, int cap);
int vslam_get_points(vslam_system* sys, int stream, double* pos3, int* bad, int* n_inlier, int* n_outlier, int cap);
int vslam_get_keyframe_pose(vslam_system* sys, int stream, int keyframe, double pose12[12]);
/* The reference's only on-disk format, MapMaker::GUICommandHandler("SaveMap") (jni/MapMaker.cc:1254-1286): writes
 * <dir>/map.dump -- per map point (bad ones are in the reference's trash list and not written) v3WorldPos the way Eigen
 * streams a Vector3d (one coefficient per line, right-aligned to a common width, 6 significant digits) followed by two
 * blanks and nSourceLevel -- and <dir>/keyframes/<i>.info -- se3CfromW the way jni/RT.h:303-312 streams it (three lines
 * "r0 r1 r2 t") plus the blank line of the trailing endl.  The directories must exist.  Returns the number of points written. */
int vslam_save_map(vslam_system* sys, int stream, const char* dir);
int vslam_get_keyframe_measurements(vslam_system* sys, int stream, int keyframe, int* point, int* level,
                                    double* root_pos, int* source, int cap);
int vslam_get_template(vslam_system* sys, int stream, int point, uint8_t* tmpl /* P*P */, int* sum, int* sumsq,
                       int* bad);
/* the same for the points [first, first + n): tmpl n * P * P bytes; sum, sumsq, bad, have n ints each (any may be NULL) */
int vslam_get_templates(vslam_system* sys, int stream, int first, int n, uint8_t* tmpl, int* sum, int* sumsq, int* bad, int* have);
/* Size of the last bundle-adjustment problem MapMaker::BundleAdjust (jni/MapMaker.cc:854-960) assembled for the stream:
 * out[0..5] = cameras, adjustable cameras, points, measurements added, LM trials (mnCounter), accepted steps. */
int vslam_get_bundle_stats(vslam_system* sys, int stream, int out[6]);
/* Counters of the map-maker's idle jobs (vslam_params.idle_iterations, MapMaker::run jni/MapMaker.cc:94-117): out[0..5] = measurements
 * added by ReFindNewlyMade, by ReFindFromFailureQueue, BundleAdjustAll calls, idle BundleAdjustRecent calls, failure-queue length,
 * new-queue length. */
int vslam_get_idle_stats(vslam_system* sys, int stream, int out[6]);
/* One idle job of MapMaker::run (jni/MapMaker.cc:94-117) for every stream, between frames: 0 BundleAdjustRecent if not converged
 * (:97-98), 1 ReFindNewlyMade (:102-103), 2 BundleAdjustAll if not converged (:107-108), 3 every 20th call
 * ReFindFromFailureQueue (:112-113); each followed by HandleBadPoints (:117).  Needs idle_iterations != 0 at creation. */
int vslam_mapmaker_idle_job(vslam_system* sys, int job);

/* ---- map bootstrap (vslam_params.bootstrap) ----
 * Tracker::mbUserPressedSpacebar (jni/Tracker.h:138) of one stream, or of every stream (stream < 0): consumed by the next frame. */
int vslam_press_spacebar(vslam_system* sys, int stream);
/* MapMaker::InitFromStereo(KeyFrame&, KeyFrame&, vector<pair<ImageRef, ImageRef>>&, mySE3&) (jni/MapMaker.h:38, jni/MapMaker.cc:204-376) for a caller that
 * owns the two frames and the matches: host gray images of the first and the second keyframe, n matches as (x, y in the first, x, y in the second)
 * at level 0.  The first image becomes the first keyframe, the matches take the place of the trails, the second image is the frame the map is made
 * in.  Returns 1 and the tracker's pose (pose12_out may be NULL) when the map is good, 0 when InitFromStereo gave up (then a new attempt may
 * follow), a negative error otherwise.  One-stream systems created with bootstrap = 1 that have no map yet.  Synchronous. */
int vslam_init_from_stereo(vslam_system* sys, const uint8_t* gray_first, const uint8_t* gray_second, size_t row_stride, int n_matches,
                           const int* matches_xyxy, double pose12_out[12]);
/* the seed that stands in for the reference's rand() state in HomographyInit's MLESAC and CalcPlaneAligner's RANSAC (default 1) */
int vslam_set_boot_seed(vslam_system* sys, int stream, unsigned seed);
/* out[0..5] = mnInitialStage (0 not started, 1 trails running, 2 complete), trails alive, InitFromStereo succeeded, homography inliers,
 * map points made from the stereo pair, map good */
int vslam_get_init_info(vslam_system* sys, int stream, int out[6]);
/* the trails (jni/Tracker.h Trail): irInitialPos x, y, irCurrentPos x, y per trail, in list order */
int vslam_get_trails(vslam_system* sys, int stream, int* out4, int cap, int* n);
/* Reads the directory vslam_save_map wrote (MapMaker "SaveMap", jni/MapMaker.cc:1254-1286: map.dump, keyframes/<i>.info) back: point
 * positions + source levels, keyframe poses as R (9, row-major) then t (3).  Arrays may be null to count only. */
/* Writes the state such a dump holds (positions of the good points, keyframe poses; 6 significant digits) back into the map it was
 * saved from: same keyframes and points, checked by count and source level.  Images, templates and measurements are not part of the
 * reference's format, so this restores an estimate, it does not build a map. */
int vslam_load_map(vslam_system* sys, int stream, const char* dir);
int vslam_read_map_dump(const char* dir, double* pos3, int* level, int point_cap, int* n_points, double* pose12, int kf_cap, int* n_keyframes);

/* ---- measurement: HIP-event time per stage of vslam_track_frame, on the system's own stream ---- */
#define VSLAM_N_STAGES 14
/* stages: 0 pyr_fast0, 1 fast_lvl, 2 compact, 3 pvs, 4 plan_coarse, 5 search_coarse, 6 pose_coarse, 7 plan_fine,
 * 8 search_fine, 9 pose_fine, 10 add_keyframe, 11 ba_assemble, 12 ba_compute, 13 ba_writeback */
const char* vslam_stage_name(int stage);
int vslam_profile_begin(vslam_system* sys, int max_frames);
/* synchronises; stage_ms[VSLAM_N_STAGES] = summed milliseconds over the recorded frames */
int vslam_profile_end(vslam_system* sys, double* stage_ms, int* n_frames);
/* launches[VSLAM_N_STAGES] = how many launches of each stage the last vslam_profile_begin/end pair recorded (every frame for all stages
 * but ba_compute with the asynchronous map-maker, which is launched once per batch of frames) */
int vslam_profile_launches(vslam_system* sys, int* launches);
/* What the Bundle::Compute launches (k_ba_compute) of the last vslam_profile_begin/end window actually ran, counted on the device by
 * the launches themselves: stats[0] problems, [1] LM trials (Do_LM_Step's inner trials, jni/Bundle.cc:327-501), and the
 * trial-weighted sums SURVEY.md 8(d)'s formulas need -- [2] trials x measurements, [3] trials x cameras, [4] trials x points,
 * [5] trials x points x C(adjustable cameras, 2), [6] trials x (6 x adjustable cameras)^3 -- [7] launches.  After vslam_profile_end. */
int vslam_profile_ba_stats(vslam_system* sys, unsigned long long stats[8]);
/* the same counters over every Bundle::Compute launch of the system since its creation (the last 1024); stats[7] = launches that ran a problem */
int vslam_get_ba_launch_totals(vslam_system* sys, unsigned long long stats[8]);
/* HIP-event time of the last host-driven vslam_bundle_adjust_recent / vslam_bundle_adjust_all on the system's stream:
 * ms[0] selection + assembly (k_ba_select, k_ba_assemble), ms[1] Bundle::Compute (k_ba_compute, one workgroup per stream's
 * problem), ms[2] write-back + HandleBadPoints; stats (may be NULL): the launch's counters as vslam_profile_ba_stats. Synchronises. */
int vslam_get_mapmaker_timing(vslam_system* sys, double ms[3], unsigned long long stats[8]);

/* ---- mapping ------------------------------------------------------------------------------- */
/* MapMaker::BundleAdjustRecent / BundleAdjustAll (jni/MapMaker.cc:801-851, 776-798) on every stream, followed by
 * HandleBadPoints (:140-164); what the reference's map-maker thread loop (:80-123) would run next. */
int vslam_bundle_adjust_recent(vslam_system* sys);
int vslam_bundle_adjust_all(vslam_system* sys);
/* MapMaker::AddKeyFrame (jni/MapMaker.cc:470-478) called from the host: the current frame of `stream` (all streams
 * if stream < 0) becomes a keyframe now, followed by the same BundleAdjustRecent + HandleBadPoints as the
 * tracker-driven path -- on the system's own stream, finished before the next frame, also with the asynchronous map-maker
 * (whose adjustments in flight are collected first).  Timed like vslam_bundle_adjust_recent (vslam_get_mapmaker_timing).
 * Needs a tracked current frame. */
int vslam_add_keyframe(vslam_system* sys, int stream);

/* ---- stand-alone Bundle (jni/Bundle.h:111-121), batched: n_problems independent problems ---- */
typedef struct vslam_bundle vslam_bundle;
int vslam_bundle_create(const vslam_params* p, int n_problems, int max_cameras, int max_points, int max_meas,
                        vslam_bundle** out);
int vslam_bundle_destroy(vslam_bundle* b);
int vslam_bundle_add_camera(vslam_bundle* b, int problem, const double pose12[12], int fixed);   /* Bundle::AddCamera */
int vslam_bundle_add_point(vslam_bundle* b, int problem, const double pos[3]);                   /* Bundle::AddPoint  */
int vslam_bundle_add_meas(vslam_bundle* b, int problem, int cam, int point, const double pos[2],
                          double sigma_squared);                                                  /* Bundle::AddMeas   */
/* Bundle::AddCamera / AddPoint / AddMeas for a whole problem at once, replacing what it held: pose12 n_cams x 12, fixed n_cams,
 * pos3 n_pts x 3; cam, point, sigma_squared n_meas and xy n_meas x 2 in AddMeas order. */
int vslam_bundle_set_problem(vslam_bundle* b, int problem, int n_cams, const double* pose12, const int* fixed, int n_pts, const double* pos3,
                             int n_meas, const int* cam, const int* point, const double* xy, const double* sigma_squared);
/* HIP-event milliseconds of the last vslam_bundle_compute launch (its inputs were resident before the first event) and the counters
 * the launch kept on the device (stats[8] as vslam_profile_ba_stats; may be NULL).  Synchronises. */
int vslam_bundle_get_timing(vslam_bundle* b, double* ms, unsigned long long stats[8]);
/* Bundle::Compute for every problem (one launch), each from the cameras / points / measurements the caller added -- a second call
 * starts again from those, not from the first call's result; asynchronous. accepted[problem] read by _get_result. */
int vslam_bundle_compute(vslam_bundle* b);
int vslam_bundle_synchronize(vslam_bundle* b);
int vslam_bundle_get_result(vslam_bundle* b, int problem, int* accepted, int* converged, double* sigma_squared,
                            double* lambda, long long* trials);
int vslam_bundle_get_camera(vslam_bundle* b, int problem, int n, double pose12[12]);             /* Bundle::GetCamera */
int vslam_bundle_get_point(vslam_bundle* b, int problem, int n, double pos[3]);                  /* Bundle::GetPoint  */
int vslam_bundle_get_outlier_meas(vslam_bundle* b, int problem, int* pc_pairs, int cap);         /* GetOutlierMeasurements */
int vslam_bundle_get_outlier_points(vslam_bundle* b, int problem, int* idx, int cap);            /* GetOutliers */

#ifdef __cplusplus
}
#endif
#endif
