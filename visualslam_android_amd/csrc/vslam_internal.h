// Internal layout of a vslam_system: every buffer lives in HBM for the lifetime of the handle.
// gfx950 only.  Host-side bookkeeping + device pointers; kernels receive POD views.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/vslam_c.h"
#include "dev_math.h"

#define NLEV VSLAM_LEVELS

void vslam_set_error(const char* fmt, ...);

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      vslam_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return VSLAM_E_HIP;                                                                    \
    }                                                                                        \
  } while (0)

// Geometry of one pyramid level (same for every stream).
struct LevelGeom {
  int w, h;        // level size: (W >> l, H >> l)            (jni/KeyFrame.cc:21)
  int pitch;       // bytes between rows of the internal level image (multiple of 64)
  int nchunk;      // ceil(w / 64): 64-bit corner-mask words per row
  int cap;         // corner capacity per stream
  int thr;         // FAST threshold                          (jni/KeyFrame.cc:32-39)
};

// Device view of the current frame of all streams (front-end outputs).
struct FrameDev {
  // level images: img[l] + s*img_sstride[l] + y*pitch.  Level 0 may point into caller memory.
  const uint8_t* img[NLEV];
  size_t img_sstride[NLEV];
  int img_pitch[NLEV];
  unsigned long long* cmask[NLEV];  // [S][h][nchunk] corner bit masks (bit i of word c = pixel 64c+i)
  int* rowcnt[NLEV];                // [S][h]
  int* rowlut[NLEV];                // [S][h+1]  rowlut[y] = first corner index with y' >= y; [h] = n
  uint32_t* corners[NLEV];          // [S][cap]  x | y<<16, raster order
  int* ncorners;                    // [S][NLEV]
  int* overflow;                    // [1] set when any corner list hit its capacity
  // non-max products (MakeKeyFrame_Rest)
  int* scores[NLEV];                // [S][cap]
  uint32_t* maxcorners[NLEV];       // [S][cap]
  int* nmax;                        // [S][NLEV]
  // SmallBlurryImage of the frame (jni/SmallBlurryImage.h) and the rotation prior computed against the previous frame's
  uint8_t* sbi_small;               // [S][hs*ws]       mimSmall
  float* sbi_tmpl;                  // [S][hs*ws]       mimTemplate (zero-mean, blurred)
  float* sbi_jacs;                  // [S][hs*ws][2]    mimImageJacs
  double* sbi_rot;                  // [S][8]           mv6SBIRot (6), final ESM score, spare
};


// ---- map + tracker state (all device resident) -------------------------------------------------------------------
#define POSE_WS_COMPS 12
#define TMPL_PITCH 128      // bytes reserved per cached template (11 x 11 = 121)

struct MapPointDev {        // MapPoint, jni/MapPoint.h:22-69
  double pos[3], right[3], down[3];   // v3WorldPos, v3PixelRight_W, v3PixelDown_W
  int src_kf, src_level, irx, iry;    // pPatchSourceKF, nSourceLevel, irCenter
  int bad, n_in, n_out, n_meas_kfs;   // bBad, nMEstimatorInlier/OutlierCount, |MapMakerData::sMeasurementKFs|
};

#define TDF_IN_IMAGE 1
#define TDF_FOUND 2
#define TDF_SEARCHED 4
#define TDF_SUBPIX 8
#define TDF_TMPL_BAD 16
#define TDF_HAVE_LAST 32

struct TrackData {          // TrackerData (jni/TrackerData.h:36-66) + persistent PatchFinder state (jni/PatchFinder.h:96-128)
  double cam[3], image[2], derivs[4];
  double vfound[2], sqrt_inv_noise;      // the 2x6 Jacobian and the residual live in k_pose's registers only
  double warp_inv[4], last_warp[4];
  int tsum, tsumsq;                      // nSearchLevel and the TDF_* flags live in MapDev::pt_level / pt_flags (coalesced)
};

struct MeasDev {            // Measurement, jni/KeyFrame.h:46-51 (one slot per keyframe x map point)
  double root[2];
  signed char valid, level, subpix, source;
  int pad;
};

#define BOOT_MAX_TRAILS 1000   // MaxInitialTrails, jni/Tracker.cc:305

struct TrackerState {       // Tracker members, jni/Tracker.h:77-150 (+ MapMaker flags used by the BA driver)
  Pose pose_final, start_pose, pose_cur;
  double velocity[6];
  double msd_vel, depth_mean, depth_sigma, wiggle_depth_norm;
  int frame, last_kf_dropped, lost_frames, quality;
  int attempted[NLEV], found[NLEV];
  int did_coarse, just_recovered, map_good, kf_pending;
  int n_points, n_kf;
  int pvs_count[NLEV], pvs_head[NLEV];
  int n_coarse, n_search, n_iter, n_l3;
  int coarse_range, fine_range, coarse_found;
  int ba_accepted, kf_added, ba_converged_recent, ba_converged_full;
  int ba_countdown;         // > 0: a bundle adjustment is in flight, its results are applied when this reaches 0
  unsigned long long n_zmssd, n_ba_trials;
  // the map-maker's idle jobs (vslam_params.idle_iterations)
  int newq_head;            // mqNewQueue = the points [newq_head, n_points): made by AddPointEpipolar, not yet seen by ReFindNewlyMade
  int fq_n;                 // mvFailureQueue length
  int idle_count;           // evaluations of the lowest-priority job's condition (rand() % 20 == 0 made deterministic: every 20th)
  int idle_do_fail;         // this pass runs ReFindFromFailureQueue
  int n_refound_new, n_refound_failed, n_ba_all, n_ba_recent_idle;   // statistics
  // map bootstrap (vslam_params.bootstrap; boot.hip): Tracker::TrackForInitialMap, jni/Tracker.cc:247-288
  int init_stage;           // mnInitialStage: 0 TRAIL_TRACKING_NOT_STARTED, 1 STARTED, 2 COMPLETE
  int spacebar;             // mbUserPressedSpacebar
  int n_trails, trail_buf;  // mlTrails (double-buffered: a frame's survivors are compacted into the other buffer)
  int boot_action;          // this frame: 1 TrailTracking_Start, 2 TrailTracking_Advance
  int boot_run;             // InitFromStereo is running for this stream (gates its kernels)
  int boot_ok, n_hom_inliers, n_init_points;
  unsigned boot_seed;       // stands in for the reference's rand() state
  int boot_host_matches;    // vslam_init_from_stereo: the trails are the caller's matches; this frame's TrailTracking_Advance does not search
};

struct TrackParams {        // device copy of the tunables the kernels read
  CamModel cam;
  int P;                    // patch size
  int max_ssd;              // 500 * P * P (jni/PatchFinder.cc:19-20)
  int max_patches, coarse_min, coarse_max, coarse_range, coarse_subpix_its, coarse_disabled, fine_subpix_its;
  double coarse_min_vel, wls_prior;
  int min_frames_between_kf; double max_kf_dist_wiggle_mult, wiggle_scale;
  int ba_max_iterations; double ba_convergence_limit, ba_min_sigma2; int ba_window, ba_min_keyframes;
  int quirks;
  int max_points, max_keyframes;
  int ba_delay;             // vslam_params.ba_delay_frames
  int ba_batch;             // vslam_params.ba_batch_frames (>= 1)
  int ba_sum_order;         // vslam_params.ba_sum_order
  int grow_map;             // vslam_params.grow_map
  int idle;                 // vslam_params.idle_iterations
  int fq_cap;               // capacity of a stream's failure queue
  double one_pixel_dist;    // ATANCamera::OnePixelDist, jni/ATANCamera.cc:86-91
  int kcap[NLEV];           // capacity of a keyframe's stored corner list per level (grow_map)
};

struct MapDev {             // device pointers of the map + tracker of all streams
  MapPointDev* pts;         // [S][max_points]
  TrackData* td;            // [S][max_points]
  uint8_t* tmpl;            // [S][max_points][TMPL_PITCH]
  MeasDev* kf_meas;         // [S][max_keyframes][max_points]
  MeasDev* cur_meas;        // [S][max_points]            mCurrentKF.mMeasurements
  Pose* kf_pose;            // [S][max_keyframes]
  int* kf_fixed;            // [S][max_keyframes]
  double* kf_depth;         // [S][max_keyframes][2]
  uint8_t* kf_img[NLEV];    // [S][max_keyframes][h*pitch]
  uint32_t* kf_corners[NLEV];   // [S][max_keyframes][kcap_l]  Level::vCorners of the keyframes (grow_map only: epipolar search)
  int* kf_ncorners;         // [S][max_keyframes][NLEV]
  unsigned long long* never_retry;   // [S][max_points][2]  MapMakerData::sNeverRetryKFs as a bit set over the keyframes (idle jobs only)
  int2* fq;                 // [S][fq_cap]  mvFailureQueue: (keyframe, point)
  // map bootstrap (vslam_params.bootstrap): the trails and InitFromStereo's work arrays
  uint8_t* trail_patch;     // [S][2][BOOT_MAX_TRAILS][81]  Trail::mPatch
  int* trail_pos;           // [S][2][BOOT_MAX_TRAILS][4]   irInitialPos, irCurrentPos
  double* boot_match;       // [S][BOOT_MAX_TRAILS][8]      HomographyMatch
  int* boot_inl;            // [S][BOOT_MAX_TRAILS]
  double* boot_ws;          // [S][max(3 * max_points, BOOT_MAX_TRAILS)]
  TrackerState* st;         // [S]
  int* pvs_list;            // [S][NLEV][max_points]
  int2* search_list;        // [S][max_points]  (point index, sub-pixel iterations)
  int* pt_level;            // [S][max_points]  PatchFinder::mnSearchLevel of the frame, -1 = not in the PVS (read by every planning pass)
  int* pt_flags;            // [S][max_points]  TDF_* flags
  int* iter_list;           // [S][max_points]  vIterationSet
  double* pose_ws;          // [S][POSE_WS_COMPS][max_points]  k_pose working set, component-major, indexed by iteration-set entry
  int* pose_wsi;            // [S][2][max_points]  flags, map point index
};

struct vslam_system {
  vslam_params p;
  int S;
  hipStream_t stream;
  LevelGeom geom[NLEV];
  FrameDev fr;                 // view of the CURRENT frame's front-end products (= frbuf[fr_idx])
  uint8_t* d_lvl[NLEV];        // owned level images of the current buffer (level 0 = staging copy for host input)
  // Front-end products are double-buffered and built on their own HIP stream so MakeKeyFrame_Lite of frame t+1
  // overlaps TrackMap / pose / bundle adjustment of frame t (which are latency-bound and leave the CUs mostly idle).
  FrameDev frbuf[2];
  uint8_t* d_lvl_buf[2][NLEV];
  int fr_idx = 0;
  hipStream_t fe_stream = nullptr;
  hipEvent_t ev_fe_done[2] = {nullptr, nullptr}, ev_track_done[2] = {nullptr, nullptr};
  // asynchronous map-maker (ba_delay_frames > 0): Bundle::Compute runs on ba_stream beside the following frames
  hipStream_t ba_stream = nullptr;   // the map-maker stream of the frame being enqueued (= ba_streams[frame_no % ba_streams.size()])
  std::vector<hipStream_t> ba_streams;   // a bundle adjustment outlasts a frame: successive keyframe frames' launches overlap on a ring of streams
  double* grow_implane = nullptr;   // [S][kcap_0][2]: the epipolar search's target corners on the image plane (mapgrow.hip)
  int n_cu = 0;             // compute units of the device (asynchronous map-maker: size of the background BA grid)
  std::vector<hipEvent_t> ev_asm, ev_ba;   // rings of ba_delay + 2 events, indexed by batch number
  long frame_no = 0;
  long ba_token = 0;           // ba_run calls so far: k_ba_select marks the problems it wants with the call's token, k_ba_assemble takes only those
  // batches of the asynchronous map-maker: the problems assembled in ba_batch consecutive frames share one work list and one launch
  long ba_batch_id = 0;        // the open batch
  int ba_batch_fill = 0;       // frames assembled into it so far
  std::vector<long> frame_batch;   // ring [ba_delay + 2]: the batch a frame's keyframes were assembled into
  std::vector<int> prof_ba_launched;    // per profiled frame: 0, or 1 + the launch record (BaPool::lstat) of the k_ba_compute it launched
  long ba_launch_no = 0;       // k_ba_compute launches of this system so far (ring index of the launch records)
  hipEvent_t ev_mm[4] = {nullptr, nullptr, nullptr, nullptr};   // host-driven BundleAdjustRecent / All: before select+assemble, compute, write-back, after
  int mm_lrec = -1;            // launch record of the last host-driven call
  std::vector<void*> allocs;   // everything to hipFree
  bool have_frame;
  bool frame_open = false;  // stage-wise TrackFrame (vslam_patch_search ... vslam_finish_frame) in progress
  bool have_sbi;            // a SmallBlurryImage of a previous frame exists (mpSBILastFrame)
  // KeyFrame::Level::vCandidates of the current frame (jni/KeyFrame.h:62-70), filled by vslam_make_keyframe_rest
  uint32_t* cand[NLEV]; double* cand_score[NLEV]; int* ncand; bool have_candidates;
  bool boot_key_pressed = false;   // vslam_press_spacebar since the last frame: that frame launches the start / InitFromStereo pipelines
  TrackParams tp;
  MapDev map;
  void* ba_ws;                 // bundle-adjustment workspace (ba.hip)
  // per-stage HIP-event timing of vslam_track_frame (vslam_profile_begin/end)
  std::vector<hipEvent_t> prof_ev;
  int prof_cap = 0, prof_frame = 0;
  bool prof_on = false;
};

#define PROF_MARKS (VSLAM_N_STAGES + 3)   // marks 0..2 + PROF_FE_END on the front-end stream, 3..VSLAM_N_STAGES on the main stream
#define PROF_FE_END (VSLAM_N_STAGES + 1)
#define PROF_BA_END (VSLAM_N_STAGES + 2)  // asynchronous map-maker: marks 12 and PROF_BA_END live on the BA stream
static inline void prof_mark(vslam_system* sys, int k) {
  if (!(sys->prof_on && sys->prof_frame < sys->prof_cap)) return;
  hipStream_t st = sys->stream;
  if (k < 3 || k == PROF_FE_END) st = sys->fe_stream;
  else if (sys->tp.ba_delay > 0 && (k == 12 || k == PROF_BA_END)) st = sys->ba_stream;
  (void)hipEventRecord(sys->prof_ev[(size_t)sys->prof_frame * PROF_MARKS + k], st);
}

// frontend.hip
int fe_make_keyframe_lite(vslam_system* sys, const uint8_t* gray, size_t row_stride, size_t stream_stride,
                          int on_device);
int fe_fast_nonmax(vslam_system* sys);
int fe_keyframe_corners(vslam_system* sys, int s, int kf);   // FAST corners of a stored keyframe into MapDev::kf_corners (map upload)
int fe_keyframe_rest_gated(vslam_system* sys);               // non-max + candidates of the current frame for the streams with kf_pending
int fe_thin_new_keyframe(vslam_system* sys, int level);      // ThinCandidates(new keyframe, level) for the streams with kf_pending
int grow_alloc(vslam_system* sys);
int grow_on_keyframe(vslam_system* sys);                      // AddSomeMapPoints(3, 0, 1, 2) for the streams with kf_pending
int grow_idle_refind(vslam_system* sys, int mode);            // idle jobs: 0 ReFindNewlyMade, 1 ReFindFromFailureQueue (gated per stream on device)
int mm_idle(vslam_system* sys);
int boot_alloc(vslam_system* sys);
int boot_frame(vslam_system* sys);                                        // TrackForInitialMap for the streams without a map (vslam_params.bootstrap)
int grow_copy_corners(vslam_system* sys);                                 // Level::vCorners of the current frame into the keyframe slot n_kf (streams with kf_pending)
int grow_levels(vslam_system* sys, const int* order, int n);              // ThinCandidates + AddPointEpipolar per level, in this order (streams with kf_pending)
int ba_launch_add_keyframe(vslam_system* sys);                            // k_add_keyframe for the streams with kf_pending
int mm_idle_job(vslam_system* sys, int job);                               // vslam_params.idle_iterations passes through MapMaker::run's idle jobs
int fe_sbi(vslam_system* sys, const FrameDev& last);   // k_sbi on the front-end stream: this frame's SBI + rotation prior against `last`
void cam_fill(CamModel& c, const double cam5[5], double width, double height, int quirks);
int fe_make_keyframe_rest(vslam_system* sys, double min_score);
int fe_thin_candidates(vslam_system* sys, int keyframe);
// track.hip
void trk_fill_params(const vslam_params& p, TrackParams& t);
int trk_alloc(vslam_system* sys);
int trk_track_map(vslam_system* sys);
int trk_search_stage(vslam_system* sys, int stage);
int trk_pose_stage(vslam_system* sys, int stage);
// ba.hip
int ba_alloc(vslam_system* sys);
int ba_add_keyframe_and_adjust(vslam_system* sys);
int ba_run(vslam_system* sys, int mode, bool host_driven_keyframe = false);
int ba_frame_start(vslam_system* sys);
int ba_sync_streams(vslam_system* sys);   // launches an open batch first, then   // host wait for every map-maker stream   // asynchronous map-maker: apply the results that are due at this frame
// map.hip
int map_init_states(vslam_system* sys);

