// Map upload, per-frame entry points (Tracker::TrackFrame / JNI-equivalent update) and read-back of the C ABI.
#include "vslam_internal.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#define CHK_STREAM(sys, s) do { if (!(sys) || (s) < 0 || (s) >= (sys)->S) { vslam_set_error("bad system/stream"); return VSLAM_E_INVALID; } } while (0)

static int get_state(vslam_system* sys, int s, TrackerState* st) {
  HIPCHK(hipMemcpyAsync(st, sys->map.st + s, sizeof(TrackerState), hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  return VSLAM_OK;
}
static int put_state(vslam_system* sys, int s, const TrackerState* st) {
  HIPCHK(hipMemcpyAsync(sys->map.st + s, st, sizeof(TrackerState), hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  return VSLAM_OK;
}

// one pyramid level of a stored keyframe: (a+b+c+d+2)>>2 (jni/KeyFrame.cc:19-23, see frontend.hip)
__global__ void k_halve_plain(const uint8_t* src, int sp, uint8_t* dst, int dp, int dw, int dh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dw || y >= dh) return;
  const uint8_t* r0 = src + (size_t)(2 * y) * sp + 2 * x;
  const uint8_t* r1 = r0 + sp;
  dst[(size_t)y * dp + x] = (uint8_t)((r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2);
}

extern "C" int vslam_map_add_keyframe(vslam_system* sys, int s, const double pose12[12], int fixed, const uint8_t* gray,
                                      size_t row_stride, double depth_mean, double depth_sigma) {
  CHK_STREAM(sys, s);
  if (!pose12 || !gray || (int)row_stride < sys->geom[0].w) { vslam_set_error("map_add_keyframe: bad argument"); return VSLAM_E_INVALID; }
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  const int K = sys->p.max_keyframes;
  if (st.n_kf >= K) { vslam_set_error("keyframe capacity %d reached", K); return VSLAM_E_CAPACITY; }
  const int k = st.n_kf;
  const LevelGeom* g = sys->geom;
  uint8_t* lvl[NLEV];
  for (int l = 0; l < NLEV; l++) lvl[l] = sys->map.kf_img[l] + ((size_t)s * K + k) * ((size_t)g[l].pitch * g[l].h);
  HIPCHK(hipMemcpy2DAsync(lvl[0], g[0].pitch, gray, row_stride, g[0].w, g[0].h, hipMemcpyHostToDevice, sys->stream));
  for (int l = 1; l < NLEV; l++)
    hipLaunchKernelGGL(k_halve_plain, dim3((g[l].w + 255) / 256, g[l].h), dim3(256), 0, sys->stream, lvl[l - 1], g[l - 1].pitch, lvl[l], g[l].pitch, g[l].w, g[l].h);
  Pose p;
  for (int i = 0; i < 9; i++) p.R[i] = pose12[i];
  for (int i = 0; i < 3; i++) p.t[i] = pose12[9 + i];
  const double dd[2] = {depth_mean, depth_sigma};
  const int fx = fixed ? 1 : 0;
  HIPCHK(hipMemcpyAsync(sys->map.kf_pose + (size_t)s * K + k, &p, sizeof(Pose), hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemcpyAsync(sys->map.kf_fixed + (size_t)s * K + k, &fx, sizeof(int), hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemcpyAsync(sys->map.kf_depth + ((size_t)s * K + k) * 2, dd, sizeof(dd), hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemsetAsync(sys->map.kf_meas + ((size_t)s * K + k) * sys->p.max_points, 0, sizeof(MeasDev) * sys->p.max_points, sys->stream));
  if (sys->p.grow_map || sys->p.idle_iterations != 0) { r = fe_keyframe_corners(sys, s, k); if (r) return r; }   // Level::vCorners, needed as an epipolar-search target
  st.n_kf = k + 1;
  r = put_state(sys, s, &st); if (r) return r;
  return k;
}

// the same level for n images at once (blockIdx.z = image; images sstride / dstride bytes apart)
__global__ void k_halve_batch(const uint8_t* src, int sp, size_t sstride, uint8_t* dst, int dp, size_t dstride, int dw, int dh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dw || y >= dh) return;
  const uint8_t* r0 = src + (size_t)blockIdx.z * sstride + (size_t)(2 * y) * sp + 2 * x;
  const uint8_t* r1 = r0 + sp;
  dst[(size_t)blockIdx.z * dstride + (size_t)y * dp + x] = (uint8_t)((r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2);
}

// vslam_map_add_keyframe for n keyframes of one stream at once: one pyramid launch per level and one clear of the measurement rows for
// all of them (a map upload of eight keyframes is 4 kernel launches instead of 32: set-up of thousands of streams, and profiler passes
// that serialise every dispatch).  gray: n images image_stride bytes apart.  Returns the index of the first new keyframe.
extern "C" int vslam_map_add_keyframes(vslam_system* sys, int s, int n, const double* pose12, const int* fixed, const uint8_t* gray, size_t row_stride,
                                       size_t image_stride, const double* depth_mean_sigma) {
  CHK_STREAM(sys, s);
  if (n < 1 || !pose12 || !fixed || !gray || !depth_mean_sigma || (int)row_stride < sys->geom[0].w) { vslam_set_error("map_add_keyframes: bad argument"); return VSLAM_E_INVALID; }
  if (sys->p.grow_map || sys->p.idle_iterations != 0) {                 // keyframe corner lists are made one keyframe at a time
    int first = -1;
    for (int i = 0; i < n; i++) {
      const int k = vslam_map_add_keyframe(sys, s, pose12 + 12 * i, fixed[i], gray + (size_t)i * image_stride, row_stride, depth_mean_sigma[2 * i], depth_mean_sigma[2 * i + 1]);
      if (k < 0) return k;
      if (i == 0) first = k;
    }
    return first;
  }
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  const int K = sys->p.max_keyframes;
  if (st.n_kf + n > K) { vslam_set_error("keyframe capacity %d reached", K); return VSLAM_E_CAPACITY; }
  const int k = st.n_kf;
  const LevelGeom* g = sys->geom;
  uint8_t* lvl[NLEV]; size_t ls[NLEV];
  for (int l = 0; l < NLEV; l++) { ls[l] = (size_t)g[l].pitch * g[l].h; lvl[l] = sys->map.kf_img[l] + ((size_t)s * K + k) * ls[l]; }
  for (int i = 0; i < n; i++)
    HIPCHK(hipMemcpy2DAsync(lvl[0] + (size_t)i * ls[0], g[0].pitch, gray + (size_t)i * image_stride, row_stride, g[0].w, g[0].h, hipMemcpyHostToDevice, sys->stream));
  for (int l = 1; l < NLEV; l++)
    hipLaunchKernelGGL(k_halve_batch, dim3((g[l].w + 255) / 256, g[l].h, n), dim3(256), 0, sys->stream, lvl[l - 1], g[l - 1].pitch, ls[l - 1], lvl[l], g[l].pitch, ls[l], g[l].w, g[l].h);
  std::vector<Pose> p((size_t)n); std::vector<int> fx((size_t)n);
  for (int i = 0; i < n; i++) { for (int q = 0; q < 9; q++) p[i].R[q] = pose12[12 * i + q]; for (int q = 0; q < 3; q++) p[i].t[q] = pose12[12 * i + 9 + q]; fx[i] = fixed[i] ? 1 : 0; }
  HIPCHK(hipMemcpyAsync(sys->map.kf_pose + (size_t)s * K + k, p.data(), sizeof(Pose) * n, hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemcpyAsync(sys->map.kf_fixed + (size_t)s * K + k, fx.data(), sizeof(int) * n, hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemcpyAsync(sys->map.kf_depth + ((size_t)s * K + k) * 2, depth_mean_sigma, sizeof(double) * 2 * n, hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemsetAsync(sys->map.kf_meas + ((size_t)s * K + k) * sys->p.max_points, 0, sizeof(MeasDev) * sys->p.max_points * (size_t)n, sys->stream));
  st.n_kf = k + n;
  r = put_state(sys, s, &st); if (r) return r;         // (synchronises: the host arrays above are done with)
  return k;
}

extern "C" int vslam_map_add_point(vslam_system* sys, int s, const double pos[3], int src_keyframe, int src_level, int ir_x,
                                   int ir_y, const double right[3], const double down[3]) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  if (!pos || !right || !down || src_keyframe < 0 || src_keyframe >= st.n_kf || src_level < 0 || src_level >= NLEV) { vslam_set_error("map_add_point: bad argument"); return VSLAM_E_INVALID; }
  const int P = sys->p.max_points;
  if (st.n_points >= P) { vslam_set_error("map point capacity %d reached", P); return VSLAM_E_CAPACITY; }
  const int i = st.n_points;
  MapPointDev mp; memset(&mp, 0, sizeof(mp));
  for (int q = 0; q < 3; q++) { mp.pos[q] = pos[q]; mp.right[q] = right[q]; mp.down[q] = down[q]; }
  mp.src_kf = src_keyframe; mp.src_level = src_level; mp.irx = ir_x; mp.iry = ir_y;
  TrackData td; memset(&td, 0, sizeof(td));
  td.last_warp[0] = 9999.9; td.last_warp[3] = 9999.9;   // jni/PatchFinder.cc:23
  const int lvl0[2] = {-1, 0};
  HIPCHK(hipMemcpyAsync(sys->map.pt_level + (size_t)s * P + i, &lvl0[0], sizeof(int), hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemcpyAsync(sys->map.pt_flags + (size_t)s * P + i, &lvl0[1], sizeof(int), hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemcpyAsync(sys->map.pts + (size_t)s * P + i, &mp, sizeof(mp), hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemcpyAsync(sys->map.td + (size_t)s * P + i, &td, sizeof(td), hipMemcpyHostToDevice, sys->stream));
  st.n_points = i + 1;
  st.newq_head = st.n_points;                 // uploaded points are not "newly made" (mqNewQueue holds AddPointEpipolar's)
  r = put_state(sys, s, &st); if (r) return r;
  return i;
}

extern "C" int vslam_map_add_measurement(vslam_system* sys, int s, int keyframe, int point, int level, const double root_pos[2],
                                         int subpix, int source) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  if (!root_pos || keyframe < 0 || keyframe >= st.n_kf || point < 0 || point >= st.n_points || level < 0 || level >= NLEV) { vslam_set_error("map_add_measurement: bad argument"); return VSLAM_E_INVALID; }
  const int P = sys->p.max_points, K = sys->p.max_keyframes;
  MeasDev* slot = sys->map.kf_meas + ((size_t)s * K + keyframe) * P + point;
  MeasDev old;
  HIPCHK(hipMemcpy(&old, slot, sizeof(old), hipMemcpyDeviceToHost));
  MeasDev m; memset(&m, 0, sizeof(m));
  m.root[0] = root_pos[0]; m.root[1] = root_pos[1]; m.valid = 1; m.level = (signed char)level; m.subpix = subpix ? 1 : 0; m.source = (signed char)source;
  HIPCHK(hipMemcpy(slot, &m, sizeof(m), hipMemcpyHostToDevice));
  if (!old.valid) {   // MapMakerData::sMeasurementKFs.insert
    MapPointDev mp;
    HIPCHK(hipMemcpy(&mp, sys->map.pts + (size_t)s * P + point, sizeof(mp), hipMemcpyDeviceToHost));
    mp.n_meas_kfs++;
    HIPCHK(hipMemcpy(sys->map.pts + (size_t)s * P + point, &mp, sizeof(mp), hipMemcpyHostToDevice));
  }
  return VSLAM_OK;
}

// bulk upload used by the Python mirror for speed: arrays of n measurements
extern "C" int vslam_map_add_measurements(vslam_system* sys, int s, int n, const int* keyframe, const int* point, const int* level,
                                          const double* root_pos, const int* subpix, const int* source) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  const int P = sys->p.max_points, K = sys->p.max_keyframes;
  std::vector<MeasDev> km((size_t)K * P);
  std::vector<MapPointDev> pts(P);
  HIPCHK(hipMemcpy(km.data(), sys->map.kf_meas + (size_t)s * K * P, sizeof(MeasDev) * km.size(), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(pts.data(), sys->map.pts + (size_t)s * P, sizeof(MapPointDev) * P, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) {
    if (keyframe[i] < 0 || keyframe[i] >= st.n_kf || point[i] < 0 || point[i] >= st.n_points || level[i] < 0 || level[i] >= NLEV) { vslam_set_error("map_add_measurements: bad entry %d", i); return VSLAM_E_INVALID; }
    MeasDev& m = km[(size_t)keyframe[i] * P + point[i]];
    if (!m.valid) pts[point[i]].n_meas_kfs++;
    m.root[0] = root_pos[2 * i]; m.root[1] = root_pos[2 * i + 1]; m.valid = 1; m.level = (signed char)level[i]; m.subpix = subpix[i] ? 1 : 0; m.source = (signed char)source[i]; m.pad = 0;
  }
  HIPCHK(hipMemcpy(sys->map.kf_meas + (size_t)s * K * P, km.data(), sizeof(MeasDev) * km.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(sys->map.pts + (size_t)s * P, pts.data(), sizeof(MapPointDev) * P, hipMemcpyHostToDevice));
  return VSLAM_OK;
}

// bulk form of vslam_map_add_point: n points at once (arrays of n; pos/right/down 3n doubles, ir 2n ints)
extern "C" int vslam_map_add_points(vslam_system* sys, int s, int n, const double* pos, const int* src_keyframe, const int* src_level,
                                    const int* ir_xy, const double* right, const double* down) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  const int P = sys->p.max_points;
  if (n < 0 || st.n_points + n > P) { vslam_set_error("map point capacity %d reached", P); return VSLAM_E_CAPACITY; }
  if (n == 0) return st.n_points;
  std::vector<MapPointDev> mp(n);
  std::vector<TrackData> td(n);
  memset(mp.data(), 0, sizeof(MapPointDev) * n);
  memset(td.data(), 0, sizeof(TrackData) * n);
  for (int i = 0; i < n; i++) {
    if (src_keyframe[i] < 0 || src_keyframe[i] >= st.n_kf || src_level[i] < 0 || src_level[i] >= NLEV) { vslam_set_error("map_add_points: bad entry %d", i); return VSLAM_E_INVALID; }
    for (int q = 0; q < 3; q++) { mp[i].pos[q] = pos[3 * i + q]; mp[i].right[q] = right[3 * i + q]; mp[i].down[q] = down[3 * i + q]; }
    mp[i].src_kf = src_keyframe[i]; mp[i].src_level = src_level[i]; mp[i].irx = ir_xy[2 * i]; mp[i].iry = ir_xy[2 * i + 1];
    td[i].last_warp[0] = 9999.9; td[i].last_warp[3] = 9999.9;   // jni/PatchFinder.cc:23
  }
  HIPCHK(hipMemcpy(sys->map.pts + (size_t)s * P + st.n_points, mp.data(), sizeof(MapPointDev) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(sys->map.td + (size_t)s * P + st.n_points, td.data(), sizeof(TrackData) * n, hipMemcpyHostToDevice));
  {
    std::vector<int> lv(n, -1), fl(n, 0);
    HIPCHK(hipMemcpy(sys->map.pt_level + (size_t)s * P + st.n_points, lv.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sys->map.pt_flags + (size_t)s * P + st.n_points, fl.data(), sizeof(int) * n, hipMemcpyHostToDevice));
  }
  st.n_points += n;
  st.newq_head = st.n_points;
  r = put_state(sys, s, &st); if (r) return r;
  return st.n_points;
}

extern "C" int vslam_map_set_good(vslam_system* sys, int s) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  if (st.n_kf < 1) { vslam_set_error("map_set_good: no keyframes"); return VSLAM_E_STATE; }
  double dd[2];
  HIPCHK(hipMemcpy(dd, sys->map.kf_depth + (size_t)s * sys->p.max_keyframes * 2, sizeof(dd), hipMemcpyDeviceToHost));
  st.map_good = 1;
  st.wiggle_depth_norm = sys->p.wiggle_scale / dd[0];   // jni/MapMaker.cc:353
  st.ba_converged_recent = 1; st.ba_converged_full = 1; // MapMaker::Reset :72-73
  return put_state(sys, s, &st);
}

static void reset_tracker_fields(TrackerState& st) {   // Tracker::Reset, jni/Tracker.cc:45-62 (first call for a stream)
  if (st.frame == 0 && st.last_kf_dropped == 0 && st.depth_mean == 0.0) {
    st.boot_seed = 1u;
    st.quality = 2; st.last_kf_dropped = -20; st.depth_mean = 1.0; st.depth_sigma = 1.0; st.ba_accepted = -2; st.ba_countdown = -1;
    for (int i = 0; i < 9; i++) st.pose_final.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    st.pose_cur = st.pose_final; st.start_pose = st.pose_final;
  }
}

extern "C" int vslam_set_pose(vslam_system* sys, int s, const double pose12[12]) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  reset_tracker_fields(st);
  for (int i = 0; i < 9; i++) st.pose_final.R[i] = pose12[i];
  for (int i = 0; i < 3; i++) st.pose_final.t[i] = pose12[9 + i];
  st.pose_cur = st.pose_final; st.start_pose = st.pose_final;
  return put_state(sys, s, &st);
}

extern "C" int vslam_set_velocity(vslam_system* sys, int s, const double v6[6]) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  reset_tracker_fields(st);
  for (int i = 0; i < 6; i++) st.velocity[i] = v6[i];
  {   // mdMSDScaledVelocityMagnitude as UpdateMotionModel would leave it (jni/Tracker.cc:811-819)
    double ss = 0;
    for (int i = 0; i < 6; i++) { double v = v6[i]; if (i < 3) v *= 1.0 / st.depth_mean; ss += v * v; }
    st.msd_vel = sqrt(ss);
  }
  return put_state(sys, s, &st);
}

// Tracker::mnLastKeyFrameDropped (jni/Tracker.h:118; Reset sets -20, :59): the frame number of the stream's last keyframe.
// A benchmark spreads the keyframe phases of its independent sequences with it (they would otherwise all ask for their first
// keyframe in frame 1 and stay in step, one burst of bundle adjustments every ~21 frames).
extern "C" int vslam_set_last_keyframe_dropped(vslam_system* sys, int s, int frame) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  reset_tracker_fields(st);
  st.last_kf_dropped = frame;
  return put_state(sys, s, &st);
}

int map_init_states(vslam_system* sys) {
  std::vector<TrackerState> v(sys->S);
  memset(v.data(), 0, sizeof(TrackerState) * sys->S);
  for (auto& st : v) reset_tracker_fields(st);
  HIPCHK(hipMemcpy(sys->map.st, v.data(), sizeof(TrackerState) * sys->S, hipMemcpyHostToDevice));
  return VSLAM_OK;
}

// ---- per-frame entry points -------------------------------------------------------------------------------------
// VSLAM_PROFILE_SERIAL=1 (diagnostic): no overlap between the front-end, tracking and map-maker streams, so the
// per-stage HIP-event times are those of each kernel running alone on the device.
static bool profile_serial() { static const bool serial = getenv("VSLAM_PROFILE_SERIAL") != nullptr; return serial; }

// TrackFrame in its pieces; vslam_track_frame is exactly vslam_make_keyframe_lite, vslam_patch_search(0), vslam_pose_update(0),
// vslam_patch_search(1), vslam_pose_update(1), vslam_finish_frame.
extern "C" int vslam_patch_search(vslam_system* sys, int stage) {
  if (!sys || stage < 0 || stage > 1) { vslam_set_error("patch_search: bad argument"); return VSLAM_E_INVALID; }
  if (!sys->have_frame) { vslam_set_error("patch_search: no current frame (vslam_make_keyframe_lite first)"); return VSLAM_E_STATE; }
  if (stage == 0) {
    if (sys->frame_open) { vslam_set_error("patch_search: the previous frame was not finished (vslam_finish_frame)"); return VSLAM_E_STATE; }
    int r = ba_frame_start(sys);                                                       // deferred map-maker results that are due now
    if (r) return r;
    sys->frame_open = true;
  } else if (!sys->frame_open) { vslam_set_error("patch_search: stage 1 before stage 0"); return VSLAM_E_STATE; }
  return trk_search_stage(sys, stage);
}

extern "C" int vslam_pose_update(vslam_system* sys, int stage) {
  if (!sys || stage < 0 || stage > 1) { vslam_set_error("pose_update: bad argument"); return VSLAM_E_INVALID; }
  if (!sys->frame_open) { vslam_set_error("pose_update: no search stage has run on this frame"); return VSLAM_E_STATE; }
  return trk_pose_stage(sys, stage);
}

extern "C" int vslam_finish_frame(vslam_system* sys) {
  if (!sys) return VSLAM_E_INVALID;
  if (!sys->frame_open) { vslam_set_error("finish_frame: no frame in progress"); return VSLAM_E_STATE; }
  sys->frame_open = false;
  int r = ba_add_keyframe_and_adjust(sys);                                             // :128-132 -> MapMaker::AddKeyFrame
  if (!r && sys->p.idle_iterations > 0) r = mm_idle(sys);                              // the map-maker's idle jobs
  if (!r && sys->p.bootstrap) r = boot_frame(sys);                                     // jni/Tracker.cc:144-145: TrackForInitialMap for the streams without a map
  prof_mark(sys, VSLAM_N_STAGES);
  if (sys->prof_on && sys->prof_frame < sys->prof_cap) sys->prof_frame++;
  if (!r) HIPCHK(hipEventRecord(sys->ev_track_done[sys->fr_idx], sys->stream));       // the front-end may now reuse this buffer
  sys->frame_no++;
  if (profile_serial() && !r) { HIPCHK(hipStreamSynchronize(sys->stream)); int rs = ba_sync_streams(sys); if (rs) return rs; }
  return r;
}

extern "C" int vslam_track_frame(vslam_system* sys, const uint8_t* gray, size_t row_stride, size_t stream_stride, int on_device) {
  if (!sys) return VSLAM_E_INVALID;
  int r = fe_make_keyframe_lite(sys, gray, row_stride, stream_stride, on_device);   // jni/Tracker.cc:85
  if (r) return r;
  if (profile_serial()) HIPCHK(hipStreamSynchronize(sys->fe_stream));
  r = vslam_patch_search(sys, 0);                                                    // :103-124 TrackMap
  if (!r) r = vslam_pose_update(sys, 0);
  if (!r) r = vslam_patch_search(sys, 1);
  if (!r) r = vslam_pose_update(sys, 1);
  if (r) { sys->frame_open = false; return r; }
  return vslam_finish_frame(sys);
}

// Map editing after the upload (a host-side map-maker, map loading, tests that re-synchronise the map to a reference):
// MapPoint::v3WorldPos of points [first, first + n) and KeyFrame::se3CfromW of one keyframe.
extern "C" int vslam_map_set_point_positions(vslam_system* sys, int s, int first, int n, const double* pos3) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  if (!pos3 || first < 0 || n < 0 || first + n > st.n_points) { vslam_set_error("map_set_point_positions: bad range"); return VSLAM_E_INVALID; }
  if (n == 0) return VSLAM_OK;
  const int P = sys->p.max_points;
  std::vector<MapPointDev> mp(n);
  HIPCHK(hipMemcpy(mp.data(), sys->map.pts + (size_t)s * P + first, sizeof(MapPointDev) * n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) for (int q = 0; q < 3; q++) mp[i].pos[q] = pos3[3 * i + q];
  HIPCHK(hipMemcpy(sys->map.pts + (size_t)s * P + first, mp.data(), sizeof(MapPointDev) * n, hipMemcpyHostToDevice));
  return VSLAM_OK;
}

extern "C" int vslam_map_set_keyframe_pose(vslam_system* sys, int s, int keyframe, const double pose12[12]) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  if (!pose12 || keyframe < 0 || keyframe >= st.n_kf) { vslam_set_error("map_set_keyframe_pose: bad argument"); return VSLAM_E_INVALID; }
  Pose p;
  for (int i = 0; i < 9; i++) p.R[i] = pose12[i];
  for (int i = 0; i < 3; i++) p.t[i] = pose12[9 + i];
  HIPCHK(hipMemcpy(sys->map.kf_pose + (size_t)s * sys->p.max_keyframes + keyframe, &p, sizeof(Pose), hipMemcpyHostToDevice));
  return VSLAM_OK;
}

static const char* kStageNames[VSLAM_N_STAGES] = {"pyr_fast0", "fast_lvl", "compact", "pvs", "plan_coarse", "search_coarse", "pose_coarse",
                                                  "plan_fine", "search_fine", "pose_fine", "add_keyframe", "ba_assemble", "ba_compute", "ba_writeback"};
extern "C" const char* vslam_stage_name(int stage) { return stage >= 0 && stage < VSLAM_N_STAGES ? kStageNames[stage] : ""; }

extern "C" int vslam_profile_begin(vslam_system* sys, int max_frames) {
  if (!sys || max_frames < 1) return VSLAM_E_INVALID;
  HIPCHK(hipStreamSynchronize(sys->stream));
  while ((int)sys->prof_ev.size() < max_frames * PROF_MARKS) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); sys->prof_ev.push_back(e); }
  sys->prof_cap = max_frames; sys->prof_frame = 0; sys->prof_on = true;
  sys->prof_ba_launched.assign((size_t)max_frames, 0);
  return VSLAM_OK;
}

extern "C" int vslam_profile_end(vslam_system* sys, double* stage_ms, int* n_frames) {
  if (!sys || !stage_ms) return VSLAM_E_INVALID;
  HIPCHK(hipStreamSynchronize(sys->fe_stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  { int rs = ba_sync_streams(sys); if (rs) return rs; }
  sys->prof_on = false;
  for (int k = 0; k < VSLAM_N_STAGES; k++) stage_ms[k] = 0.0;
  for (int f = 0; f < sys->prof_frame; f++)
    for (int k = 0; k < VSLAM_N_STAGES; k++) {
      float ms = 0.f;
      int end = k == 2 ? PROF_FE_END : k + 1;   // stages 0..2 run on the front-end stream
      if (sys->tp.ba_delay > 0) { if (k == 11) end = VSLAM_N_STAGES; else if (k == 12) end = PROF_BA_END; else if (k == 13) end = 3; }
      if (sys->tp.ba_delay > 0 && k == 12 && !(f < (int)sys->prof_ba_launched.size() && sys->prof_ba_launched[f])) continue;   // no launch in this frame
      HIPCHK(hipEventElapsedTime(&ms, sys->prof_ev[(size_t)f * PROF_MARKS + k], sys->prof_ev[(size_t)f * PROF_MARKS + end]));
      stage_ms[k] += ms;
    }
  if (n_frames) *n_frames = sys->prof_frame;
  return VSLAM_OK;
}

extern "C" int vslam_profile_launches(vslam_system* sys, int* launches) {
  if (!sys || !launches) return VSLAM_E_INVALID;
  for (int k = 0; k < VSLAM_N_STAGES; k++) launches[k] = sys->prof_frame;
  if (sys->tp.ba_delay > 0) {
    int n = 0;
    for (int f = 0; f < sys->prof_frame && f < (int)sys->prof_ba_launched.size(); f++) n += sys->prof_ba_launched[f] ? 1 : 0;
    launches[12] = n;
  }
  return VSLAM_OK;
}

extern "C" int vslam_update(vslam_system* sys, const uint8_t* gray, size_t row_stride, size_t stream_stride) {
  int r = vslam_track_frame(sys, gray, row_stride, stream_stride, 0);
  if (r) return r;
  HIPCHK(hipStreamSynchronize(sys->stream));   // the caller may reuse its buffer (TrackFrame copies its input)
  return VSLAM_OK;
}

// native_touchScreen (jni/jni_part.cpp:120-124): mbUserPressedSpacebar = true.  Only the map bootstrap reads the key.
extern "C" int vslam_touch(vslam_system* sys) {
  if (!sys) return VSLAM_E_INVALID;
  return sys->p.bootstrap ? vslam_press_spacebar(sys, -1) : VSLAM_OK;
}

extern "C" int vslam_bundle_adjust_recent(vslam_system* sys) { if (!sys) return VSLAM_E_INVALID; return ba_run(sys, 1); }
extern "C" int vslam_bundle_adjust_all(vslam_system* sys) { if (!sys) return VSLAM_E_INVALID; return ba_run(sys, 2); }

// ---- read-back ------------------------------------------------------------------------------------------------------
static void export_state(const vslam_system* sys, const TrackerState& st, vslam_track_state* o);

extern "C" int vslam_get_state(vslam_system* sys, int s, vslam_track_state* o) {
  CHK_STREAM(sys, s);
  if (!o) return VSLAM_E_INVALID;
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  export_state(sys, st, o);
  return VSLAM_OK;
}

// vslam_get_state for the streams [first, first + n) with one copy (a caller that polls thousands of streams one by one idles the GPU
// for tens of milliseconds, long enough for its clocks to drop)
extern "C" int vslam_get_states(vslam_system* sys, int first, int n, vslam_track_state* out) {
  if (!sys || !out || first < 0 || n < 0 || first + n > sys->S) { vslam_set_error("get_states: bad argument"); return VSLAM_E_INVALID; }
  if (n == 0) return VSLAM_OK;
  std::vector<TrackerState> v((size_t)n);
  HIPCHK(hipMemcpyAsync(v.data(), sys->map.st + first, sizeof(TrackerState) * n, hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  for (int i = 0; i < n; i++) export_state(sys, v[(size_t)i], out + i);
  return VSLAM_OK;
}

static void export_state(const vslam_system* sys, const TrackerState& st, vslam_track_state* o) {
  const Pose& T = sys->frame_open ? st.pose_cur : st.pose_final;   // between the stages of a frame: the tracker's current estimate
  for (int i = 0; i < 9; i++) o->pose[i] = T.R[i];
  for (int i = 0; i < 3; i++) o->pose[9 + i] = T.t[i];
  for (int i = 0; i < 6; i++) o->velocity[i] = st.velocity[i];
  o->msd_velocity = st.msd_vel; o->depth_mean = st.depth_mean; o->depth_sigma = st.depth_sigma;
  for (int i = 0; i < NLEV; i++) { o->attempted[i] = st.attempted[i]; o->found[i] = st.found[i]; }
  o->quality = st.quality; o->lost_frames = st.lost_frames; o->frame = st.frame; o->did_coarse = st.did_coarse;
  o->kf_added = st.kf_added; o->n_keyframes = st.n_kf; o->n_points = st.n_points; o->ba_accepted = st.ba_accepted;
  o->n_zmssd = (long long)st.n_zmssd; o->n_ba_trials = (long long)st.n_ba_trials;
}

// counters of the map-maker's idle jobs (vslam_params.idle_iterations): points re-found by ReFindNewlyMade and by
// ReFindFromFailureQueue, BundleAdjustAll and idle BundleAdjustRecent calls, failure-queue and new-queue lengths
extern "C" int vslam_get_idle_stats(vslam_system* sys, int s, int out[6]) {
  CHK_STREAM(sys, s);
  if (!out) return VSLAM_E_INVALID;
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  out[0] = st.n_refound_new; out[1] = st.n_refound_failed; out[2] = st.n_ba_all; out[3] = st.n_ba_recent_idle; out[4] = st.fq_n; out[5] = st.n_points - st.newq_head;
  return VSLAM_OK;
}

// One idle job of MapMaker::run for every stream, outside a frame (the streams decide on device whether it is due)
extern "C" int vslam_mapmaker_idle_job(vslam_system* sys, int job) {
  if (!sys) { vslam_set_error("mapmaker_idle_job: null system"); return VSLAM_E_INVALID; }
  if (sys->frame_open) { vslam_set_error("mapmaker_idle_job: a frame is open (vslam_finish_frame first)"); return VSLAM_E_STATE; }
  if (sys->p.idle_iterations == 0) { vslam_set_error("mapmaker_idle_job: created with idle_iterations = 0 (no failure queue / never-retry sets are kept); use -1 for idle jobs on request only"); return VSLAM_E_STATE; }
  return mm_idle_job(sys, job);
}

// The dump of vslam_save_map read back (MapMaker's "SaveMap", jni/MapMaker.cc:1254-1286): <dir>/map.dump holds v3WorldPos (Eigen's
// column print, one coordinate per line) + two blanks + nSourceLevel per good point, <dir>/keyframes/<i>.info the rows of [R | t].
static int read_numbers(const char* path, std::vector<double>& out) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  char line[1024];
  while (fgets(line, sizeof(line), f)) {
    char* q = line;
    for (;;) { char* e = nullptr; const double v = strtod(q, &e); if (e == q) break; q = e; out.push_back(v); }
  }
  fclose(f);
  return 0;
}
extern "C" int vslam_read_map_dump(const char* dir, double* pos3, int* level, int point_cap, int* n_points, double* pose12, int kf_cap, int* n_keyframes) {
  if (!dir) { vslam_set_error("read_map_dump: null directory"); return VSLAM_E_INVALID; }
  char path[4096];
  snprintf(path, sizeof(path), "%s/map.dump", dir);
  std::vector<double> v;
  if (read_numbers(path, v)) { vslam_set_error("read_map_dump: cannot open map.dump in the given directory"); return VSLAM_E_INVALID; }
  if (v.size() % 4) { vslam_set_error("read_map_dump: map.dump does not hold whole points (x, y, z, level)"); return VSLAM_E_INVALID; }
  const int np = (int)(v.size() / 4);
  for (int i = 0; i < np && i < point_cap; i++) {
    if (pos3) for (int k = 0; k < 3; k++) pos3[3 * i + k] = v[4 * (size_t)i + k];
    if (level) level[i] = (int)v[4 * (size_t)i + 3];
  }
  int nk = 0;
  for (;; nk++) {
    snprintf(path, sizeof(path), "%s/keyframes/%d.info", dir, nk);
    std::vector<double> q;
    if (read_numbers(path, q)) break;
    if (q.size() != 12) { vslam_set_error("read_map_dump: keyframes/%d.info does not hold a 3x4 pose", nk); return VSLAM_E_INVALID; }
    if (nk < kf_cap && pose12) {
      double* o = pose12 + 12 * (size_t)nk;
      for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) o[r * 3 + c] = q[r * 4 + c]; o[9 + r] = q[r * 4 + 3]; }
    }
  }
  if (n_points) *n_points = np;
  if (n_keyframes) *n_keyframes = nk;
  return VSLAM_OK;
}

// The state a dump holds -- positions of the good points, keyframe poses -- written back into the map it was saved from (same
// keyframes, same points; the images, templates and measurements are not part of the reference's format): a checkpoint of what the
// bundle adjustment has estimated.
extern "C" int vslam_load_map(vslam_system* sys, int s, const char* dir) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  const int n = st.n_points, nk = st.n_kf;
  std::vector<double> pos(3 * (size_t)(n > 0 ? n : 1)), poses(12 * (size_t)(nk > 0 ? nk : 1));
  std::vector<int> lev((size_t)(n > 0 ? n : 1));
  int np = 0, nkf = 0;
  r = vslam_read_map_dump(dir, pos.data(), lev.data(), n, &np, poses.data(), nk, &nkf); if (r) return r;
  std::vector<MapPointDev> p((size_t)(n > 0 ? n : 1));
  if (n > 0) HIPCHK(hipMemcpy(p.data(), sys->map.pts + (size_t)s * sys->p.max_points, sizeof(MapPointDev) * n, hipMemcpyDeviceToHost));
  int good = 0;
  for (int i = 0; i < n; i++) if (!p[i].bad) good++;
  if (np != good || nkf != nk) { vslam_set_error("load_map: the dump holds %d points and %d keyframes, the map %d good points and %d keyframes", np, nkf, good, nk); return VSLAM_E_STATE; }
  int k = 0;
  for (int i = 0; i < n; i++) {
    if (p[i].bad) continue;
    if (lev[k] != p[i].src_level) { vslam_set_error("load_map: point %d of the dump has source level %d, the map's %d", k, lev[k], p[i].src_level); return VSLAM_E_STATE; }
    for (int q = 0; q < 3; q++) p[i].pos[q] = pos[3 * (size_t)k + q];
    k++;
  }
  if (n > 0) HIPCHK(hipMemcpy(sys->map.pts + (size_t)s * sys->p.max_points, p.data(), sizeof(MapPointDev) * n, hipMemcpyHostToDevice));
  for (int kf = 0; kf < nk; kf++) { r = vslam_map_set_keyframe_pose(sys, s, kf, poses.data() + 12 * (size_t)kf); if (r) return r; }
  return VSLAM_OK;
}

extern "C" int vslam_get_message(vslam_system* sys, int s, char* buf, size_t cap) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  if (!buf || !cap) return VSLAM_E_INVALID;
  if (!st.map_good && st.init_stage == 1) snprintf(buf, cap, "Translate the camera slowly sideways, and press spacebar again to perform stereo init.");   // jni/Tracker.cc:284
  else if (!st.map_good && st.init_stage == 2) buf[0] = 0;                                                                       // TRAIL_TRACKING_COMPLETE without a map: TrackForInitialMap says nothing
  else if (!st.map_good) snprintf(buf, cap, "Point camera at planar scene and press spacebar to start tracking for initial map.");   // :260
  else if (st.lost_frames >= 3) snprintf(buf, cap, "** Attempting recovery **.");                                                // :134
  else snprintf(buf, cap, "Tracking Map, quality %s Found: %d/%d %d/%d %d/%d %d/%d Map: %dP, %dKF%s",                            // :110-124
                st.quality == 2 ? "good." : st.quality == 1 ? "poor." : "bad.", st.found[0], st.attempted[0], st.found[1], st.attempted[1],
                st.found[2], st.attempted[2], st.found[3], st.attempted[3], st.n_points, st.n_kf, st.kf_added ? " Adding key-frame." : "");
  return VSLAM_OK;
}

extern "C" int vslam_get_point_tracks(vslam_system* sys, int s, int* found, int* searched, int* level, int* subpix, double* vfound,
                                      double* image, int cap) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  const int n = st.n_points < cap ? st.n_points : cap;
  std::vector<TrackData> td(n > 0 ? n : 1);
  std::vector<int> lv(n > 0 ? n : 1), fl(n > 0 ? n : 1);
  if (n > 0) {
    HIPCHK(hipMemcpy(td.data(), sys->map.td + (size_t)s * sys->p.max_points, sizeof(TrackData) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lv.data(), sys->map.pt_level + (size_t)s * sys->p.max_points, sizeof(int) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fl.data(), sys->map.pt_flags + (size_t)s * sys->p.max_points, sizeof(int) * n, hipMemcpyDeviceToHost));
  }
  for (int i = 0; i < n; i++) {
    if (found) found[i] = (fl[i] & TDF_FOUND) ? 1 : 0;
    if (searched) searched[i] = (fl[i] & TDF_SEARCHED) ? 1 : 0;
    if (level) level[i] = lv[i];
    if (subpix) subpix[i] = (fl[i] & TDF_SUBPIX) ? 1 : 0;
    if (vfound) { vfound[2 * i] = td[i].vfound[0]; vfound[2 * i + 1] = td[i].vfound[1]; }
    if (image) { image[2 * i] = td[i].image[0]; image[2 * i + 1] = td[i].image[1]; }
  }
  return st.n_points;
}

extern "C" int vslam_get_points(vslam_system* sys, int s, double* pos3, int* bad, int* n_in, int* n_out, int cap) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  const int n = st.n_points < cap ? st.n_points : cap;
  std::vector<MapPointDev> p(n > 0 ? n : 1);
  if (n > 0) HIPCHK(hipMemcpy(p.data(), sys->map.pts + (size_t)s * sys->p.max_points, sizeof(MapPointDev) * n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) {
    if (pos3) for (int q = 0; q < 3; q++) pos3[3 * i + q] = p[i].pos[q];
    if (bad) bad[i] = p[i].bad;
    if (n_in) n_in[i] = p[i].n_in;
    if (n_out) n_out[i] = p[i].n_out;
  }
  return st.n_points;
}

extern "C" int vslam_get_keyframe_pose(vslam_system* sys, int s, int k, double pose12[12]) {
  CHK_STREAM(sys, s);
  if (k < 0 || k >= sys->p.max_keyframes || !pose12) return VSLAM_E_INVALID;
  Pose p;
  HIPCHK(hipStreamSynchronize(sys->stream));
  HIPCHK(hipMemcpy(&p, sys->map.kf_pose + (size_t)s * sys->p.max_keyframes + k, sizeof(p), hipMemcpyDeviceToHost));
  for (int i = 0; i < 9; i++) pose12[i] = p.R[i];
  for (int i = 0; i < 3; i++) pose12[9 + i] = p.t[i];
  return VSLAM_OK;
}

// MapMaker::NeedNewKeyFrame (jni/MapMaker.cc:761-773) and IsDistanceToNearestKeyFrameExcessive (:1098-1101) for the stream's current
// pose, on the host from the keyframe poses (the tracker takes both decisions on device, k_pose; these are the public members)
static int nearest_keyframe_dist(vslam_system* sys, int s, TrackerState* st, double* dist) {
  int r = get_state(sys, s, st); if (r) return r;
  if (st->n_kf < 1) { vslam_set_error("no keyframes"); return VSLAM_E_STATE; }
  std::vector<Pose> kp(st->n_kf);
  HIPCHK(hipMemcpy(kp.data(), sys->map.kf_pose + (size_t)s * sys->p.max_keyframes, sizeof(Pose) * st->n_kf, hipMemcpyDeviceToHost));
  const Pose ic = pose_inverse(st->pose_final);
  double best = 9999999999.9;                                      // ClosestKeyFrame :737-758 with KeyFrameLinearDist :705-712
  for (int k = 0; k < st->n_kf; k++) {
    const Pose ik = pose_inverse(kp[k]);
    const double d0 = ik.t[0] - ic.t[0], d1 = ik.t[1] - ic.t[1], d2 = ik.t[2] - ic.t[2];
    const double d = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    if (d < best) best = d;
  }
  *dist = best;
  return VSLAM_OK;
}

extern "C" int vslam_need_new_keyframe(vslam_system* sys, int s, int* need) {
  CHK_STREAM(sys, s);
  if (!need) return VSLAM_E_INVALID;
  TrackerState st; double d;
  int r = nearest_keyframe_dist(sys, s, &st, &d); if (r) return r;
  d *= (1.0 / st.depth_mean);
  *need = d > sys->p.max_kf_dist_wiggle_mult * st.wiggle_depth_norm ? 1 : 0;
  return VSLAM_OK;
}

extern "C" int vslam_distance_to_nearest_keyframe_excessive(vslam_system* sys, int s, int* excessive) {
  CHK_STREAM(sys, s);
  if (!excessive) return VSLAM_E_INVALID;
  TrackerState st; double d;
  int r = nearest_keyframe_dist(sys, s, &st, &d); if (r) return r;
  *excessive = d > sys->p.wiggle_scale * 10.0 ? 1 : 0;
  return VSLAM_OK;
}

// MapMaker::GUICommandHandler("SaveMap"), jni/MapMaker.cc:1254-1286.  std::ostream's default floating-point format is
// %g with precision 6; Eigen's default IOFormat right-aligns the coefficients of a matrix to the widest one.
extern "C" int vslam_save_map(vslam_system* sys, int s, const char* dir) {
  CHK_STREAM(sys, s);
  if (!dir) return VSLAM_E_INVALID;
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  const int n = st.n_points, nk = st.n_kf;
  std::vector<MapPointDev> p(n > 0 ? n : 1);
  if (n > 0) HIPCHK(hipMemcpy(p.data(), sys->map.pts + (size_t)s * sys->p.max_points, sizeof(MapPointDev) * n, hipMemcpyDeviceToHost));
  std::vector<Pose> kp(nk > 0 ? nk : 1);
  if (nk > 0) HIPCHK(hipMemcpy(kp.data(), sys->map.kf_pose + (size_t)s * sys->p.max_keyframes, sizeof(Pose) * nk, hipMemcpyDeviceToHost));
  char path[4096];
  snprintf(path, sizeof(path), "%s/map.dump", dir);
  FILE* f = fopen(path, "w");
  if (!f) { vslam_set_error("save_map: cannot open map.dump in the given directory"); return VSLAM_E_INVALID; }
  int written = 0;
  for (int i = 0; i < n; i++) {
    if (p[i].bad) continue;
    char c[3][64]; size_t wmax = 0;
    for (int q = 0; q < 3; q++) { snprintf(c[q], sizeof(c[q]), "%g", p[i].pos[q]); if (strlen(c[q]) > wmax) wmax = strlen(c[q]); }
    fprintf(f, "%*s\n%*s\n%*s  %d\n", (int)wmax, c[0], (int)wmax, c[1], (int)wmax, c[2], p[i].src_level);
    written++;
  }
  fclose(f);
  for (int k = 0; k < nk; k++) {
    snprintf(path, sizeof(path), "%s/keyframes/%d.info", dir, k);
    FILE* g = fopen(path, "w");
    if (!g) { vslam_set_error("save_map: cannot open keyframes/<i>.info in the given directory"); return VSLAM_E_INVALID; }
    for (int i = 0; i < 3; i++) fprintf(g, "%g %g %g %g\n", kp[k].R[i * 3], kp[k].R[i * 3 + 1], kp[k].R[i * 3 + 2], kp[k].t[i]);
    fprintf(g, "\n");
    fclose(g);
  }
  return written;
}

extern "C" int vslam_get_keyframe_measurements(vslam_system* sys, int s, int k, int* point, int* level, double* root_pos, int* source, int cap) {
  CHK_STREAM(sys, s);
  TrackerState st;
  int r = get_state(sys, s, &st); if (r) return r;
  if (k < 0 || k >= st.n_kf) return VSLAM_E_INVALID;
  const int P = sys->p.max_points;
  std::vector<MeasDev> km(P);
  HIPCHK(hipMemcpy(km.data(), sys->map.kf_meas + ((size_t)s * sys->p.max_keyframes + k) * P, sizeof(MeasDev) * P, hipMemcpyDeviceToHost));
  int n = 0;
  for (int i = 0; i < st.n_points; i++) {
    if (!km[i].valid) continue;
    if (n < cap) {
      if (point) point[n] = i;
      if (level) level[n] = km[i].level;
      if (root_pos) { root_pos[2 * n] = km[i].root[0]; root_pos[2 * n + 1] = km[i].root[1]; }
      if (source) source[n] = km[i].source;
    }
    n++;
  }
  return n;
}

extern "C" int vslam_get_template(vslam_system* sys, int s, int point, uint8_t* tmpl, int* sum, int* sumsq, int* bad) {
  CHK_STREAM(sys, s);
  const int P = sys->p.max_points, PS = sys->p.patch_size;
  if (point < 0 || point >= P) return VSLAM_E_INVALID;
  TrackData td;
  HIPCHK(hipStreamSynchronize(sys->stream));
  HIPCHK(hipMemcpy(&td, sys->map.td + (size_t)s * P + point, sizeof(td), hipMemcpyDeviceToHost));
  if (tmpl) HIPCHK(hipMemcpy(tmpl, sys->map.tmpl + ((size_t)s * P + point) * TMPL_PITCH, PS * PS, hipMemcpyDeviceToHost));
  if (sum) *sum = td.tsum;
  if (sumsq) *sumsq = td.tsumsq;
  int fl = 0;
  HIPCHK(hipMemcpy(&fl, sys->map.pt_flags + (size_t)s * P + point, sizeof(int), hipMemcpyDeviceToHost));
  if (bad) *bad = (fl & TDF_TMPL_BAD) ? 1 : 0;
  return (fl & TDF_HAVE_LAST) ? 1 : 0;
}

// the same for the points [first, first + n) at once: tmpl n * P * P bytes, the other arrays n ints (any may be NULL)
extern "C" int vslam_get_templates(vslam_system* sys, int s, int first, int n, uint8_t* tmpl, int* sum, int* sumsq, int* bad, int* have) {
  CHK_STREAM(sys, s);
  const int P = sys->p.max_points, PS = sys->p.patch_size;
  if (first < 0 || n < 0 || first + n > P) return VSLAM_E_INVALID;
  if (n == 0) return VSLAM_OK;
  HIPCHK(hipStreamSynchronize(sys->stream));
  std::vector<TrackData> td(n);
  std::vector<uint8_t> raw((size_t)n * TMPL_PITCH);
  std::vector<int> fl(n);
  HIPCHK(hipMemcpy(td.data(), sys->map.td + (size_t)s * P + first, sizeof(TrackData) * n, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(raw.data(), sys->map.tmpl + ((size_t)s * P + first) * TMPL_PITCH, raw.size(), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(fl.data(), sys->map.pt_flags + (size_t)s * P + first, sizeof(int) * n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) {
    if (tmpl) memcpy(tmpl + (size_t)i * PS * PS, raw.data() + (size_t)i * TMPL_PITCH, (size_t)PS * PS);
    if (sum) sum[i] = td[i].tsum;
    if (sumsq) sumsq[i] = td[i].tsumsq;
    if (bad) bad[i] = (fl[i] & TDF_TMPL_BAD) ? 1 : 0;
    if (have) have[i] = (fl[i] & TDF_HAVE_LAST) ? 1 : 0;
  }
  return VSLAM_OK;
}
