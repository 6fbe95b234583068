// ORACLE (test infrastructure only).  extern "C" surface of the whole-path oracle for ctypes.
#include "ptam_system.hpp"

using namespace orc;

static SE3 pose_from12(const double p[12]) { SE3 T; for (int i = 0; i < 9; i++) T.R[i] = p[i]; for (int i = 0; i < 3; i++) T.t[i] = p[9 + i]; return T; }
static void pose_to12(const SE3& T, double p[12]) { for (int i = 0; i < 9; i++) p[i] = T.R[i]; for (int i = 0; i < 3; i++) p[9 + i] = T.t[i]; }

extern "C" {

void* orc_sys_create(const orc_params* q) {
  Params p;
  p.width = q->width; p.height = q->height; p.patch_size = q->patch_size;
  for (int i = 0; i < 4; i++) p.thr[i] = q->thr[i];
  p.nonmax_barrier = q->nonmax_barrier; p.max_patches = q->max_patches;
  p.coarse_min = q->coarse_min; p.coarse_max = q->coarse_max; p.coarse_range = q->coarse_range;
  p.coarse_subpix_its = q->coarse_subpix_its; p.coarse_disabled = q->coarse_disabled; p.coarse_min_vel = q->coarse_min_vel;
  p.fine_subpix_its = q->fine_subpix_its; p.wls_prior = q->wls_prior; p.min_frames_between_kf = q->min_frames_between_kf;
  p.max_kf_dist_wiggle_mult = q->max_kf_dist_wiggle_mult; p.wiggle_scale = q->wiggle_scale;
  p.ba_max_iterations = q->ba_max_iterations; p.ba_convergence_limit = q->ba_convergence_limit;
  p.ba_min_tukey_sigma = q->ba_min_tukey_sigma; p.ba_window = q->ba_window; p.ba_min_keyframes = q->ba_min_keyframes;
  for (int i = 0; i < 5; i++) p.cam[i] = q->cam[i];
  p.quirks = q->quirks;
  p.ba_delay_frames = q->ba_delay_frames;
  p.use_sbi = q->use_sbi;
  p.grow_map = q->grow_map;
  p.idle_iterations = q->idle_iterations;
  return new System(p);
}
void orc_sys_destroy(void* s) { delete (System*)s; }
int orc_sys_add_keyframe(void* s, const double pose12[12], int fixed, const uint8_t* gray, int stride, double dmean, double dsigma) {
  return ((System*)s)->AddKeyFrameRaw(pose12, fixed != 0, gray, stride, dmean, dsigma);
}
int orc_sys_add_point(void* s, const double pos[3], int src_kf, int src_level, int irx, int iry, const double right[3], const double down[3]) {
  return ((System*)s)->AddPointRaw(pos, src_kf, src_level, irx, iry, right, down);
}
void orc_sys_add_meas(void* s, int kf, int pt, int level, const double root[2], int subpix, int source) {
  ((System*)s)->AddMeasRaw(kf, pt, level, root, subpix != 0, source);
}
void orc_sys_set_map_good(void* s) { ((System*)s)->SetMapGood(); }
void orc_sys_set_pose(void* s, const double pose12[12]) { ((System*)s)->pose = pose_from12(pose12); }
void orc_sys_set_velocity(void* sv, const double v6[6]) {
  System* s = (System*)sv;
  double ss = 0;   // mdMSDScaledVelocityMagnitude as UpdateMotionModel would leave it (jni/Tracker.cc:811-819)
  for (int i = 0; i < 6; i++) { s->velocity[i] = v6[i]; double v = v6[i]; if (i < 3) v *= 1.0 / s->cur.depth_mean; ss += v * v; }
  s->msd_vel = sqrt(ss);
}
void orc_sys_track_frame(void* s, const uint8_t* gray, int stride) { ((System*)s)->TrackFrame(gray, stride); }
// TrackFrame in stages: begin; search 0; pose 0; search 1; pose 1; end (the stages do nothing while there is no map / the tracker is lost)
void orc_sys_frame_begin(void* s, const uint8_t* gray, int stride) { ((System*)s)->FrameBegin(gray, stride); }
void orc_sys_search_stage(void* s, int stage) { System* S = (System*)s; if (S->tracked_this_frame) S->SearchStage(stage); }
void orc_sys_pose_stage(void* s, int stage) { System* S = (System*)s; if (S->tracked_this_frame) S->PoseStage(stage); }
void orc_sys_frame_end(void* s) { ((System*)s)->FrameEnd(); }
void orc_sys_idle_iteration(void* s) { ((System*)s)->IdleIteration(); }
void orc_sys_idle_job(void* s, int job) { ((System*)s)->IdleJob(job); }
void orc_sys_set_last_keyframe_dropped(void* s, int frame) { ((System*)s)->last_kf_dropped = frame; }   // Tracker::mnLastKeyFrameDropped
void orc_sys_press_spacebar(void* s) { ((System*)s)->spacebar = true; }
void orc_sys_set_boot_seed(void* s, unsigned seed) { ((System*)s)->boot_seed = seed; }
// MapMaker::InitFromStereo(kFirst, kSecond, vMatches, se3) with the caller's frames and matches (jni/MapMaker.cc:204-376): the keyframes are
// made from the two images the way the tracker makes its current frame (MakeKeyFrame_Lite); the second is the tracker's current frame
int orc_sys_init_from_stereo(void* sv, const uint8_t* first, const uint8_t* second, int stride, const int* matches4, int n, double pose12[12]) {
  System* s = (System*)sv;
  if (s->map_good || s->init_stage != 0) return -1;
  KeyFrame kF, kS;
  make_keyframe_lite(kF, first, s->p.width, s->p.height, stride, s->p.thr);
  make_keyframe_lite(kS, second, s->p.width, s->p.height, stride, s->p.thr);
  std::vector<std::array<int, 4>> m((size_t)n);
  for (int i = 0; i < n; i++) m[(size_t)i] = {matches4[4 * i], matches4[4 * i + 1], matches4[4 * i + 2], matches4[4 * i + 3]};
  s->cur = kS;
  s->frame += 2;                                                     // the two frames the device path spends on it
  s->n_hom_inliers = 0; s->n_init_points = 0;
  s->init_ok = s->InitFromStereo(kF, kS, m);
  s->init_stage = 2;
  if (pose12) pose_to12(s->pose, pose12);
  return s->init_ok ? 1 : 0;
}
void orc_sys_get_init_info(void* sv, int out[6]) {
  System* s = (System*)sv;
  out[0] = s->init_stage; out[1] = (int)s->trails.size(); out[2] = s->init_ok ? 1 : 0; out[3] = s->n_hom_inliers; out[4] = s->n_init_points; out[5] = s->map_good ? 1 : 0;
}
int orc_sys_get_trails(void* sv, int* out4, int cap) {
  System* s = (System*)sv;
  const int n = (int)s->trails.size();
  for (int i = 0; i < n && i < cap; i++) { out4[4 * i] = s->trails[i].init[0]; out4[4 * i + 1] = s->trails[i].init[1]; out4[4 * i + 2] = s->trails[i].cur[0]; out4[4 * i + 3] = s->trails[i].cur[1]; }
  return n;
}
void orc_sys_get_idle_stats(void* sv, int out[6]) {
  System* s = (System*)sv;
  out[0] = s->n_refound_new; out[1] = s->n_refound_failed; out[2] = s->n_ba_all; out[3] = s->n_ba_recent_idle; out[4] = (int)s->failure_queue.size(); out[5] = (int)s->new_queue.size();
}

void orc_sys_get_state(void* sv, orc_track_state* o) {
  System* s = (System*)sv;
  pose_to12(s->pose, o->pose);
  for (int i = 0; i < 6; i++) o->velocity[i] = s->velocity[i];
  o->msd_velocity = s->msd_vel; o->depth_mean = s->cur.depth_mean; o->depth_sigma = s->cur.depth_sigma;
  for (int i = 0; i < 4; i++) { o->attempted[i] = s->attempted[i]; o->found[i] = s->found[i]; }
  o->quality = s->quality; o->lost_frames = s->lost_frames; o->frame = s->frame; o->did_coarse = s->did_coarse;
  o->kf_added = s->kf_added_this_frame; o->n_keyframes = (int)s->kfs.size(); o->n_points = (int)s->pts.size();
  o->ba_accepted = s->last_ba_accepted; o->n_zmssd = s->n_zmssd; o->n_ba_trials = s->n_ba_trials;
}

int orc_sys_get_point_tracks(void* sv, int* found, int* searched, int* level, int* subpix, double* vfound, double* image, int cap) {
  System* s = (System*)sv;
  const int n = (int)s->pts.size();
  for (int i = 0; i < n && i < cap; i++) {
    const MapPoint& p = *s->pts[i];
    found[i] = p.found; searched[i] = p.searched; level[i] = p.search_level; subpix[i] = p.did_subpix;
    vfound[2 * i] = p.vfound[0]; vfound[2 * i + 1] = p.vfound[1];
    image[2 * i] = p.image[0]; image[2 * i + 1] = p.image[1];
  }
  return n;
}

int orc_sys_get_points(void* sv, double* pos3, int* bad, int* n_in, int* n_out, int cap) {
  System* s = (System*)sv;
  const int n = (int)s->pts.size();
  for (int i = 0; i < n && i < cap; i++) {
    const MapPoint& p = *s->pts[i];
    for (int k = 0; k < 3; k++) pos3[3 * i + k] = p.pos[k];
    bad[i] = p.bad; n_in[i] = p.n_inlier; n_out[i] = p.n_outlier;
  }
  return n;
}

void orc_sys_get_keyframe_pose(void* sv, int kf, double pose12[12]) { pose_to12(((System*)sv)->kfs[kf]->pose, pose12); }

int orc_sys_get_keyframe_meas(void* sv, int kf, int* pt, int* level, double* root, int* source, int cap) {
  System* s = (System*)sv;
  int n = 0;
  for (auto& it : s->kfs[kf]->meas) {
    if (n < cap) { pt[n] = it.first; level[n] = it.second.level; root[2 * n] = it.second.root[0]; root[2 * n + 1] = it.second.root[1]; source[n] = it.second.source; }
    n++;
  }
  return n;
}

int orc_sys_get_template(void* sv, int pt, uint8_t* tmpl, int* sum, int* sumsq, int* bad) {
  System* s = (System*)sv;
  const Finder& f = s->pts[pt]->finder;
  if ((int)f.tmpl.size() >= f.P * f.P) memcpy(tmpl, f.tmpl.data(), f.P * f.P);   // points the map-maker created have no template until the tracker first searches them
  else memset(tmpl, 0, f.P * f.P);
  *sum = f.tsum; *sumsq = f.tsumsq; *bad = f.bad;
  return f.have_last;
}

int orc_sys_bundle_adjust_recent(void* sv) { System* s = (System*)sv; const int r = s->BundleAdjustRecent(); s->HandleBadPoints(); return r; }
int orc_sys_bundle_adjust_all(void* sv) { System* s = (System*)sv; const int r = s->BundleAdjustAll(); s->HandleBadPoints(); return r; }

// ---- stand-alone Bundle ------------------------------------------------------------------------------------
void* orc_ba_create(const double cam5[5], int width, int height, int quirks, int max_iterations, double convergence_limit, double min_sigma) {
  Bundle* b = new Bundle;
  b->camera.init(cam5, width, height, (quirks & ORC_Q_CAM_INT_RADIUS) != 0);
  b->max_iterations = max_iterations; b->convergence_limit = convergence_limit; b->min_sigma = min_sigma;
  return b;
}
void orc_ba_destroy(void* b) { delete (Bundle*)b; }
int orc_ba_add_camera(void* b, const double pose12[12], int fixed) { return ((Bundle*)b)->AddCamera(pose_from12(pose12), fixed != 0); }
int orc_ba_add_point(void* b, const double pos[3]) { return ((Bundle*)b)->AddPoint(v3(pos[0], pos[1], pos[2])); }
void orc_ba_add_meas(void* b, int cam, int point, const double pos[2], double sigma_squared) { ((Bundle*)b)->AddMeas(cam, point, pos, sigma_squared); }
int orc_ba_compute(void* b) { bool abort = false; return ((Bundle*)b)->Compute(&abort); }
void orc_ba_get_camera(void* b, int n, double pose12[12]) { pose_to12(((Bundle*)b)->cams.at(n).pose, pose12); }
void orc_ba_get_point(void* b, int n, double pos[3]) { const V3& p = ((Bundle*)b)->pts.at(n).pos; pos[0] = p[0]; pos[1] = p[1]; pos[2] = p[2]; }
int orc_ba_converged(void* b) { return ((Bundle*)b)->converged; }
int orc_ba_get_outlier_meas(void* bv, int* pc, int cap) {
  Bundle* b = (Bundle*)bv;
  int n = 0;
  for (auto& e : b->outlier_meas) { if (n < cap) { pc[2 * n] = e.first; pc[2 * n + 1] = e.second; } n++; }
  return n;
}
int orc_ba_get_outlier_points(void* bv, int* idx, int cap) {
  int n = 0;
  for (int i : ((Bundle*)bv)->GetOutliers()) { if (n < cap) idx[n] = i; n++; }
  return n;
}
void orc_ba_get_stats(void* bv, double* sigma2, double* lambda, long long* trials) {
  Bundle* b = (Bundle*)bv; *sigma2 = b->sigma2; *lambda = b->lambda; *trials = b->n_trials;
}

// ---- substrate ----------------------------------------------------------------------------------------------
void orc_se3_exp(const double mu[6], double pose12[12]) { pose_to12(se3_exp(mu), pose12); }
void orc_se3_ln(const double pose12[12], double mu[6]) { se3_ln(pose_from12(pose12), mu); }
void orc_cam_project(const double cam5[5], int w, int h, int quirks, double cx, double cy, double im[2], double derivs[4], int* invalid, double* largest_radius) {
  Camera c; c.init(cam5, w, h, (quirks & ORC_Q_CAM_INT_RADIUS) != 0);
  const Camera::Proj p = c.project(cx, cy);
  im[0] = p.im[0]; im[1] = p.im[1]; c.derivs(p, derivs); *invalid = p.invalid; *largest_radius = c.largest_radius;
}
void orc_cam_unproject(const double cam5[5], int w, int h, double ix, double iy, double out[2]) {
  Camera c; c.init(cam5, w, h, false); c.unproject(ix, iy, out);
}
double orc_find_sigma_squared(int est, const double* v, int n) { std::vector<double> x(v, v + n); return find_sigma_squared(est, x); }
double orc_weight(int est, double e2, double s2) { return weight(est, e2, s2); }
double orc_sqrt_weight(int est, double e2, double s2) { return sqrt_weight(est, e2, s2); }
double orc_objective(int est, double e2, double s2) { return objective(est, e2, s2); }
int orc_transform_image(const uint8_t* in, int iw, int ih, int istride, uint8_t* out, int P, const double M[4], const double inOrig[2], const double outOrig[2]) {
  return transform_image(in, iw, ih, istride, out, P, M, inOrig, outOrig);
}
int orc_zmssd(const uint8_t* tmpl, int P, const uint8_t* img, int w, int h, int stride, int icol, int irow) {
  Finder f; f.P = P; f.max_ssd = P * P * 500; f.tmpl.assign(tmpl, tmpl + P * P);
  int s = 0, sq = 0; for (int i = 0; i < P * P; i++) { s += tmpl[i]; sq += tmpl[i] * tmpl[i]; }
  f.tsum = s; f.tsumsq = sq;
  return finder_zmssd(f, img, w, h, stride, icol, irow);
}

}  // extern "C"

// PatchFinder sub-pixel refinement alone (jni/PatchFinder.cc:242-350) on a level-0 image: template P x P, start at the
// integer position (x, y); returns 1 if converged and writes the refined position.
extern "C" int orc_subpix_refine(const uint8_t* tmpl, int P, const uint8_t* img, int w, int h, int x, int y, int max_its, double out[2]) {
  Finder f; f.P = P; f.max_ssd = P * P * 500; f.tmpl.assign(tmpl, tmpl + P * P); f.level = 0;
  f.coarse[0] = x; f.coarse[1] = y;
  KeyFrame kf;
  kf.w[0] = w; kf.h[0] = h; kf.im[0].assign(img, img + (size_t)w * h);
  finder_make_subpix(f);
  const bool ok = finder_iterate_subpix_to_convergence(f, kf, max_its);
  out[0] = f.subpix[0]; out[1] = f.subpix[1];
  return ok ? 1 : 0;
}
