// vslam_system lifetime, parameters and read-back entry points of the C ABI (include/vslam_c.h).
#include "vslam_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void vslam_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vslam_last_error(void) { return g_err; }

extern "C" int vslam_default_params(vslam_params* p, int width, int height, int n_streams) {
  if (!p || width < 48 || height < 48 || n_streams < 1) { vslam_set_error("default_params: bad size"); return VSLAM_E_INVALID; }
  memset(p, 0, sizeof(*p));
  p->width = width; p->height = height; p->n_streams = n_streams;
  p->fast_threshold[0] = 10; p->fast_threshold[1] = 15; p->fast_threshold[2] = 15; p->fast_threshold[3] = 10;
  p->nonmax_barrier = 10;
  p->patch_size = 11;
  for (int l = 0; l < NLEV; l++) {
    const int n = ((width >> l) * (height >> l)) / 2;
    p->max_corners[l] = n < 256 ? 256 : n;
  }
  p->max_points = 4096;
  p->max_keyframes = 32;
  p->max_patches_per_frame = 1000;
  p->coarse_min = 20; p->coarse_max = 60; p->coarse_range = 30; p->coarse_subpix_its = 8;
  p->coarse_disabled = 0; p->coarse_min_vel = 0.006;
  p->fine_subpix_its = 8;
  p->wls_prior = 100.0;
  p->use_sbi = 0;
  p->min_frames_between_kf = 20;
  p->max_kf_dist_wiggle_mult = 0.2;
  p->wiggle_scale = 0.1;
  p->ba_max_iterations = 20;
  p->ba_convergence_limit = 1e-6;
  p->ba_min_tukey_sigma = 0.4;
  p->ba_window = 5;
  p->ba_min_keyframes = 8;
  p->cam[0] = 0.841906; p->cam[1] = 1.10893; p->cam[2] = 0.505171; p->cam[3] = 0.470265; p->cam[4] = -0.0133843;
  p->quirks = 0;
  p->device = 0;
  p->ba_delay_frames = 0;
  p->grow_map = 0;
  p->ba_batch_frames = 1;
  p->idle_iterations = 0;
  p->bootstrap = 0;
  p->ba_sum_order = 0;
  return VSLAM_OK;
}

template <class T>
static int dev_alloc(vslam_system* sys, T** out, size_t count) {
  void* ptr = nullptr;
  HIPCHK(hipMalloc(&ptr, count * sizeof(T) + 64));
  HIPCHK(hipMemsetAsync(ptr, 0, count * sizeof(T) + 64, sys->stream));
  sys->allocs.push_back(ptr);
  *out = (T*)ptr;
  return VSLAM_OK;
}
#define ALLOC(ptr, count) do { int _r = dev_alloc(sys, &(ptr), (count)); if (_r) { vslam_destroy(sys); return _r; } } while (0)

extern "C" int vslam_create(const vslam_params* p, vslam_system** out) {
  if (!p || !out) { vslam_set_error("create: null argument"); return VSLAM_E_INVALID; }
  if (p->width < 48 || p->height < 48 || p->width > 4096 || p->height > 4096 || p->n_streams < 1 ||
      (p->patch_size != 8 && p->patch_size != 11) || p->ba_delay_frames < 0 || p->ba_delay_frames >= 20 ||
      (p->ba_delay_frames > 0 && p->ba_delay_frames >= p->min_frames_between_kf) || p->ba_batch_frames < 0 ||
      (p->ba_batch_frames > 1 && p->ba_batch_frames > p->ba_delay_frames) || p->idle_iterations < -1 ||
      (p->idle_iterations != 0 && p->ba_delay_frames > 0)) {
    vslam_set_error("create: unsupported parameters: size %dx%d (48..4096), streams %d (>= 1), patch %d (8 or 11), ba_delay_frames %d (0..19 and below min_frames_between_kf %d), ba_batch_frames %d (0..ba_delay_frames), idle_iterations %d (>= -1, non-zero only with ba_delay_frames = 0)",
                    p->width, p->height, p->n_streams, p->patch_size, p->ba_delay_frames, p->min_frames_between_kf, p->ba_batch_frames, p->idle_iterations);
    return VSLAM_E_INVALID;
  }
  if (p->bootstrap && (p->grow_map == 0 || p->ba_delay_frames > 0)) {
    vslam_set_error("create: bootstrap needs grow_map != 0 (keyframe corner lists for InitFromStereo's AddSomeMapPoints) and ba_delay_frames = 0, got grow_map %d, ba_delay_frames %d", p->grow_map, p->ba_delay_frames);
    return VSLAM_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    vslam_set_error("create: no HIP device visible (the MI355X path has no CPU fallback)");
    return VSLAM_E_HIP;
  }
  HIPCHK(hipSetDevice(p->device));
  vslam_system* sys = new vslam_system();
  sys->p = *p;
  sys->S = p->n_streams;
  sys->have_frame = false;
  sys->stream = nullptr;
  // (default priorities: the tracker's stream at the highest priority, with or without the front end's at the lowest, was measured at
  // 3072 streams: 376 k against 398-400 k frames/s -- the front end of frame t+1 then finishes late and the tracker waits for it)
  if (hipStreamCreateWithFlags(&sys->stream, hipStreamNonBlocking) != hipSuccess) {
    vslam_set_error("create: hipStreamCreate failed"); delete sys; return VSLAM_E_HIP;
  }
  const int S = sys->S;
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device) == hipSuccess && ncu > 0) sys->n_cu = ncu;
  }
  if (hipStreamCreateWithFlags(&sys->fe_stream, hipStreamNonBlocking) != hipSuccess) { vslam_set_error("create: hipStreamCreate failed"); vslam_destroy(sys); return VSLAM_E_HIP; }
  for (int b = 0; b < 2; b++) {
    if (hipEventCreateWithFlags(&sys->ev_fe_done[b], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sys->ev_track_done[b], hipEventDisableTiming) != hipSuccess) { vslam_set_error("create: hipEventCreate failed"); vslam_destroy(sys); return VSLAM_E_HIP; }
  }
  for (int k = 0; k < 4; k++)
    if (hipEventCreate(&sys->ev_mm[k]) != hipSuccess) { vslam_set_error("create: hipEventCreate failed"); vslam_destroy(sys); return VSLAM_E_HIP; }
  for (int l = 0; l < NLEV; l++) {
    LevelGeom& g = sys->geom[l];
    g.w = p->width >> l; g.h = p->height >> l;
    g.pitch = (g.w + 63) & ~63;
    g.nchunk = (g.w + 63) >> 6;
    g.cap = p->max_corners[l];
    g.thr = p->fast_threshold[l];
  }
  for (int b = 0; b < 2; b++) {
    FrameDev& fr = sys->frbuf[b];
    for (int l = 0; l < NLEV; l++) {
      const LevelGeom& g = sys->geom[l];
      ALLOC(sys->d_lvl_buf[b][l], (size_t)S * g.pitch * g.h);
      ALLOC(fr.cmask[l], (size_t)S * g.h * g.nchunk);
      ALLOC(fr.rowcnt[l], (size_t)S * g.h);
      ALLOC(fr.rowlut[l], (size_t)S * (g.h + 1));
      ALLOC(fr.corners[l], (size_t)S * g.cap);
      ALLOC(fr.scores[l], (size_t)S * g.cap);
      ALLOC(fr.maxcorners[l], (size_t)S * g.cap);
      fr.img[l] = sys->d_lvl_buf[b][l];
      fr.img_sstride[l] = (size_t)g.pitch * g.h;
      fr.img_pitch[l] = g.pitch;
    }
    {
      const size_t ns = (size_t)(sys->geom[3].w / 2) * (sys->geom[3].h / 2);
      ALLOC(fr.sbi_small, (size_t)S * ns); ALLOC(fr.sbi_tmpl, (size_t)S * ns); ALLOC(fr.sbi_jacs, (size_t)S * ns * 2); ALLOC(fr.sbi_rot, (size_t)S * 8);
    }
    ALLOC(fr.ncorners, (size_t)S * NLEV);
    ALLOC(fr.nmax, (size_t)S * NLEV);
    ALLOC(fr.overflow, 1);
  }
  for (int l = 0; l < NLEV; l++) { ALLOC(sys->cand[l], (size_t)S * sys->geom[l].cap); ALLOC(sys->cand_score[l], (size_t)S * sys->geom[l].cap); }
  ALLOC(sys->ncand, (size_t)S * NLEV);
  sys->have_candidates = false;
  sys->have_sbi = false;
  if (p->ba_delay_frames > 0) {
    const int nb = p->ba_delay_frames < 8 ? p->ba_delay_frames : 8;
    for (int i = 0; i < nb; i++) {
      hipStream_t st = nullptr;
      // (default priority: giving the map-maker's streams the lowest one was measured -- the adjustments then finish late and the
      // frames that apply them wait: 215 k against 243 k frames/s; the highest one changes nothing: 340 k either way at 2048 streams)
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { vslam_set_error("create: hipStreamCreate failed"); vslam_destroy(sys); return VSLAM_E_HIP; }
      sys->ba_streams.push_back(st);
    }
    sys->ba_stream = sys->ba_streams[0];
    sys->frame_batch.assign((size_t)p->ba_delay_frames + 2, -1L);
    for (int i = 0; i < p->ba_delay_frames + 2; i++) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) { vslam_set_error("create: hipEventCreate failed"); vslam_destroy(sys); return VSLAM_E_HIP; }
      sys->ev_asm.push_back(a); sys->ev_ba.push_back(b);
    }
  }
  sys->fr_idx = 0;
  sys->fr = sys->frbuf[0];
  for (int l = 0; l < NLEV; l++) sys->d_lvl[l] = sys->d_lvl_buf[0][l];
  sys->ba_ws = nullptr;
  {
    int r = trk_alloc(sys);
    if (!r) r = ba_alloc(sys);
    if (!r) r = grow_alloc(sys);
    if (!r) r = boot_alloc(sys);
    if (!r && hipStreamSynchronize(sys->stream) != hipSuccess) r = VSLAM_E_HIP;
    if (!r) r = map_init_states(sys);
    if (r) { vslam_destroy(sys); return r; }
  }
  if (hipStreamSynchronize(sys->stream) != hipSuccess) { vslam_set_error("create: sync failed"); vslam_destroy(sys); return VSLAM_E_HIP; }
  *out = sys;
  return VSLAM_OK;
}

extern "C" int vslam_destroy(vslam_system* sys) {
  if (!sys) return VSLAM_OK;
  if (sys->fe_stream) (void)hipStreamSynchronize(sys->fe_stream);
  for (hipStream_t st : sys->ba_streams) (void)hipStreamSynchronize(st);
  if (sys->stream) (void)hipStreamSynchronize(sys->stream);
  for (hipEvent_t e : sys->ev_asm) (void)hipEventDestroy(e);
  for (hipEvent_t e : sys->ev_ba) (void)hipEventDestroy(e);
  for (hipStream_t st : sys->ba_streams) (void)hipStreamDestroy(st);
  for (void* p : sys->allocs) (void)hipFree(p);
  for (hipEvent_t e : sys->prof_ev) (void)hipEventDestroy(e);
  for (int k = 0; k < 4; k++) if (sys->ev_mm[k]) (void)hipEventDestroy(sys->ev_mm[k]);
  for (int b = 0; b < 2; b++) { if (sys->ev_fe_done[b]) (void)hipEventDestroy(sys->ev_fe_done[b]); if (sys->ev_track_done[b]) (void)hipEventDestroy(sys->ev_track_done[b]); }
  if (sys->fe_stream) (void)hipStreamDestroy(sys->fe_stream);
  if (sys->stream) (void)hipStreamDestroy(sys->stream);
  delete sys;
  return VSLAM_OK;
}

extern "C" int vslam_synchronize(vslam_system* sys) {
  if (!sys) return VSLAM_E_INVALID;
  HIPCHK(hipStreamSynchronize(sys->stream));
  return ba_sync_streams(sys);
}

extern "C" int vslam_make_keyframe_lite(vslam_system* sys, const uint8_t* gray, size_t row_stride, size_t stream_stride,
                                        int on_device) {
  if (!sys) return VSLAM_E_INVALID;
  return fe_make_keyframe_lite(sys, gray, row_stride, stream_stride, on_device);
}

extern "C" int vslam_fast_nonmax(vslam_system* sys) {
  if (!sys) return VSLAM_E_INVALID;
  return fe_fast_nonmax(sys);
}

static int check_sl(vslam_system* sys, int stream, int level) {
  if (!sys || stream < 0 || stream >= sys->S || level < 0 || level >= NLEV) { vslam_set_error("bad stream/level"); return VSLAM_E_INVALID; }
  if (!sys->have_frame) { vslam_set_error("no current frame"); return VSLAM_E_STATE; }
  return VSLAM_OK;
}

static int check_overflow(vslam_system* sys) {
  int ov = 0;
  HIPCHK(hipMemcpyAsync(&ov, sys->fr.overflow, sizeof(int), hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  if (ov) { vslam_set_error("corner capacity exceeded (raise vslam_params.max_corners)"); return VSLAM_E_CAPACITY; }
  return VSLAM_OK;
}

extern "C" int vslam_read_level_image(vslam_system* sys, int stream, int level, uint8_t* dst, size_t dst_stride) {
  int r = check_sl(sys, stream, level); if (r) return r;
  const LevelGeom& g = sys->geom[level];
  HIPCHK(hipMemcpy2DAsync(dst, dst_stride, sys->fr.img[level] + (size_t)stream * sys->fr.img_sstride[level],
                          sys->fr.img_pitch[level], g.w, g.h, hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  return VSLAM_OK;
}

extern "C" int vslam_read_corners(vslam_system* sys, int stream, int level, uint32_t* corners, int cap, int* n) {
  int r = check_sl(sys, stream, level); if (r) return r;
  r = check_overflow(sys); if (r) return r;
  int cnt = 0;
  HIPCHK(hipMemcpyAsync(&cnt, sys->fr.ncorners + stream * NLEV + level, sizeof(int), hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  if (n) *n = cnt;
  const int m = cnt < cap ? cnt : cap;
  if (corners && m > 0) {
    HIPCHK(hipMemcpyAsync(corners, sys->fr.corners[level] + (size_t)stream * sys->geom[level].cap, (size_t)m * 4,
                          hipMemcpyDeviceToHost, sys->stream));
    HIPCHK(hipStreamSynchronize(sys->stream));
  }
  return VSLAM_OK;
}

extern "C" int vslam_read_row_lut(vslam_system* sys, int stream, int level, int* lut) {
  int r = check_sl(sys, stream, level); if (r) return r;
  const int h = sys->geom[level].h;
  HIPCHK(hipMemcpyAsync(lut, sys->fr.rowlut[level] + (size_t)stream * (h + 1), (size_t)h * 4, hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  return VSLAM_OK;
}

extern "C" int vslam_read_sbi(vslam_system* sys, int stream, uint8_t* small_img, float* tmpl, double rot8[8]) {
  int r = check_sl(sys, stream, 0); if (r) return r;
  if (!sys->p.use_sbi || !sys->have_sbi) { vslam_set_error("read_sbi: use_sbi is off or no frame yet"); return VSLAM_E_STATE; }
  HIPCHK(hipStreamSynchronize(sys->fe_stream));
  const size_t ns = (size_t)(sys->geom[3].w / 2) * (sys->geom[3].h / 2);
  if (small_img) HIPCHK(hipMemcpy(small_img, sys->fr.sbi_small + stream * ns, ns, hipMemcpyDeviceToHost));
  if (tmpl) HIPCHK(hipMemcpy(tmpl, sys->fr.sbi_tmpl + stream * ns, ns * sizeof(float), hipMemcpyDeviceToHost));
  if (rot8) HIPCHK(hipMemcpy(rot8, sys->fr.sbi_rot + (size_t)stream * 8, 8 * sizeof(double), hipMemcpyDeviceToHost));
  return VSLAM_OK;
}

extern "C" int vslam_make_keyframe_rest(vslam_system* sys, double min_shi_tomasi_score) {
  if (!sys) return VSLAM_E_INVALID;
  return fe_make_keyframe_rest(sys, min_shi_tomasi_score);
}

extern "C" int vslam_thin_candidates(vslam_system* sys, int keyframe) {
  if (!sys) return VSLAM_E_INVALID;
  return fe_thin_candidates(sys, keyframe);
}

extern "C" int vslam_read_candidates(vslam_system* sys, int stream, int level, uint32_t* pos, double* score, int cap, int* n) {
  int r = check_sl(sys, stream, level); if (r) return r;
  if (!sys->have_candidates) { vslam_set_error("read_candidates: call vslam_make_keyframe_rest first"); return VSLAM_E_STATE; }
  int cnt = 0;
  HIPCHK(hipMemcpyAsync(&cnt, sys->ncand + stream * NLEV + level, sizeof(int), hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  if (n) *n = cnt;
  const size_t off = (size_t)stream * sys->geom[level].cap;
  const size_t m = (size_t)(cnt < cap ? cnt : cap);
  if (pos && m) HIPCHK(hipMemcpyAsync(pos, sys->cand[level] + off, m * 4, hipMemcpyDeviceToHost, sys->stream));
  if (score && m) HIPCHK(hipMemcpyAsync(score, sys->cand_score[level] + off, m * 8, hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  return VSLAM_OK;
}

extern "C" int vslam_read_max_corners(vslam_system* sys, int stream, int level, uint32_t* corners, int* scores, int cap,
                                      int* n) {
  int r = check_sl(sys, stream, level); if (r) return r;
  int cnt = 0, nc = 0;
  HIPCHK(hipMemcpyAsync(&cnt, sys->fr.nmax + stream * NLEV + level, sizeof(int), hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipMemcpyAsync(&nc, sys->fr.ncorners + stream * NLEV + level, sizeof(int), hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  if (n) *n = cnt;
  const size_t off = (size_t)stream * sys->geom[level].cap;
  if (corners && cnt > 0)
    HIPCHK(hipMemcpyAsync(corners, sys->fr.maxcorners[level] + off, (size_t)(cnt < cap ? cnt : cap) * 4, hipMemcpyDeviceToHost, sys->stream));
  if (scores && nc > 0)  // scores of ALL corners of the level (same order as vslam_read_corners)
    HIPCHK(hipMemcpyAsync(scores, sys->fr.scores[level] + off, (size_t)(nc < cap ? nc : cap) * 4, hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  return VSLAM_OK;
}

// ---- self-test hook: the transcendentals of vslam_libm.h evaluated on the device ---------------------------------------
__global__ void k_eval_transcendental(int fn, int n, const double* x, double* y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r;
  switch (fn) {
    case 0: r = vlm::vsin(v); break;
    case 1: r = vlm::vcos(v); break;
    case 2: r = vlm::vtan(v); break;
    case 3: r = vlm::vatan(v); break;
    case 4: r = vlm::vasin(v); break;
    case 5: r = vlm::vacos(v); break;
    case 6: r = sqrt(v); break;
    default: r = 1.0 / v; break;
  }
  y[i] = r;
}

extern "C" int vslam_eval_transcendental(int fn, int n, const double* x, double* y, int on_host) {
  if (fn < 0 || fn > 7 || n < 0 || !x || !y) { vslam_set_error("eval_transcendental: bad argument"); return VSLAM_E_INVALID; }
  if (on_host) {                                      // the same source compiled for the host (what cam_fill and the oracle evaluate)
    for (int i = 0; i < n; i++) {
      const double v = x[i];
      y[i] = fn == 0 ? vlm::vsin(v) : fn == 1 ? vlm::vcos(v) : fn == 2 ? vlm::vtan(v) : fn == 3 ? vlm::vatan(v) : fn == 4 ? vlm::vasin(v)
           : fn == 5 ? vlm::vacos(v) : fn == 6 ? sqrt(v) : 1.0 / v;
    }
    return VSLAM_OK;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { vslam_set_error("eval_transcendental: no HIP device visible"); return VSLAM_E_HIP; }
  if (n == 0) return VSLAM_OK;
  double *dx = nullptr, *dy = nullptr;
  auto run = [&]() -> int {
    HIPCHK(hipMalloc((void**)&dx, sizeof(double) * n));
    HIPCHK(hipMalloc((void**)&dy, sizeof(double) * n));
    HIPCHK(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_eval_transcendental, dim3((n + 255) / 256), dim3(256), 0, 0, fn, n, dx, dy);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(y, dy, sizeof(double) * n, hipMemcpyDeviceToHost));
    return VSLAM_OK;
  };
  const int rc = run();
  if (dx) (void)hipFree(dx);
  if (dy) (void)hipFree(dy);
  return rc;
}
