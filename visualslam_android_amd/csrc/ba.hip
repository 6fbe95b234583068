// Mapping side of the hot path on device:
//   * vslam_bundle_*  : stand-alone, batched Bundle (jni/Bundle.h:111-121), one persistent workgroup per problem
//   * MapMaker::AddKeyFrame (jni/MapMaker.cc:470-506) + BundleAdjustRecent/All (:776-851) + BundleAdjust (:854-960)
//     + HandleBadPoints (:140-164) for every stream of a system, driven by the tracker's device-side decision
//     (kf_pending) -- no host synchronisation between TrackFrame and the local bundle adjustment.  With
//     vslam_params.ba_delay_frames = D > 0 Bundle::Compute runs on the map-maker stream and its result is applied at the
//     start of frame t + D (ba_frame_start), the deterministic stand-in for PTAM's map-maker thread.
#include "vslam_internal.h"
#include "ba_device.h"
#include "ba_ordered.h"
#define BA_MAX_KF 128        // keyframes per stream = cameras of a bundle-adjustment problem (8-bit camera field of a slot: < 256)
#include <string.h>
#include <stdlib.h>

struct BaPool {            // device arrays for N problems
  int N, max_cams, max_pts, max_meas, max_free;
  BaResult* res;
  Pose* cam_pose; Pose* cam_new; int* cam_fixed; int* cam_row; double* cam_U; double* cam_ea;
  double* pt_pos; double* pt_new; double* pt_V; double* pt_eb; int* pt_nmeas; int* pt_nout;
  int* ms_p; int* ms_c; int* ms_state; double* ms_found; double* ms_sin;
  int* lut;
  int* sl_info; int* sl_pt; int* sl_logical; double* sl_found; double* sl_sin; double* sl_cm; double* sl_d; double* sl_eps;
  int* pt_offF; int* pt_offX; unsigned long long* pt_maskF; int* chF; int* chX; int* ch_n;
  double* S; double* E; double* cam_up; double* scratch; int* outl; int* free_cams;
  int* id_view; int* id_point;   // BundleAdjust translation tables (:861-864)
  // asynchronous map-maker: the problems k_ba_assemble built in frame t, as a list per slot (t mod slots).  The compute launch
  // of that frame walks exactly this list: it never looks at a problem a later frame's assemble (running beside it on the
  // main stream) is still writing.
  int* work; int* work_n; int work_slots;
  int* view_of_kf;               // [N][BA_MAX_KF] keyframe index -> camera id of the problem being assembled, or -1 (k_ba_select -> k_ba_assemble)
  // measurement: what every k_ba_compute launch of a system actually ran, one record of BA_LSTAT_N counters per launch in a ring
  // (problems, LM trials, and the trial-weighted sums of measurements, cameras, points, points x pairs of adjustable cameras,
  // (6 n_free)^3 that SURVEY.md 8(d)'s byte and flop formulas need); null for the stand-alone Bundle
  unsigned long long* lstat;
  // parity mode (vslam_params.ba_sum_order = 1, ba_ordered.h): the per-slot terms of every sum; null otherwise
  int* o_of_logical; double* o_obj; double* o_v9; double* o_u27; double* o_W; double* o_Y; double* o_ve; double* o_up;
};
#define BA_LSTAT_N 8
#define BA_LSTAT_RING 1024

__host__ __device__ inline BaView ba_view(const BaPool& b, int n) {
  BaView v;
  const size_t C = b.max_cams, P = b.max_pts, M = b.max_meas, F = (size_t)b.max_free * 6;
  v.max_cams = b.max_cams; v.max_pts = b.max_pts; v.max_meas = b.max_meas;
  v.res = b.res + n;
  v.cam_pose = b.cam_pose + n * C; v.cam_new = b.cam_new + n * C; v.cam_fixed = b.cam_fixed + n * C; v.cam_row = b.cam_row + n * C;
  v.cam_U = b.cam_U + n * C * 36; v.cam_ea = b.cam_ea + n * C * 6;
  v.pt_pos = b.pt_pos + n * P * 3; v.pt_new = b.pt_new + n * P * 3; v.pt_V = b.pt_V + n * P * 6; v.pt_eb = b.pt_eb + n * P * 3;
  v.pt_nmeas = b.pt_nmeas + n * P; v.pt_nout = b.pt_nout + n * P;
  v.ms_p = b.ms_p + n * M; v.ms_c = b.ms_c + n * M; v.ms_state = b.ms_state + n * M; v.ms_found = b.ms_found + n * M * 2; v.ms_sin = b.ms_sin + n * M;
  v.lut = b.lut + n * C * P;
  v.sl_info = b.sl_info + n * M; v.sl_pt = b.sl_pt + n * M; v.sl_logical = b.sl_logical + n * M; v.sl_found = b.sl_found + n * M * 2; v.sl_sin = b.sl_sin + n * M;
  v.sl_cm = b.sl_cm + n * M * 3; v.sl_d = b.sl_d + n * M * 4; v.sl_eps = b.sl_eps + n * M * 2;
  v.pt_offF = b.pt_offF + n * (P + 1); v.pt_offX = b.pt_offX + n * (P + 1); v.pt_maskF = b.pt_maskF + n * P;
  v.chF = b.chF + n * (P + 2); v.chX = b.chX + n * (P + 2); v.ch_n = b.ch_n + n * 4;
  v.S = b.S + n * F * F; v.E = b.E + n * F; v.cam_up = b.cam_up + n * F;
  v.scratch = b.scratch + n * M; v.outl = b.outl + n * M * 2; v.free_cams = b.free_cams + n * C;
  return v;
}

__host__ __device__ inline BaOrdView ba_ord_view(const BaPool& b, int n) {
  BaOrdView o;
  const size_t P = b.max_pts, M = b.max_meas;
  o.of_logical = b.o_of_logical + n * M; o.obj = b.o_obj + n * M; o.v9 = b.o_v9 + n * M * 9; o.u27 = b.o_u27 + n * M * 27;
  o.W = b.o_W + n * M * 18; o.Y = b.o_Y + n * M * 18; o.ve = b.o_ve + n * P * 3; o.up = b.o_up + n * P * 3;
  return o;
}

#ifndef VSLAM_BA_WPE
#define VSLAM_BA_WPE 2
#endif
__global__ __launch_bounds__(BA_THREADS) __attribute__((amdgpu_waves_per_eu(VSLAM_BA_WPE, VSLAM_BA_WPE))) void k_ba_compute(BaPool pool, BaConfig cfg, int slot, int lrec /* record of the launch in pool.lstat, or -1 */) {
  // slot < 0: one workgroup per problem of the pool (synchronous map-maker, stand-alone Bundle); slot >= 0: the grid walks the
  // work list of one frame (asynchronous map-maker)
  // (the persistent workgroups of an asynchronous launch draw their next list entry from a counter: the long and the short
  // adjustments of a batch balance themselves instead of leaving the workgroups that drew two long ones alone at the end)
  const int count = slot < 0 ? pool.N : pool.work_n[slot];
  __shared__ int s_next;
  int i = blockIdx.x;
  while (i < count) {
    const int n = slot < 0 ? i : pool.work[(size_t)slot * pool.N + i];
    const BaView v = ba_view(pool, n);
    if (v.res->active && !v.res->computed) {
      ba_compute(v, cfg);
      if (threadIdx.x == 0) {
        v.res->computed = 1;
        if (lrec >= 0 && pool.lstat) {                                // the work this launch did, for the roofline of THIS launch
          unsigned long long* L = pool.lstat + (size_t)lrec * BA_LSTAT_N;
          const unsigned long long t = (unsigned long long)v.res->trials, nf = (unsigned long long)v.res->n_free;
          atomicAdd(&L[0], 1ull); atomicAdd(&L[1], t);
          atomicAdd(&L[2], t * (unsigned long long)v.res->n_meas); atomicAdd(&L[3], t * (unsigned long long)v.res->n_cams); atomicAdd(&L[4], t * (unsigned long long)v.res->n_pts);
          atomicAdd(&L[5], t * (unsigned long long)v.res->n_pts * (nf * (nf > 0 ? nf - 1 : 0) / 2)); atomicAdd(&L[6], t * (6 * nf) * (6 * nf) * (6 * nf));
        }
      }
    }
    if (slot < 0) break;                                            // one workgroup per problem
    __syncthreads();
    if (threadIdx.x == 0) s_next = (int)gridDim.x + atomicAdd(&pool.work_n[pool.work_slots + slot], 1);
    __syncthreads();
    i = s_next;
  }
}

// The same launch in the parity mode (vslam_params.ba_sum_order = 1): every sum in the reference's order (ba_ordered.h).  A kernel of
// its own, so that its registers and LDS do not weigh on the fast path's.
// (the same register bound as k_ba_compute: the two kernels share the out-of-line phase functions, which are compiled once)
__global__ __launch_bounds__(BA_THREADS) __attribute__((amdgpu_waves_per_eu(VSLAM_BA_WPE, VSLAM_BA_WPE))) void k_ba_compute_ordered(BaPool pool, BaConfig cfg, int slot, int lrec) {
  const int count = slot < 0 ? pool.N : pool.work_n[slot];
  __shared__ int s_next;
  int i = blockIdx.x;
  while (i < count) {
    const int n = slot < 0 ? i : pool.work[(size_t)slot * pool.N + i];
    const BaView v = ba_view(pool, n);
    if (v.res->active && !v.res->computed) {
      ba_compute_ordered(v, cfg, ba_ord_view(pool, n));
      if (threadIdx.x == 0) {
        v.res->computed = 1;
        if (lrec >= 0 && pool.lstat) {
          unsigned long long* L = pool.lstat + (size_t)lrec * BA_LSTAT_N;
          const unsigned long long t = (unsigned long long)v.res->trials, nf = (unsigned long long)v.res->n_free;
          atomicAdd(&L[0], 1ull); atomicAdd(&L[1], t);
          atomicAdd(&L[2], t * (unsigned long long)v.res->n_meas); atomicAdd(&L[3], t * (unsigned long long)v.res->n_cams); atomicAdd(&L[4], t * (unsigned long long)v.res->n_pts);
          atomicAdd(&L[5], t * (unsigned long long)v.res->n_pts * (nf * (nf > 0 ? nf - 1 : 0) / 2)); atomicAdd(&L[6], t * (6 * nf) * (6 * nf) * (6 * nf));
        }
      }
    }
    if (slot < 0) break;
    __syncthreads();
    if (threadIdx.x == 0) s_next = (int)gridDim.x + atomicAdd(&pool.work_n[pool.work_slots + slot], 1);
    __syncthreads();
    i = s_next;
  }
}

static void ba_launch_compute(const BaPool& pool, const BaConfig& cfg, int grid, hipStream_t st, int slot, int lrec) {
  static const int dyn_lds = getenv("VSLAM_BA_DYN_LDS") ? atoi(getenv("VSLAM_BA_DYN_LDS")) : 0;   // diagnostic: extra LDS per workgroup, to lower the workgroups per CU
  if (cfg.sum_order) hipLaunchKernelGGL(k_ba_compute_ordered, dim3(grid), dim3(BA_THREADS), 0, st, pool, cfg, slot, lrec);
  else hipLaunchKernelGGL(k_ba_compute, dim3(grid), dim3(BA_THREADS), dyn_lds, st, pool, cfg, slot, lrec);
}

// The arrays of a pool are carved out of ONE device allocation (256-byte aligned each): some forty separate allocations, most of them
// below the 2 MB that earns large page-table fragments, cost a launch over thousands of problems milliseconds of address-translation
// misses in its first phase (observed: Bundle::Compute of 1024 small problems 4 ms or 26-31 ms from run to run).
struct PoolArena { char* base; size_t off; };
template <class T> static void arena_take(PoolArena& a, T** out, size_t count) {
  a.off = (a.off + 255) & ~(size_t)255;
  if (a.base) *out = (T*)(a.base + a.off);
  a.off += count * sizeof(T);
}
#define PALLOC(field, count) arena_take(a, &b.field, (count))

static int pool_create(BaPool& b, std::vector<void*>& allocs, hipStream_t st, int N, int max_cams, int max_pts, int max_meas, int work_slots = 1, bool ordered = false, size_t lstat_records = 0) {
  b.N = N; b.max_cams = max_cams; b.max_pts = max_pts; b.max_meas = max_meas; b.max_free = max_cams;
  b.work_slots = work_slots > 0 ? work_slots : 1;
  const size_t n = N, C = max_cams, P = max_pts, M = max_meas, F = (size_t)max_cams * 6;
  auto carve = [&](PoolArena& a) {
    PALLOC(res, n);
    PALLOC(cam_pose, n * C); PALLOC(cam_new, n * C); PALLOC(cam_fixed, n * C); PALLOC(cam_row, n * C); PALLOC(cam_U, n * C * 36); PALLOC(cam_ea, n * C * 6);
    PALLOC(pt_pos, n * P * 3); PALLOC(pt_new, n * P * 3); PALLOC(pt_V, n * P * 6); PALLOC(pt_eb, n * P * 3);
    PALLOC(pt_nmeas, n * P); PALLOC(pt_nout, n * P);
    PALLOC(ms_p, n * M); PALLOC(ms_c, n * M); PALLOC(ms_state, n * M); PALLOC(ms_found, n * M * 2); PALLOC(ms_sin, n * M);
    PALLOC(lut, n * C * P);
    PALLOC(sl_info, n * M); PALLOC(sl_pt, n * M); PALLOC(sl_logical, n * M); PALLOC(sl_found, n * M * 2); PALLOC(sl_sin, n * M);
    PALLOC(sl_cm, n * M * 3); PALLOC(sl_d, n * M * 4); PALLOC(sl_eps, n * M * 2);
    PALLOC(pt_offF, n * (P + 1)); PALLOC(pt_offX, n * (P + 1)); PALLOC(pt_maskF, n * P); PALLOC(chF, n * (P + 2)); PALLOC(chX, n * (P + 2)); PALLOC(ch_n, n * 4);
    PALLOC(S, n * F * F); PALLOC(E, n * F); PALLOC(cam_up, n * F);
    PALLOC(scratch, n * M); PALLOC(outl, n * M * 2); PALLOC(free_cams, n * C); PALLOC(id_view, n * C); PALLOC(id_point, n * P);
    PALLOC(work, n * b.work_slots); PALLOC(work_n, (size_t)2 * b.work_slots);   /* work_n[work_slots + slot]: the launch's draw counter */ PALLOC(view_of_kf, n * BA_MAX_KF);
    b.lstat = nullptr;
    if (lstat_records) PALLOC(lstat, lstat_records * BA_LSTAT_N);
    b.o_of_logical = nullptr; b.o_obj = b.o_v9 = b.o_u27 = b.o_W = b.o_Y = b.o_ve = b.o_up = nullptr;
    if (ordered) {
      PALLOC(o_of_logical, n * M); PALLOC(o_obj, n * M); PALLOC(o_v9, n * M * 9); PALLOC(o_u27, n * M * 27); PALLOC(o_W, n * M * 18); PALLOC(o_Y, n * M * 18);
      PALLOC(o_ve, n * P * 3); PALLOC(o_up, n * P * 3);
    }
  };
  PoolArena a = {nullptr, 0};
  carve(a);                                                           // sizes only
  const size_t bytes = a.off + 256;
  void* ptr = nullptr;
  HIPCHK(hipMalloc(&ptr, bytes));
  allocs.push_back(ptr);
  HIPCHK(hipMemsetAsync(ptr, 0, bytes, st));
  a.base = (char*)ptr; a.off = 0;
  carve(a);
  return VSLAM_OK;
}
#undef PALLOC

static BaConfig make_cfg(const TrackParams& tp) {
  BaConfig c; c.cam = tp.cam; c.max_iterations = tp.ba_max_iterations; c.convergence_limit = tp.ba_convergence_limit; c.min_sigma2 = tp.ba_min_sigma2;
  c.sum_order = tp.ba_sum_order;
  return c;
}

// =================================================================================================================
// stand-alone batched Bundle
// =================================================================================================================
struct HostProblem { std::vector<Pose> cams; std::vector<int> fixed; std::vector<double> pts; std::vector<int> mp, mc; std::vector<double> mfound, msin; };

struct vslam_bundle {
  BaPool pool; std::vector<void*> allocs; hipStream_t stream; BaConfig cfg; TrackParams tp;
  std::vector<HostProblem> host; bool uploaded;
  hipEvent_t ev[2] = {nullptr, nullptr};      // around the last Bundle::Compute launch (vslam_bundle_get_timing)
  // Compute() overwrites cameras, points and results; a second Compute() of the same problems restores them from these device-side
  // copies of what the caller added instead of staging everything from the host again (dirty: the host side has changed since)
  bool dirty = true; BaResult* res0 = nullptr; Pose* cam0 = nullptr; double* pt0 = nullptr;
};



extern "C" int vslam_bundle_create(const vslam_params* p, int n_problems, int max_cameras, int max_points, int max_meas, vslam_bundle** out) {
  if (!p || !out || n_problems < 1 || max_cameras < 1 || max_cameras > BA_MAX_KF || max_points < 1 || max_points > 4096 || max_meas < 1 || max_meas > 65536) {
    vslam_set_error("bundle_create: bad argument (1..128 cameras of which at most 64 adjustable, 1..4096 points, 1..65536 measurements per problem)"); return VSLAM_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { vslam_set_error("bundle_create: no HIP device visible (no CPU fallback)"); return VSLAM_E_HIP; }
  HIPCHK(hipSetDevice(p->device));
  vslam_bundle* b = new vslam_bundle();
  b->uploaded = false;
  HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  trk_fill_params(*p, b->tp);
  b->cfg = make_cfg(b->tp);
  int r = pool_create(b->pool, b->allocs, b->stream, n_problems, max_cameras, max_points, max_meas, 1, b->cfg.sum_order != 0, 1);   // one launch record (vslam_bundle_get_timing)
  if (!r && (hipEventCreate(&b->ev[0]) != hipSuccess || hipEventCreate(&b->ev[1]) != hipSuccess)) { vslam_set_error("bundle_create: hipEventCreate failed"); r = VSLAM_E_HIP; }
  if (!r) {
    void* q = nullptr;
    if (hipMalloc(&q, sizeof(BaResult) * n_problems + 64) == hipSuccess) { b->res0 = (BaResult*)q; b->allocs.push_back(q); } else r = VSLAM_E_HIP;
    if (!r && hipMalloc(&q, sizeof(Pose) * (size_t)n_problems * max_cameras + 64) == hipSuccess) { b->cam0 = (Pose*)q; b->allocs.push_back(q); } else if (!r) r = VSLAM_E_HIP;
    if (!r && hipMalloc(&q, sizeof(double) * 3 * (size_t)n_problems * max_points + 64) == hipSuccess) { b->pt0 = (double*)q; b->allocs.push_back(q); } else if (!r) r = VSLAM_E_HIP;
    if (r) vslam_set_error("bundle_create: hipMalloc failed");
  }
  if (r) { vslam_bundle_destroy(b); return r; }
  b->host.resize(n_problems);
  HIPCHK(hipStreamSynchronize(b->stream));
  *out = b;
  return VSLAM_OK;
}

extern "C" int vslam_bundle_destroy(vslam_bundle* b) {
  if (!b) return VSLAM_OK;
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  for (void* p : b->allocs) (void)hipFree(p);
  for (int k = 0; k < 2; k++) if (b->ev[k]) (void)hipEventDestroy(b->ev[k]);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return VSLAM_OK;
}

#define BCHECK(cond, msg) do { if (!(cond)) { vslam_set_error("bundle: %s", msg); return VSLAM_E_INVALID; } } while (0)

extern "C" int vslam_bundle_add_camera(vslam_bundle* b, int n, const double pose12[12], int fixed) {
  BCHECK(b && n >= 0 && n < b->pool.N && pose12, "bad camera argument");
  HostProblem& h = b->host[n];
  if ((int)h.cams.size() >= b->pool.max_cams) { vslam_set_error("bundle: camera capacity"); return VSLAM_E_CAPACITY; }
  Pose p; for (int i = 0; i < 9; i++) p.R[i] = pose12[i]; for (int i = 0; i < 3; i++) p.t[i] = pose12[9 + i];
  h.cams.push_back(p); h.fixed.push_back(fixed ? 1 : 0);
  b->dirty = true;
  return (int)h.cams.size() - 1;
}

extern "C" int vslam_bundle_add_point(vslam_bundle* b, int n, const double pos[3]) {
  BCHECK(b && n >= 0 && n < b->pool.N && pos, "bad point argument");
  HostProblem& h = b->host[n];
  if ((int)h.pts.size() / 3 >= b->pool.max_pts) { vslam_set_error("bundle: point capacity"); return VSLAM_E_CAPACITY; }
  double q[3] = {pos[0], pos[1], pos[2]};
  if (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] != q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) q[0] = q[1] = q[2] = 0;   // NaN guard, jni/Bundle.cc:93-96
  h.pts.insert(h.pts.end(), q, q + 3);
  b->dirty = true;
  return (int)h.pts.size() / 3 - 1;
}

extern "C" int vslam_bundle_add_meas(vslam_bundle* b, int n, int cam, int point, const double pos[2], double sigma_squared) {
  BCHECK(b && n >= 0 && n < b->pool.N && pos, "bad measurement argument");
  HostProblem& h = b->host[n];
  BCHECK(cam >= 0 && cam < (int)h.cams.size() && point >= 0 && point < (int)h.pts.size() / 3, "measurement refers to an unknown camera/point");   // asserts :107-108
  if ((int)h.mp.size() >= b->pool.max_meas) { vslam_set_error("bundle: measurement capacity"); return VSLAM_E_CAPACITY; }
  h.mp.push_back(point); h.mc.push_back(cam); h.mfound.push_back(pos[0]); h.mfound.push_back(pos[1]);
  h.msin.push_back(sqrt(1.0 / sigma_squared));   // :115
  b->dirty = true;
  return VSLAM_OK;
}

// Bundle::AddCamera / AddPoint / AddMeas for a whole problem at once (replaces what the problem held): n_cams poses (12 doubles each)
// and fixed flags, n_pts positions, n_meas measurements (camera, point, position, sigma squared) in AddMeas order.
extern "C" int vslam_bundle_set_problem(vslam_bundle* b, int n, int n_cams, const double* pose12, const int* fixed, int n_pts, const double* pos3,
                                        int n_meas, const int* cam, const int* point, const double* xy, const double* sigma_squared) {
  BCHECK(b && n >= 0 && n < b->pool.N && n_cams >= 0 && n_pts >= 0 && n_meas >= 0, "set_problem: bad argument");
  BCHECK((n_cams == 0 || (pose12 && fixed)) && (n_pts == 0 || pos3) && (n_meas == 0 || (cam && point && xy && sigma_squared)), "set_problem: null array");
  if (n_cams > b->pool.max_cams || n_pts > b->pool.max_pts || n_meas > b->pool.max_meas) { vslam_set_error("bundle: set_problem exceeds the capacity"); return VSLAM_E_CAPACITY; }
  HostProblem& h = b->host[n];
  h = HostProblem();
  b->dirty = true;
  for (int c = 0; c < n_cams; c++) { Pose p; for (int i = 0; i < 9; i++) p.R[i] = pose12[12 * c + i]; for (int i = 0; i < 3; i++) p.t[i] = pose12[12 * c + 9 + i]; h.cams.push_back(p); h.fixed.push_back(fixed[c] ? 1 : 0); }
  for (int i = 0; i < n_pts; i++) {
    double q[3] = {pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]};
    if (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] != q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) q[0] = q[1] = q[2] = 0;   // NaN guard, jni/Bundle.cc:93-96
    h.pts.insert(h.pts.end(), q, q + 3);
  }
  for (int i = 0; i < n_meas; i++) {
    BCHECK(cam[i] >= 0 && cam[i] < n_cams && point[i] >= 0 && point[i] < n_pts, "measurement refers to an unknown camera/point");
    h.mp.push_back(point[i]); h.mc.push_back(cam[i]); h.mfound.push_back(xy[2 * i]); h.mfound.push_back(xy[2 * i + 1]);
    h.msin.push_back(sqrt(1.0 / sigma_squared[i]));
  }
  return VSLAM_OK;
}

extern "C" int vslam_bundle_compute(vslam_bundle* b) {
  BCHECK(b, "null");
  const BaPool& P = b->pool;
  const size_t N = P.N, C = P.max_cams, PP = P.max_pts, M = P.max_meas;
  if (b->dirty) {
    // the problems as the caller built them, staged per array for all problems and uploaded with one copy per array
    std::vector<BaResult> res(N);
    std::vector<Pose> cams(N * C); std::vector<int> fixed(N * C, 0);
    std::vector<double> pts(N * PP * 3, 0.0), found(N * M * 2, 0.0), msin(N * M, 0.0);
    std::vector<int> mp(N * M, 0), mc(N * M, 0), lut(N * C * PP, -1), nmeas(N * PP, 0);
    for (size_t n = 0; n < N; n++) {
      const HostProblem& h = b->host[n];
      BaResult r; memset(&r, 0, sizeof(r));
      r.n_cams = (int)h.cams.size(); r.n_pts = (int)h.pts.size() / 3; r.n_meas = (int)h.mp.size();
      r.active = r.n_cams > 0 && r.n_pts > 0 && r.n_meas > 0;
      res[n] = r;
      for (int c = 0; c < r.n_cams; c++) { cams[n * C + c] = h.cams[c]; fixed[n * C + c] = h.fixed[c]; }
      for (int i = 0; i < 3 * r.n_pts; i++) pts[n * PP * 3 + i] = h.pts[i];
      for (int i = 0; i < r.n_meas; i++) {
        mp[n * M + i] = h.mp[i]; mc[n * M + i] = h.mc[i]; msin[n * M + i] = h.msin[i];
        found[n * M * 2 + i] = h.mfound[2 * i]; found[n * M * 2 + M + i] = h.mfound[2 * i + 1];      // component-major on device (ba_device.h MS())
        lut[n * C * PP + (size_t)h.mc[i] * PP + h.mp[i]] = i; nmeas[n * PP + h.mp[i]]++;              // GenerateMeasLUTs
      }
    }
    HIPCHK(hipMemcpyAsync(b->res0, res.data(), sizeof(BaResult) * N, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->cam0, cams.data(), sizeof(Pose) * N * C, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->pt0, pts.data(), sizeof(double) * N * PP * 3, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(P.cam_fixed, fixed.data(), sizeof(int) * N * C, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(P.ms_p, mp.data(), sizeof(int) * N * M, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(P.ms_c, mc.data(), sizeof(int) * N * M, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(P.ms_found, found.data(), sizeof(double) * N * M * 2, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(P.ms_sin, msin.data(), sizeof(double) * N * M, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(P.lut, lut.data(), sizeof(int) * N * C * PP, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(P.pt_nmeas, nmeas.data(), sizeof(int) * N * PP, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));           // the staging vectors go out of scope
    b->dirty = false;
  }
  // Compute leaves its result in the cameras and points: every call starts again from the caller's values (device-to-device)
  HIPCHK(hipMemcpyAsync(P.res, b->res0, sizeof(BaResult) * N, hipMemcpyDeviceToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(P.cam_pose, b->cam0, sizeof(Pose) * N * C, hipMemcpyDeviceToDevice, b->stream));
  HIPCHK(hipMemcpyAsync(P.pt_pos, b->pt0, sizeof(double) * N * PP * 3, hipMemcpyDeviceToDevice, b->stream));
  HIPCHK(hipMemsetAsync(P.ms_state, 0, sizeof(int) * N * M, b->stream));
  HIPCHK(hipMemsetAsync(P.pt_nout, 0, sizeof(int) * N * PP, b->stream));
  HIPCHK(hipMemsetAsync(P.lstat, 0, sizeof(unsigned long long) * BA_LSTAT_N, b->stream));
  b->uploaded = true;
  HIPCHK(hipEventRecord(b->ev[0], b->stream));         // everything Compute reads is resident
  ba_launch_compute(b->pool, b->cfg, P.N, b->stream, -1, 0);
  HIPCHK(hipEventRecord(b->ev[1], b->stream));
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

// HIP-event time of the last vslam_bundle_compute launch (the problems already resident) and what it ran: stats as vslam_profile_ba_stats
extern "C" int vslam_bundle_get_timing(vslam_bundle* b, double* ms, unsigned long long stats[8]) {
  BCHECK(b && ms, "get_timing: bad argument");
  if (!b->uploaded) { vslam_set_error("bundle: compute has not run"); return VSLAM_E_STATE; }
  HIPCHK(hipStreamSynchronize(b->stream));
  float t = 0.f;
  HIPCHK(hipEventElapsedTime(&t, b->ev[0], b->ev[1]));
  *ms = t;
  if (stats) { HIPCHK(hipMemcpy(stats, b->pool.lstat, sizeof(unsigned long long) * BA_LSTAT_N, hipMemcpyDeviceToHost)); stats[7] = 1; }
  return VSLAM_OK;
}

extern "C" int vslam_bundle_synchronize(vslam_bundle* b) { BCHECK(b, "null"); HIPCHK(hipStreamSynchronize(b->stream)); return VSLAM_OK; }

static int bundle_result(vslam_bundle* b, int n, BaResult* r) {
  BCHECK(b && n >= 0 && n < b->pool.N, "bad problem index");
  if (!b->uploaded) { vslam_set_error("bundle: compute has not run"); return VSLAM_E_STATE; }
  HIPCHK(hipMemcpyAsync(r, ba_view(b->pool, n).res, sizeof(BaResult), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return VSLAM_OK;
}

extern "C" int vslam_bundle_get_result(vslam_bundle* b, int n, int* accepted, int* converged, double* sigma_squared, double* lambda, long long* trials) {
  BaResult r; int rc = bundle_result(b, n, &r); if (rc) return rc;
  if (accepted) *accepted = r.accepted; if (converged) *converged = r.converged; if (sigma_squared) *sigma_squared = r.sigma2;
  if (lambda) *lambda = r.lambda; if (trials) *trials = r.trials;
  return VSLAM_OK;
}
extern "C" int vslam_bundle_get_camera(vslam_bundle* b, int n, int i, double pose12[12]) {
  BaResult r; int rc = bundle_result(b, n, &r); if (rc) return rc;
  BCHECK(i >= 0 && i < r.n_cams, "camera index");
  Pose p;
  HIPCHK(hipMemcpy(&p, ba_view(b->pool, n).cam_pose + i, sizeof(Pose), hipMemcpyDeviceToHost));
  for (int k = 0; k < 9; k++) pose12[k] = p.R[k]; for (int k = 0; k < 3; k++) pose12[9 + k] = p.t[k];
  return VSLAM_OK;
}
extern "C" int vslam_bundle_get_point(vslam_bundle* b, int n, int i, double pos[3]) {
  BaResult r; int rc = bundle_result(b, n, &r); if (rc) return rc;
  BCHECK(i >= 0 && i < r.n_pts, "point index");
  HIPCHK(hipMemcpy(pos, ba_view(b->pool, n).pt_pos + 3 * i, sizeof(double) * 3, hipMemcpyDeviceToHost));
  return VSLAM_OK;
}
extern "C" int vslam_bundle_get_outlier_meas(vslam_bundle* b, int n, int* pc, int cap) {
  BaResult r; int rc = bundle_result(b, n, &r); if (rc) return rc;
  const int m = r.n_outlier_meas < cap ? r.n_outlier_meas : cap;
  if (pc && m > 0) HIPCHK(hipMemcpy(pc, ba_view(b->pool, n).outl, sizeof(int) * 2 * m, hipMemcpyDeviceToHost));
  return r.n_outlier_meas;
}
extern "C" int vslam_bundle_get_outlier_points(vslam_bundle* b, int n, int* idx, int cap) {
  BaResult r; int rc = bundle_result(b, n, &r); if (rc) return rc;
  std::vector<int> nm(r.n_pts), no(r.n_pts);
  if (r.n_pts) {
    HIPCHK(hipMemcpy(nm.data(), ba_view(b->pool, n).pt_nmeas, sizeof(int) * r.n_pts, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(no.data(), ba_view(b->pool, n).pt_nout, sizeof(int) * r.n_pts, hipMemcpyDeviceToHost));
  }
  int cnt = 0;
  for (int i = 0; i < r.n_pts; i++) if (nm[i] > 0 && nm[i] == no[i]) { if (idx && cnt < cap) idx[cnt] = i; cnt++; }   // GetOutliers :628-637
  return cnt;
}

// =================================================================================================================
// in-system mapping: AddKeyFrame + BundleAdjust on the streams' maps
// =================================================================================================================
struct KfCopyArgs {
  const uint8_t* src[NLEV]; size_t src_sstride[NLEV]; int src_pitch[NLEV];
  int w[NLEV], h[NLEV], kf_pitch[NLEV]; size_t kf_stride[NLEV];
};

// MapMaker::AddKeyFrame (:470-478): deep copy of the tracker's current keyframe into the next map slot.
__global__ __launch_bounds__(256) void k_add_keyframe(MapDev m, TrackParams tp, KfCopyArgs a) {
  const int s = blockIdx.y;
  TrackerState* st = &m.st[s];
  if (!st->kf_pending) return;
  const int slot = st->n_kf;                         // n_kf is advanced by k_ba_assemble (next launch)
  const int nth = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int l = 0; l < NLEV; l++) {                   // Level::operator= copies pixels (jni/KeyFrame.cc:104-112)
    const uint8_t* src = a.src[l] + (size_t)s * a.src_sstride[l];
    uint8_t* dst = m.kf_img[l] + ((size_t)s * tp.max_keyframes + slot) * a.kf_stride[l];
    const int w4 = a.w[l] >> 2;
    if ((a.src_pitch[l] & 3) == 0 && (((uintptr_t)src) & 3) == 0) {
      for (int t = tid; t < a.h[l] * w4; t += nth) {
        const int y = t / w4, x = t - y * w4;
        ((uint32_t*)(dst + (size_t)y * a.kf_pitch[l]))[x] = ((const uint32_t*)(src + (size_t)y * a.src_pitch[l]))[x];
      }
      const int rem = a.w[l] - (w4 << 2);
      for (int t = tid; t < a.h[l] * rem; t += nth) { const int y = t / rem, x = (w4 << 2) + t % rem; dst[(size_t)y * a.kf_pitch[l] + x] = src[(size_t)y * a.src_pitch[l] + x]; }
    } else {
      for (int t = tid; t < a.h[l] * a.w[l]; t += nth) { const int y = t / a.w[l], x = t - y * a.w[l]; dst[(size_t)y * a.kf_pitch[l] + x] = src[(size_t)y * a.src_pitch[l] + x]; }
    }
  }
  MeasDev* km = m.kf_meas + ((size_t)s * tp.max_keyframes + slot) * tp.max_points;
  const MeasDev* cm = m.cur_meas + (size_t)s * tp.max_points;
  MapPointDev* pts = m.pts + (size_t)s * tp.max_points;
  for (int i = tid; i < st->n_points; i += nth) {    // mMeasurements copy + :491-494
    MeasDev mm = cm[i];
    if (mm.valid) { mm.source = 0; pts[i].n_meas_kfs++; }
    km[i] = mm;
  }
  if (tid == 0) {
    m.kf_pose[(size_t)s * tp.max_keyframes + slot] = st->pose_final;
    m.kf_fixed[(size_t)s * tp.max_keyframes + slot] = 0;
    m.kf_depth[((size_t)s * tp.max_keyframes + slot) * 2] = st->depth_mean;
    m.kf_depth[((size_t)s * tp.max_keyframes + slot) * 2 + 1] = st->depth_sigma;
  }
}

DEVFN double kfdist(const Pose& a, const Pose& b) {   // KeyFrameLinearDist :705-712
  const Pose ia = pose_inverse(a), ib = pose_inverse(b);
  const double d0 = ib.t[0] - ia.t[0], d1 = ib.t[1] - ia.t[1], d2 = ib.t[2] - ia.t[2];
  return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}

// mode 0: after AddKeyFrame (only streams with kf_pending) -> BundleAdjustRecent; 1: BundleAdjustRecent on every
// stream; 2: BundleAdjustAll on every stream.  Two kernels build the Bundle problem of BundleAdjust (:854-902) in the pool:
// k_ba_select (one lane per stream, on the tracker's stream) takes the decisions that the next frame's tracking must see --
// the keyframe joins the map (n_kf), which cameras are adjusted (:803-820), the countdown of the asynchronous map-maker --
// and k_ba_assemble does the long part (point set, fixed set, measurement list), which with the asynchronous map-maker runs
// on the map-maker's stream beside the following frames: nothing it reads changes while the stream's adjustment is pending.
__global__ __launch_bounds__(64) void k_ba_select(MapDev m, TrackParams tp, BaPool pool, int mode, int token) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= pool.N) return;
  TrackerState* st = &m.st[s];
  const BaView v = ba_view(pool, s);
  BaResult* R = v.res;
  const int K = tp.max_keyframes;
  const Pose* kfp = m.kf_pose + (size_t)s * K;
  const int* kff = m.kf_fixed + (size_t)s * K;
  int* view_of_kf = pool.view_of_kf + (size_t)s * BA_MAX_KF;
  int go = 0, nadj_out = 0;
  const bool in_flight = st->ba_countdown > 0;     // asynchronous map-maker: the pool still belongs to the last keyframe
  if (!in_flight) R->active = 0;
  if (mode == 0 && st->kf_pending) {
    st->n_kf++;                                                       // mMap.vpKeyFrames.push_back (:489)
    st->kf_added = 1;
    st->ba_converged_full = 0; st->ba_converged_recent = 0;           // :504-505
  }
  bool want = st->map_good && (mode != 0 || st->kf_pending) && !in_flight;
  if (mode == 5 || mode == 6) want = want && st->boot_run;                                  // InitFromStereo's BundleAdjustAll calls, jni/MapMaker.cc:344-345, 361-365
  if (mode == 6) want = want && !st->ba_converged_full;
  if (mode == 3) want = want && !st->ba_converged_recent;                                   // MapMaker::run :97-98
  if (mode == 4) want = want && st->ba_converged_recent && !st->ba_converged_full;          // :107-108
  const bool all = mode == 2 || mode == 4 || mode == 5 || mode == 6;
  if (want && mode == 3) st->n_ba_recent_idle++;
  if (want && mode == 4) st->n_ba_all++;
  const int nk = st->n_kf;
  if (want) for (int k = 0; k < BA_MAX_KF; k++) view_of_kf[k] = -1;
  if (want && !all) {
    if (nk < tp.ba_min_keyframes) { st->ba_converged_recent = 1; st->ba_accepted = -2; }   // :803-807
    else {
      // adjust set: newest + (window-1) nearest non-fixed (:812-820), camera ids in keyframe order
      const int newest = nk - 1;
      int N = tp.ba_window - 1; if (N > nk - 1) N = nk - 1;
      unsigned long long chosen[2] = {0ull, 0ull}, taken[2] = {0ull, 0ull};      // bit sets over up to 128 keyframes
      chosen[newest >> 6] |= 1ull << (newest & 63);
      for (int n = 0; n < N; n++) {                                   // partial_sort by (distance, index)
        int best = -1; double bd = 0;
        for (int k = 0; k < nk; k++) {
          if (k == newest || ((taken[k >> 6] >> (k & 63)) & 1ull)) continue;
          const double d = kfdist(kfp[newest], kfp[k]);
          if (best < 0 || d < bd) { best = k; bd = d; }
        }
        if (best < 0) break;
        taken[best >> 6] |= 1ull << (best & 63);
        if (!kff[best]) chosen[best >> 6] |= 1ull << (best & 63);
      }
      int c = 0;
      for (int k = 0; k < nk; k++) if ((chosen[k >> 6] >> (k & 63)) & 1ull) { view_of_kf[k] = c; v.cam_fixed[c] = kff[k]; v.cam_pose[c] = kfp[k]; pool.id_view[(size_t)s * pool.max_cams + c] = k; c++; }
      nadj_out = c; go = 1;
    }
  } else if (want && all) {                                           // BundleAdjustAll :776-798
    int c = 0;
    for (int k = 0; k < nk; k++) if (!kff[k]) { view_of_kf[k] = c; v.cam_fixed[c] = 0; v.cam_pose[c] = kfp[k]; pool.id_view[(size_t)s * pool.max_cams + c] = k; c++; }
    nadj_out = c; go = c > 0 && c <= 64;                            // the slot layout holds the adjustable cameras of a point in one 64-bit set
    if (c > 64) st->ba_accepted = -3;
  }
  // The pool record of a stream whose adjustment is pending belongs to that adjustment: its k_ba_assemble runs on a map-maker
  // stream and may start after this launch (of a LATER frame).  `go` carries the token of the ba_run call that wants the problem,
  // so an assemble launched for another call never picks it up.
  if (!in_flight) { R->go = go ? token : 0; R->nadj = nadj_out; }
  if (go && mode == 0 && tp.ba_delay > 0) st->ba_countdown = tp.ba_delay;   // results are applied ba_delay frames from now
}

__global__ __launch_bounds__(BA_THREADS) void k_ba_assemble(MapDev m, TrackParams tp, BaPool pool, int mode, int slot /* work list to enter, or -1 */, int token) {
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  const BaView v = ba_view(pool, s);
  BaResult* R = v.res;
  if (R->go != token) return;
  __shared__ int sh_nc, sh_nm;
  __shared__ int view_of_kf[BA_MAX_KF];      // kf index -> camera id or -1
  __shared__ int ired[BA_WAVES];
  __shared__ int kf_cnt[BA_MAX_KF], kf_off[BA_MAX_KF];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int P = tp.max_points, K = tp.max_keyframes;
  const Pose* kfp = m.kf_pose + (size_t)s * K;
  const int* kff = m.kf_fixed + (size_t)s * K;
  const MeasDev* kfm = m.kf_meas + (size_t)s * K * P;
  if (threadIdx.x < BA_MAX_KF) view_of_kf[threadIdx.x] = pool.view_of_kf[(size_t)s * BA_MAX_KF + threadIdx.x];
  const int sh_nadj = R->nadj;
  __syncthreads();
  const int nk = st->n_kf, npts = st->n_points, nadj = sh_nadj;
  MapPointDev* pts = m.pts + (size_t)s * P;
  // ---- point set (:823-831 / :791-795), ids in map order; id_point translation ----
  int* idp = pool.id_point + (size_t)s * pool.max_pts;
  int base = 0;
  for (int i0 = 0; i0 < npts; i0 += BA_THREADS) {
    const int i = i0 + threadIdx.x;
    bool in = false;
    if (i < npts) {
      if (mode == 2) in = !pts[i].bad;
      else for (int k = 0; k < nk && !in; k++) if (view_of_kf[k] >= 0 && kfm[(size_t)k * P + i].valid) in = true;
    }
    const unsigned long long bm = __ballot(in);
    __syncthreads();
    if (lane == 0) ired[wave] = __popcll(bm);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; w++) off += ired[w];
    if (i < npts) {
      int id = -1;
      if (in) { id = off + __popcll(bm & ((1ull << lane) - 1ull)); if (id < pool.max_pts) { idp[id] = i; for (int q = 0; q < 3; q++) v.pt_pos[3 * id + q] = pts[i].pos[q]; v.pt_nmeas[id] = 0; v.pt_nout[id] = 0; } }
      ((int*)v.scratch)[i] = id;                    // scratch viewed as int[npts]: map point -> bundle point id
    }
    for (int w = 0; w < BA_WAVES; w++) base += ired[w];
  }
  __syncthreads();
  const int np = base;
  const int* pid_of = (const int*)v.scratch;
  // ---- fixed set (:834-848): other keyframes measuring any point of the set; appended in keyframe order ----
  if (mode != 2) {
    for (int k = 0; k < nk; k++) {
      if (view_of_kf[k] >= 0) continue;
      int any = 0;
      for (int i = threadIdx.x; i < npts && !any; i += BA_THREADS) if (pid_of[i] >= 0 && kfm[(size_t)k * P + i].valid) any = 1;
      any = __syncthreads_or(any);
      if (any && threadIdx.x == 0) { kf_cnt[k] = 1; } else if (threadIdx.x == 0) kf_cnt[k] = 0;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = nadj;
    for (int k = 0; k < nk; k++) {
      const bool fixed_member = mode == 2 ? (kff[k] != 0) : (view_of_kf[k] < 0 && kf_cnt[k]);
      if (fixed_member && c < pool.max_cams) { view_of_kf[k] = c; v.cam_fixed[c] = 1; v.cam_pose[c] = kfp[k]; pool.id_view[(size_t)s * pool.max_cams + c] = k; c++; }
    }
    sh_nc = c;
  }
  __syncthreads();
  const int ncam = sh_nc;
  for (int t = threadIdx.x; t < ncam * pool.max_pts; t += BA_THREADS) v.lut[(size_t)(t / pool.max_pts) * pool.max_pts + t % pool.max_pts] = -1;
  // ---- measurements in map keyframe order, points ascending (:888-902) ----
  for (int k = 0; k < nk; k++) {                                         // counts per keyframe
    int c = 0;
    if (view_of_kf[k] >= 0) for (int i = threadIdx.x; i < npts; i += BA_THREADS) if (pid_of[i] >= 0 && kfm[(size_t)k * P + i].valid) c++;
    c = ba_block_sum_i(c, ired);
    if (threadIdx.x == 0) kf_cnt[k] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) { int o = 0; for (int k = 0; k < nk; k++) { kf_off[k] = o; o += kf_cnt[k]; } sh_nm = o; }
  __syncthreads();
  const int nm = sh_nm;
  if (np > pool.max_pts || nm > pool.max_meas || ncam > pool.max_cams) { if (threadIdx.x == 0) { R->active = 0; R->go = 0; st->ba_accepted = -3; } return; }
  for (int k = 0; k < nk; k++) {
    if (view_of_kf[k] < 0) continue;
    const int cam = view_of_kf[k];
    int kb = kf_off[k];
    for (int i0 = 0; i0 < npts; i0 += BA_THREADS) {
      const int i = i0 + threadIdx.x;
      const bool in = i < npts && pid_of[i] >= 0 && kfm[(size_t)k * P + i].valid;
      const unsigned long long bm = __ballot(in);
      __syncthreads();
      if (lane == 0) ired[wave] = __popcll(bm);
      __syncthreads();
      int off = kb;
      for (int w = 0; w < wave; w++) off += ired[w];
      if (in) {
        off += __popcll(bm & ((1ull << lane) - 1ull));
        const MeasDev mm = kfm[(size_t)k * P + i];
        const int pid = pid_of[i];
        v.ms_p[off] = pid; v.ms_c[off] = cam; v.ms_state[off] = MS_OK;
        MS(ms_found, 0, off) = mm.root[0]; MS(ms_found, 1, off) = mm.root[1];
        const int sc = 1 << mm.level;
        v.ms_sin[off] = sqrt(1.0 / (double)(sc * sc));                   // :899 + AddMeas :115
        v.lut[(size_t)cam * pool.max_pts + pid] = off;
        atomicAdd(&v.pt_nmeas[pid], 1);
      }
      for (int w = 0; w < BA_WAVES; w++) kb += ired[w];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    R->n_cams = ncam; R->n_pts = np; R->n_meas = nm; R->active = (ncam > 0 && np > 0 && nm > 0); R->accepted = 0; R->n_outlier_meas = 0; R->computed = 0;
    R->go = 0;
    if (slot >= 0 && R->active) pool.work[(size_t)slot * pool.N + atomicAdd(&pool.work_n[slot], 1)] = s;   // list order is immaterial: the problems are independent
  }
}

// BundleAdjust tail (:904-959) + HandleBadPoints (:140-164)
__global__ __launch_bounds__(BA_THREADS) void k_ba_writeback(MapDev m, TrackParams tp, BaPool pool, int mode) {
  const int s = blockIdx.x;
  TrackerState* st = &m.st[s];
  const BaView v = ba_view(pool, s);
  BaResult* R = v.res;
  const int P = tp.max_points, K = tp.max_keyframes;
  MapPointDev* pts = m.pts + (size_t)s * P;
  MeasDev* kfm = m.kf_meas + (size_t)s * K * P;
  __shared__ int sh_due;
  if (mode == 5) { if (!st->boot_run) return; mode = 2; }   // InitFromStereo's adjustments: only the streams it runs for
  if (mode >= 3) {                         // asynchronous map-maker: is this stream's pending result due?
    if (threadIdx.x == 0) {
      int due = 0;
      if (st->ba_countdown > 0) { if (mode == 4) st->ba_countdown = 1; if (--st->ba_countdown == 0) { due = 1; st->ba_countdown = -1; } }
      sh_due = due;
    }
    __syncthreads();
    if (!sh_due) return;
  }
  const bool deferred = mode == 0 && tp.ba_delay > 0;   // the results of this keyframe's BA are written back ba_delay frames later
  if (R->active && !deferred) {
    const int* idp = pool.id_point + (size_t)s * pool.max_pts;
    const int* idv = pool.id_view + (size_t)s * pool.max_cams;
    const int acc = R->accepted;
    if (acc > 0) {                                                       // :918-929
      for (int i = threadIdx.x; i < R->n_pts; i += BA_THREADS) for (int q = 0; q < 3; q++) pts[idp[i]].pos[q] = v.pt_pos[3 * i + q];
      for (int c = threadIdx.x; c < R->n_cams; c += BA_THREADS) m.kf_pose[(size_t)s * K + idv[c]] = v.cam_pose[c];
    }
    __syncthreads();
    // Outlier measurements (:941-959).  The reference walks the list in order; entries of different points do not interact (a
    // point's bBad flag and its sMeasurementKFs size are touched by its own entries only), so the first entry of every point
    // takes that point's entries in list order and the points proceed in parallel (a bundle adjustment reports hundreds of
    // outliers: one lane walking them through dependent global accesses was most of this kernel's time).
    if (acc >= 0) {
      const int no = R->n_outlier_meas;
      __shared__ int sh_op[2048];                                        // the outliers' point ids (the scans below read them n^2 / 2 times)
      const bool in_lds = no <= 2048;
      if (in_lds) for (int o = threadIdx.x; o < no; o += BA_THREADS) sh_op[o] = v.outl[2 * o];
      __syncthreads();
      auto op = [&](int q) -> int { return in_lds ? sh_op[q] : v.outl[2 * q]; };
      for (int o = threadIdx.x; o < no; o += BA_THREADS) {
        const int bp = op(o);
        bool first = true;
        for (int q = 0; q < o && first; q++) if (op(q) == bp) first = false;
        if (!first) continue;
        const int pp = idp[bp];
        int nkfs = pts[pp].n_meas_kfs; int bad = pts[pp].bad;
        for (int q = o; q < no; q++) {
          if (op(q) != bp) continue;
          const int pk = idv[v.outl[2 * q + 1]];
          MeasDev& mm = kfm[(size_t)pk * P + pp];
          if (nkfs <= 2 || mm.source == 2 /* SRC_ROOT */) bad = 1;
          else {
            if (tp.idle > 0) {                                             // :951-956: a second chance later, or never again
              if (mm.source == 0 /* SRC_TRACKER */ || mm.source == 4 /* SRC_EPIPOLAR */) {
                const int e = atomicAdd(&st->fq_n, 1);                     // (the queue is sorted / order-free when it is read)
                if (e < tp.fq_cap) m.fq[(size_t)s * tp.fq_cap + e] = make_int2(pk, pp);
              } else atomicOr(&m.never_retry[((size_t)s * P + pp) * 2 + (pk >> 6)], 1ull << (pk & 63));
            }
            mm.valid = 0; nkfs--;
          }
        }
        pts[pp].n_meas_kfs = nkfs; pts[pp].bad = bad;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      st->ba_accepted = acc;
      st->n_ba_trials += (unsigned long long)R->trials;
      if (acc >= 0) {
        if (acc > 0) { if (mode != 2) st->ba_converged_recent = 0; st->ba_converged_full = 0; }
        if (R->converged) { st->ba_converged_recent = 1; if (mode == 2) st->ba_converged_full = 1; }   // :931-935
      }
      R->active = 0;
    }
    __syncthreads();
  }
  if (!(st->map_good && (mode != 0 || st->kf_pending))) return;
  if (mode == 0 && tp.ba_delay > 0 && st->ba_countdown > 0) return;   // deferred: HandleBadPoints runs with the delayed write-back
  // HandleBadPoints :140-164
  const int nk = st->n_kf;
  for (int i = threadIdx.x; i < st->n_points; i += BA_THREADS) {
    if (pts[i].n_out > 20 && pts[i].n_out > pts[i].n_in) pts[i].bad = 1;
    if (pts[i].bad) for (int k = 0; k < nk; k++) kfm[(size_t)k * P + i].valid = 0;
  }
}

// HandleBadPoints (:140-164) on its own: MapMaker::run :117 calls it after every pass through the idle jobs
__global__ __launch_bounds__(BA_THREADS) void k_handle_bad_points(MapDev m, TrackParams tp) {
  const int s = blockIdx.x;
  const TrackerState* st = &m.st[s];
  if (!st->map_good) return;
  const int P = tp.max_points, K = tp.max_keyframes, nk = st->n_kf;
  MapPointDev* pts = m.pts + (size_t)s * P;
  MeasDev* kfm = m.kf_meas + (size_t)s * K * P;
  for (int i = threadIdx.x; i < st->n_points; i += BA_THREADS) {
    if (pts[i].n_out > 20 && pts[i].n_out > pts[i].n_in) pts[i].bad = 1;
    if (pts[i].bad) for (int k = 0; k < nk; k++) kfm[(size_t)k * P + i].valid = 0;
  }
}

// ---- host ---------------------------------------------------------------------------------------------------------
struct BaSystemWs { BaPool pool; };

int ba_alloc(vslam_system* sys) {
  BaSystemWs* ws = new BaSystemWs();
  sys->ba_ws = ws;
  const int K = sys->p.max_keyframes, P = sys->p.max_points;
  if (K > BA_MAX_KF) { vslam_set_error("max_keyframes %d exceeds %d", K, BA_MAX_KF); return VSLAM_E_INVALID; }
  // worst case of BundleAdjust: every keyframe a camera, every point, every (kf, point) slot a measurement
  size_t M = (size_t)K * P;
  if (M > 65536) M = 65536;
  int r = pool_create(ws->pool, sys->allocs, sys->stream, sys->S, K, P, (int)M, sys->p.ba_delay_frames > 0 ? sys->p.ba_delay_frames + 2 : 1, sys->p.ba_sum_order != 0, BA_LSTAT_RING);
  if (r) return r;
  // One launch of the full synchronous grid over the still empty pool (every problem inactive: the workgroups return at once).
  // k_ba_compute needs scratch memory, and the runtime sizes that lazily, at the first launch of a grid this large: paid here,
  // at creation, not by the first BundleAdjustRecent / BundleAdjustAll a caller times.
  ba_launch_compute(ws->pool, make_cfg(sys->tp), sys->S, sys->stream, -1, -1);
  // ... and on every stream of the asynchronous map-maker's ring: scratch belongs to the hardware queue a stream maps to, and the
  // first batch on a queue that has not seen the kernel stalled the host for ~5 ms in the middle of the timed frames.
  for (hipStream_t st : sys->ba_streams) ba_launch_compute(ws->pool, make_cfg(sys->tp), sys->S, st, -1, -1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  return VSLAM_OK;
}

// the next record of the ring, zeroed on the stream that launches k_ba_compute
static int ba_next_launch_record(vslam_system* sys, hipStream_t st, int* rec, int* ordinal) {
  BaSystemWs* ws = (BaSystemWs*)sys->ba_ws;
  *ordinal = (int)(sys->ba_launch_no & 0x3fffffff);
  *rec = (int)(sys->ba_launch_no++ % BA_LSTAT_RING);
  HIPCHK(hipMemsetAsync(ws->pool.lstat + (size_t)*rec * BA_LSTAT_N, 0, sizeof(unsigned long long) * BA_LSTAT_N, st));
  return VSLAM_OK;
}

static void fill_kfcopy(vslam_system* sys, KfCopyArgs& a) {
  for (int l = 0; l < NLEV; l++) {
    a.src[l] = sys->fr.img[l]; a.src_sstride[l] = sys->fr.img_sstride[l]; a.src_pitch[l] = sys->fr.img_pitch[l];
    a.w[l] = sys->geom[l].w; a.h[l] = sys->geom[l].h; a.kf_pitch[l] = sys->geom[l].pitch;
    a.kf_stride[l] = (size_t)sys->geom[l].pitch * sys->geom[l].h;
  }
}

// Asynchronous map-maker: launch Bundle::Compute for the open batch -- every problem the last ba_batch_fill frames assembled -- on
// the next map-maker stream of the ring, behind the main stream's last k_ba_assemble.
static int ba_launch_batch(vslam_system* sys) {
  if (sys->ba_batch_fill == 0) return VSLAM_OK;
  BaSystemWs* ws = (BaSystemWs*)sys->ba_ws;
  const BaConfig cfg = make_cfg(sys->tp);
  const int R = (int)sys->ev_ba.size(), slot = (int)(sys->ba_batch_id % R);
  sys->ba_stream = sys->ba_streams[(size_t)(sys->ba_batch_id % (long)sys->ba_streams.size())];
  int lrec = -1, lord = 0;
  { int rr = ba_next_launch_record(sys, sys->ba_stream, &lrec, &lord); if (rr) return rr; }
  prof_mark(sys, 12);                                  // the batch's assemblies precede on this very stream
  // one workgroup per problem up to two per compute unit; the grid walks the batch's work list
  static const int per_cu_x2 = getenv("VSLAM_BA_WG_PER_CU_X2") ? atoi(getenv("VSLAM_BA_WG_PER_CU_X2")) : 4;   // diagnostic: background workgroups per CU, in halves
  const int cap = per_cu_x2 * (sys->n_cu > 0 ? sys->n_cu : 256) / 2;
  const int ba_grid = sys->S < cap ? sys->S : cap;
  ba_launch_compute(ws->pool, cfg, ba_grid, sys->ba_stream, slot, lrec);
  prof_mark(sys, PROF_BA_END);
  if (sys->prof_on && sys->prof_frame < sys->prof_cap && sys->prof_frame < (int)sys->prof_ba_launched.size()) sys->prof_ba_launched[sys->prof_frame] = lord + 1;
  HIPCHK(hipEventRecord(sys->ev_ba[slot], sys->ba_stream));
  HIPCHK(hipGetLastError());
  sys->ba_batch_id++;
  sys->ba_batch_fill = 0;
  return VSLAM_OK;
}

// Asynchronous map-maker (ba_delay_frames = D > 0): make the main stream wait for the bundle adjustment of the keyframes of
// D frames ago and apply whatever is due (k_ba_writeback decides per stream with its countdown).
int ba_frame_start(vslam_system* sys) {
  const int D = sys->tp.ba_delay;
  if (D <= 0) return VSLAM_OK;
  BaSystemWs* ws = (BaSystemWs*)sys->ba_ws;
  const int R = (int)sys->ev_ba.size(), FB = (int)sys->frame_batch.size();
  prof_mark(sys, 13);
  if (sys->frame_no >= D) {
    const long b = sys->frame_batch[(size_t)((sys->frame_no - D) % FB)];
    if (b >= 0) {
      if (b == sys->ba_batch_id) { int r = ba_launch_batch(sys); if (r) return r; }   // still open (only after a host-driven flush pattern): launch it now
      HIPCHK(hipStreamWaitEvent(sys->stream, sys->ev_ba[(size_t)(b % R)], 0));
    }
  }
  hipLaunchKernelGGL(k_ba_writeback, dim3(sys->S), dim3(BA_THREADS), 0, sys->stream, sys->map, sys->tp, ws->pool, 3);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

int ba_sync_streams(vslam_system* sys) {
  if (sys->tp.ba_delay > 0) { int r = ba_launch_batch(sys); if (r) return r; }
  for (hipStream_t st : sys->ba_streams) HIPCHK(hipStreamSynchronize(st));
  return VSLAM_OK;
}

// explicit (host-driven) map-maker calls first collect a bundle adjustment that is still in flight
static int ba_drain(vslam_system* sys) {
  if (sys->tp.ba_delay <= 0) return VSLAM_OK;
  BaSystemWs* ws = (BaSystemWs*)sys->ba_ws;
  { int rs = ba_sync_streams(sys); if (rs) return rs; }
  hipLaunchKernelGGL(k_ba_writeback, dim3(sys->S), dim3(BA_THREADS), 0, sys->stream, sys->map, sys->tp, ws->pool, 4);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

// mode 0: tracker-driven AddKeyFrame + BundleAdjustRecent; 1: BundleAdjustRecent; 2: BundleAdjustAll; 3 / 4: the same two as idle jobs of
// MapMaker::run, for the streams whose adjustment has not converged (gated on device)
int ba_run(vslam_system* sys, int mode, bool host_driven_keyframe) {
  const int base = mode == 3 ? 1 : (mode == 4 || mode >= 5 ? 2 : mode);   // what the assembly does; 5 / 6: InitFromStereo's BundleAdjustAll (boot.hip)
  const int wb = mode >= 5 ? 5 : base;                                    // ... and the write-back
  BaSystemWs* ws = (BaSystemWs*)sys->ba_ws;
  const BaConfig cfg = make_cfg(sys->tp);
  const bool async = mode == 0 && sys->tp.ba_delay > 0 && !host_driven_keyframe;
  const int token = (int)(++sys->ba_token & 0x3fffffff) + 1;   // names this call's problems (k_ba_select -> k_ba_assemble)
  if (mode == 1 || mode == 2) { int r = ba_drain(sys); if (r) return r; }
  if (mode == 0) {
    KfCopyArgs a; fill_kfcopy(sys, a);
    prof_mark(sys, 10);
    hipLaunchKernelGGL(k_add_keyframe, dim3(32, sys->S), dim3(256), 0, sys->stream, sys->map, sys->tp, a);
    int rg = grow_on_keyframe(sys);                      // AddSomeMapPoints (vslam_params.grow_map), before the bundle adjustment sees the keyframe
    if (rg) return rg;
  }
  if (mode == 0) prof_mark(sys, 11);
  if (async) {
    // Bundle::Compute on a map-maker stream, beside the next frames; its write-back is launched by ba_frame_start D frames later.
    // The keyframe frames of independent sequences do not coincide, so a frame brings only a few problems, and an adjustment
    // outlasts a frame (one persistent workgroup per problem, latency-bound): the problems of ba_batch consecutive frames are
    // collected in one work list and launched together; successive launches go to a ring of streams and may overlap.
    // A launch walks exactly its batch's list -- never a problem a later frame's k_ba_assemble is writing beside it.
    const int R = (int)sys->ev_ba.size(), slot = (int)(sys->ba_batch_id % R);
    hipStream_t bs = sys->ba_streams[(size_t)(sys->ba_batch_id % (long)sys->ba_streams.size())];
    hipLaunchKernelGGL(k_ba_select, dim3((sys->S + 63) / 64), dim3(64), 0, sys->stream, sys->map, sys->tp, ws->pool, mode, token);
    // the long part of the assembly leaves the tracker's stream: behind this frame's keyframe copy / map growth, on the batch's
    // map-maker stream (every frame of a batch uses the same stream, so the batch's launch follows all its assemblies)
    const int FBn = (int)sys->frame_batch.size(), es = (int)(sys->frame_no % FBn);
    HIPCHK(hipEventRecord(sys->ev_asm[es], sys->stream));
    HIPCHK(hipStreamWaitEvent(bs, sys->ev_asm[es], 0));
    if (sys->ba_batch_fill == 0)
      HIPCHK(hipMemsetAsync(ws->pool.work_n + slot, 0, sizeof(int), bs));   // the launch that read this slot R batches ago was waited for (ba_frame_start)
    if (sys->ba_batch_fill == 0) HIPCHK(hipMemsetAsync(ws->pool.work_n + ws->pool.work_slots + slot, 0, sizeof(int), bs));
    hipLaunchKernelGGL(k_ba_assemble, dim3(sys->S), dim3(BA_THREADS), 0, bs, sys->map, sys->tp, ws->pool, mode, slot, token);
    sys->frame_batch[(size_t)(sys->frame_no % (long)sys->frame_batch.size())] = sys->ba_batch_id;
    sys->ba_batch_fill++;
    if (sys->ba_batch_fill >= sys->tp.ba_batch) { int rl = ba_launch_batch(sys); if (rl) return rl; }
    hipLaunchKernelGGL(k_ba_writeback, dim3(sys->S), dim3(BA_THREADS), 0, sys->stream, sys->map, sys->tp, ws->pool, 0);   // HandleBadPoints of streams without a pending BA
    HIPCHK(hipGetLastError());
    return VSLAM_OK;
  }
  // host-driven calls (BundleAdjustRecent / BundleAdjustAll / AddKeyFrame on request): HIP events around the three parts, read by
  // vslam_get_mapmaker_timing.  A host-driven AddKeyFrame is adjusted here and now even when the tracker-driven ones run on the
  // map-maker streams (the pending ones were collected by the caller): the kernels see ba_delay = 0.
  const bool timed = (mode == 1 || mode == 2 || host_driven_keyframe) && sys->ev_mm[0];
  TrackParams tps = sys->tp;
  if (host_driven_keyframe) tps.ba_delay = 0;
  int lrec = -1, lord = 0;
  { int rr = ba_next_launch_record(sys, sys->stream, &lrec, &lord); if (rr) return rr; }
  if (timed) HIPCHK(hipEventRecord(sys->ev_mm[0], sys->stream));
  hipLaunchKernelGGL(k_ba_select, dim3((sys->S + 63) / 64), dim3(64), 0, sys->stream, sys->map, tps, ws->pool, mode, token);
  hipLaunchKernelGGL(k_ba_assemble, dim3(sys->S), dim3(BA_THREADS), 0, sys->stream, sys->map, tps, ws->pool, base, -1, token);
  if (mode == 0) prof_mark(sys, 12);
  if (timed) HIPCHK(hipEventRecord(sys->ev_mm[1], sys->stream));
  ba_launch_compute(ws->pool, cfg, sys->S, sys->stream, -1, lrec);
  if (mode == 0) prof_mark(sys, 13);
  if (timed) HIPCHK(hipEventRecord(sys->ev_mm[2], sys->stream));
  hipLaunchKernelGGL(k_ba_writeback, dim3(sys->S), dim3(BA_THREADS), 0, sys->stream, sys->map, tps, ws->pool, wb);
  if (timed) { HIPCHK(hipEventRecord(sys->ev_mm[3], sys->stream)); sys->mm_lrec = lrec; }
  if (mode == 0 && sys->prof_on && sys->prof_frame < sys->prof_cap && sys->prof_frame < (int)sys->prof_ba_launched.size()) sys->prof_ba_launched[sys->prof_frame] = lord + 1;
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

int ba_launch_add_keyframe(vslam_system* sys) {
  KfCopyArgs a; fill_kfcopy(sys, a);
  hipLaunchKernelGGL(k_add_keyframe, dim3(32, sys->S), dim3(256), 0, sys->stream, sys->map, sys->tp, a);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}

int ba_add_keyframe_and_adjust(vslam_system* sys) { return ba_run(sys, 0, false); }

// vslam_params.idle_iterations passes through the idle jobs of MapMaker::run (jni/MapMaker.cc:94-117), every stream deciding on
// device which of them it runs: BundleAdjustRecent until converged, ReFindNewlyMade, BundleAdjustAll until converged, every 20th
// time ReFindFromFailureQueue, HandleBadPoints (the tail of the write-back kernels).  Synchronous map-maker only.
int mm_idle_job(vslam_system* sys, int job) {
  if (job < 0 || job > 3) { vslam_set_error("mapmaker_idle_job: job must be 0..3"); return VSLAM_E_INVALID; }
  if (job == 0) return ba_run(sys, 3, false);                 // the write-back ends with HandleBadPoints
  if (job == 2) return ba_run(sys, 4, false);
  const int r = grow_idle_refind(sys, job == 1 ? 0 : 1);
  if (r) return r;
  hipLaunchKernelGGL(k_handle_bad_points, dim3(sys->S), dim3(BA_THREADS), 0, sys->stream, sys->map, sys->tp);   // :117
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}
int mm_idle(vslam_system* sys) {
  for (int it = 0; it < sys->p.idle_iterations; it++)
    for (int job = 0; job < 4; job++) { const int r = mm_idle_job(sys, job); if (r) return r; }
  return VSLAM_OK;
}

extern "C" int vslam_get_bundle_stats(vslam_system* sys, int s, int out[6]) {
  if (!sys || !out || s < 0 || s >= sys->S || !sys->ba_ws) { vslam_set_error("get_bundle_stats: bad argument"); return VSLAM_E_INVALID; }
  BaSystemWs* ws = (BaSystemWs*)sys->ba_ws;
  HIPCHK(hipStreamSynchronize(sys->stream));
  { int rs = ba_sync_streams(sys); if (rs) return rs; }
  BaResult r;
  HIPCHK(hipMemcpy(&r, ws->pool.res + s, sizeof(r), hipMemcpyDeviceToHost));
  out[0] = r.n_cams; out[1] = r.n_free; out[2] = r.n_pts; out[3] = r.n_meas; out[4] = r.counter; out[5] = r.accepted;
  return VSLAM_OK;
}

// ---- measurement ----------------------------------------------------------------------------------------------------
static int ba_read_launch_record(vslam_system* sys, int rec, unsigned long long out[BA_LSTAT_N]) {
  BaSystemWs* ws = (BaSystemWs*)sys->ba_ws;
  HIPCHK(hipMemcpy(out, ws->pool.lstat + (size_t)rec * BA_LSTAT_N, sizeof(unsigned long long) * BA_LSTAT_N, hipMemcpyDeviceToHost));
  return VSLAM_OK;
}

extern "C" int vslam_profile_ba_stats(vslam_system* sys, unsigned long long stats[8]) {
  if (!sys || !stats || !sys->ba_ws) { vslam_set_error("profile_ba_stats: bad argument"); return VSLAM_E_INVALID; }
  if (sys->prof_on) { vslam_set_error("profile_ba_stats: call vslam_profile_end first"); return VSLAM_E_STATE; }
  for (int k = 0; k < BA_LSTAT_N; k++) stats[k] = 0;
  int launches = 0;
  for (int f = 0; f < sys->prof_frame && f < (int)sys->prof_ba_launched.size(); f++) {
    if (!sys->prof_ba_launched[f]) continue;
    const long ord = sys->prof_ba_launched[f] - 1;
    if ((sys->ba_launch_no & 0x3fffffff) - ord > BA_LSTAT_RING) { vslam_set_error("profile_ba_stats: the window's launch records have been overwritten (more than %d launches since)", BA_LSTAT_RING); return VSLAM_E_CAPACITY; }
    unsigned long long r[BA_LSTAT_N];
    int rc = ba_read_launch_record(sys, (int)(ord % BA_LSTAT_RING), r); if (rc) return rc;
    for (int k = 0; k < BA_LSTAT_N; k++) stats[k] += r[k];
    launches++;
  }
  stats[7] = (unsigned long long)launches;
  return VSLAM_OK;
}

// every k_ba_compute launch of the system since its creation (the last BA_LSTAT_RING of them), summed: for profiler passes, whose
// per-kernel counters cover the whole process and not a window
extern "C" int vslam_get_ba_launch_totals(vslam_system* sys, unsigned long long stats[8]) {
  if (!sys || !stats || !sys->ba_ws) { vslam_set_error("get_ba_launch_totals: bad argument"); return VSLAM_E_INVALID; }
  HIPCHK(hipStreamSynchronize(sys->stream));
  { int rs = ba_sync_streams(sys); if (rs) return rs; }
  BaSystemWs* ws = (BaSystemWs*)sys->ba_ws;
  const long n = sys->ba_launch_no < BA_LSTAT_RING ? sys->ba_launch_no : BA_LSTAT_RING;
  std::vector<unsigned long long> all((size_t)BA_LSTAT_RING * BA_LSTAT_N);
  HIPCHK(hipMemcpy(all.data(), ws->pool.lstat, sizeof(unsigned long long) * all.size(), hipMemcpyDeviceToHost));
  for (int k = 0; k < BA_LSTAT_N; k++) stats[k] = 0;
  long working = 0;
  for (long i = 0; i < n; i++) { for (int k = 0; k < 7; k++) stats[k] += all[(size_t)i * BA_LSTAT_N + k]; if (all[(size_t)i * BA_LSTAT_N]) working++; }
  stats[7] = (unsigned long long)working;               // launches that ran at least one problem
  return VSLAM_OK;
}

extern "C" int vslam_get_mapmaker_timing(vslam_system* sys, double ms[3], unsigned long long stats[8]) {
  if (!sys || !ms || !sys->ba_ws) { vslam_set_error("get_mapmaker_timing: bad argument"); return VSLAM_E_INVALID; }
  if (sys->mm_lrec < 0 || !sys->ev_mm[0]) { vslam_set_error("get_mapmaker_timing: no vslam_bundle_adjust_recent / _all call yet"); return VSLAM_E_STATE; }
  HIPCHK(hipStreamSynchronize(sys->stream));
  for (int k = 0; k < 3; k++) { float t = 0.f; HIPCHK(hipEventElapsedTime(&t, sys->ev_mm[k], sys->ev_mm[k + 1])); ms[k] = t; }
  if (stats) { int rc = ba_read_launch_record(sys, sys->mm_lrec, stats); if (rc) return rc; stats[7] = 1; }
  return VSLAM_OK;
}

#ifdef VSLAM_BA_PROF
extern "C" int vslam_debug_ba_prof(unsigned long long* out32, int reset) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_ba_prof), sizeof(unsigned long long) * 32));
  if (reset) { unsigned long long z[32] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_ba_prof), z, sizeof(z))); }
  return VSLAM_OK;
}
#endif

__global__ void k_request_keyframe(MapDev m, TrackParams tp, int S, int stream) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  TrackerState* st = &m.st[s];
  const bool want = (stream < 0 || s == stream) && st->map_good && st->n_kf < tp.max_keyframes;
  st->kf_pending = want ? 1 : 0;                      // a request left over from the last frame has been served already
  if (want) st->last_kf_dropped = st->frame;          // Tracker::AddNewKeyFrame, jni/Tracker.cc:823-827
}

extern "C" int vslam_add_keyframe(vslam_system* sys, int stream) {
  if (!sys || stream >= sys->S) { vslam_set_error("add_keyframe: bad argument"); return VSLAM_E_INVALID; }
  if (!sys->have_frame) { vslam_set_error("add_keyframe: no current frame"); return VSLAM_E_STATE; }
  { int r = ba_drain(sys); if (r) return r; }        // asynchronous map-maker: collect the adjustments in flight first
  hipLaunchKernelGGL(k_request_keyframe, dim3((sys->S + 63) / 64), dim3(64), 0, sys->stream, sys->map, sys->tp, sys->S, stream);
  return ba_run(sys, 0, true);
}
