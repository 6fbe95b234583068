// Bundle::Compute (jni/Bundle.cc:136-178) / Do_LM_Step (:202-532) as ONE persistent workgroup per problem:
// the whole Levenberg-Marquardt loop runs inside one launch, no host round-trip.
//
// The layout is built around the bytes an LM trial has to move (fp64, struct-of-arrays, component-major) and around the dependent
// loads a persistent workgroup of two wavefronts per SIMD cannot hide:
//   cameras: pose, trial pose, fixed flag, start row, U (6x6 lower), epsilon_a     (poses and camera updates are read from LDS)
//   points : position, trial position, V (3x3 lower), epsilon_b                      (V*^-1 is recomputed where it is used)
//   measurements: the caller's list (AddMeas order = the reference's std::list order: ms_p, ms_c, ms_found, ms_sin, lut[c][p])
//     is re-laid once per Compute into SLOTS: region F = the measurements in adjustable cameras, region X = those in fixed
//     cameras, each point-major (points ascending, cameras ascending within a point), cut into chunks of <= 64 slots of whole
//     points.  A point's F slots are [pt_offF[p], pt_offF[p+1]) with pt_maskF[p] = set of adjustable-camera ordinals present, so
//     "the measurement of point p in adjustable camera f" is an offset + a popcount instead of the lut's dependent 4-byte
//     gather, and every sweep reads contiguous slots.  Per slot: static (camera | state | ordinal | has-F-slots, point, found
//     position, sqrt-inv-noise = 32 B, logical index for the outlier order); rewritten once per LM step: the state, and for an
//     F slot the four weighted camera derivatives (32 B) -- an X slot is consumed inside the sweep and stores nothing else.
//   The Jacobians A (2x6), B (2x3) and W = A^T B (6x3) are NOT stored, nor is v3Cam (up to BA_FAST_FREE adjustable cameras):
//   every consumer re-derives them from the point, the pose and the weighted derivatives -- the same expressions, the same bits.
//   Nothing of FindNewError's trial projection is stored either (the accepted state is projected again by the next step's
//   sweep): a trial writes 8 B per measurement (the squared error for the median).
// Sweeps per accepted LM trial: one fused step sweep over all slots (projection, Tukey weight; V / epsilon_b by the point's first
// lane out of LDS in slot order; U / epsilon_a: each lane's 27 products packed per camera in LDS, lane (camera, value) adds them
// in slot order), then per trial the Schur operands (F slots, matrix cores), the map update (F slots) and the trial error (all
// slots, static part only).  A slot's static fields are loaded a chunk ahead; points' leaders and counts come from ballots.
// Reductions are deterministic (fixed lane / wave order, no floating-point atomics).
#pragma once
#include "dev_math.h"

#ifndef VSLAM_BA_WPE
#define VSLAM_BA_WPE 2
#endif
// the phases of Bundle::Compute are out-of-line functions with their own register allocation (VSLAM_BA_INLINE_PHASES: diagnostic,
// everything in the kernel's allocation so that amdgpu_waves_per_eu bounds it)
#ifdef VSLAM_BA_INLINE_PHASES
#define BA_PHASE_FN __device__ __forceinline__
#else
#define BA_PHASE_FN __device__ __attribute__((noinline))
#endif
#ifndef BA_THREADS
#define BA_THREADS 256
#endif
// 4 waves; with amdgpu_waves_per_eu(2,2) on the kernel two problems share a CU (measured: 512 x 1 -4 %, 128 x 4 -4 %)
#define BA_LDS_N 60      // reduced camera systems up to 60 x 60 (10 adjustable cameras) are solved in LDS
#define BA_WAVES (BA_THREADS / 64)
#ifndef BA_ILP_PROJ
#define BA_ILP_PROJ 4   // projection passes: 4 measurements in flight (1 -> 4: -31 % on FindNewError once the view pointers were global and scalar; 6 spills: 5x slower)
#endif
#define BA_ILP_S 1      // Schur-complement tasks: 36 accumulators + two 6x3 blocks per lane leave no registers for a second point
#define BA_ILP 4        // independent measurements per thread and loop trip: the loops are memory-latency bound at 2 waves/SIMD
#define BA_ILP_C 4      // fused weight / derivative pass after an accepted step
#define BA_ILP_W 8      // pass 2 (weights): 9 operands per measurement, nothing else live
#define BA_ILP_P 6      // per-point loops over cameras (V, map update): 11 cameras in 2 trips, 5 adjustable ones in 1

#define MS_OK 0
#define MS_BAD 1      // bBad: z <= 0 or zero Tukey weight in this step
#define MS_ERASED 2   // erased from the measurement list (:517-528)

struct BaResult {
  int active;           // 0: nothing to do for this problem
  int computed;         // set by Bundle::Compute: an assembled problem is solved exactly once (the gated kernel is launched every frame)
  int go, nadj;         // k_ba_select -> k_ba_assemble: a problem is to be assembled for this stream; its number of adjustable cameras
  int n_cams, n_pts, n_meas, n_free;
  int accepted;         // Compute() return value (negative on error)
  int converged, hit_max;
  int counter;          // mnCounter
  int n_outlier_meas;
  double sigma2, lambda, lambda_factor;
  long long trials;
};

struct BaView {          // pointers already offset to one problem
  int max_cams, max_pts, max_meas;
  BaResult* res;
  Pose* cam_pose; Pose* cam_new; int* cam_fixed; int* cam_row; double* cam_U; double* cam_ea;
  double* pt_pos; double* pt_new; double* pt_V; double* pt_eb; int* pt_nmeas; int* pt_nout;
  // the caller's measurement list (AddMeas order)
  int* ms_p; int* ms_c; int* ms_state; double* ms_found; double* ms_sin;
  int* lut;              // [max_cams][max_pts] -> list index or -1 (GenerateMeasLUTs :566-575); input of the slot layout
  // slots (see the header comment)
  int* sl_info;          // camera | state << 8
  int* sl_pt; int* sl_logical; double* sl_found; double* sl_sin;
  double* sl_cm; double* sl_d; double* sl_eps;
  int* pt_offF; int* pt_offX;                      // [max_pts + 1] slot offsets per point inside region F / region X
  unsigned long long* pt_maskF;                    // [max_pts] adjustable-camera ordinals that measure the point
  int* chF; int* chX; int* ch_n;                   // chunk tables of the step sweep: first slot (region-relative) of every chunk (<= 64 slots, whole points); ch_n[0..3] = chunks F, chunks X, slots F, slots
  double* S; double* E; double* cam_up;
  double* scratch;       // [max_meas] squared error per slot (+inf: not in the median)
  int* outl;             // [max_meas][2] (p, c) in erase order
  int* free_cams;        // [max_cams] indices of the adjustable cameras
};

// The view as the phase functions use it: every pointer is a global-address-space pointer held in scalar registers.
// The phases are separate (noinline) functions; through a plain `const BaView&` they would see generic pointers that
// live in the caller's private memory: every access becomes a flat load of the pointer followed by a flat load of the
// datum, and every store may alias the view itself, so nothing can be hoisted.  Each phase therefore converts the view
// once at its top (ba_g): 64-bit values through readfirstlane (the view is uniform per workgroup), typed address_space(1).
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
struct BaViewG {
  int max_cams, max_pts, max_meas;
  BaResult AS1* res;
  Pose AS1* cam_pose; Pose AS1* cam_new; int AS1* cam_fixed; int AS1* cam_row; double AS1* cam_U; double AS1* cam_ea;
  double AS1* pt_pos; double AS1* pt_new; double AS1* pt_V; double AS1* pt_eb; int AS1* pt_nmeas; int AS1* pt_nout;
  int AS1* ms_p; int AS1* ms_c; int AS1* ms_state; double AS1* ms_found; double AS1* ms_sin;
  int AS1* lut;
  int AS1* sl_info; int AS1* sl_pt; int AS1* sl_logical; double AS1* sl_found; double AS1* sl_sin;
  double AS1* sl_cm; double AS1* sl_d; double AS1* sl_eps;
  int AS1* pt_offF; int AS1* pt_offX; unsigned long long AS1* pt_maskF;
  int AS1* chF; int AS1* chX; int AS1* ch_n;
  double AS1* S; double AS1* E; double AS1* cam_up;
  double AS1* scratch;
  int AS1* outl;
  int AS1* free_cams;
};
template <class T> DEVFN T AS1* ba_uniform_ptr(T* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (T AS1*)(((unsigned long long)hi << 32) | lo);
}
DEVFN BaViewG ba_g(const BaView& v) {
  BaViewG g;
  g.max_cams = __builtin_amdgcn_readfirstlane(v.max_cams); g.max_pts = __builtin_amdgcn_readfirstlane(v.max_pts); g.max_meas = __builtin_amdgcn_readfirstlane(v.max_meas);
#define BA_G(f) g.f = ba_uniform_ptr(v.f)
  BA_G(res); BA_G(cam_pose); BA_G(cam_new); BA_G(cam_fixed); BA_G(cam_row); BA_G(cam_U); BA_G(cam_ea);
  BA_G(pt_pos); BA_G(pt_new); BA_G(pt_V); BA_G(pt_eb); BA_G(pt_nmeas); BA_G(pt_nout);
  BA_G(ms_p); BA_G(ms_c); BA_G(ms_state); BA_G(ms_found); BA_G(ms_sin); BA_G(lut);
  BA_G(sl_info); BA_G(sl_pt); BA_G(sl_logical); BA_G(sl_found); BA_G(sl_sin); BA_G(sl_cm); BA_G(sl_d); BA_G(sl_eps);
  BA_G(pt_offF); BA_G(pt_offX); BA_G(pt_maskF); BA_G(chF); BA_G(chX); BA_G(ch_n);
  BA_G(S); BA_G(E); BA_G(cam_up); BA_G(scratch); BA_G(outl); BA_G(free_cams);
#undef BA_G
  return g;
}
DEVFN Pose ba_load_pose(const Pose AS1* p) {
  Pose T;
  _Pragma("unroll") for (int k = 0; k < 9; k++) T.R[k] = p->R[k];
  _Pragma("unroll") for (int k = 0; k < 3; k++) T.t[k] = p->t[k];
  return T;
}
DEVFN void ba_store_pose(Pose AS1* p, const Pose& T) {
  _Pragma("unroll") for (int k = 0; k < 9; k++) p->R[k] = T.R[k];
  _Pragma("unroll") for (int k = 0; k < 3; k++) p->t[k] = T.t[k];
}

// Diagnostic build only (-DVSLAM_BA_PROF): clock64() stamps of block 0 / lane 0 per phase of ba_compute, accumulated in
// g_ba_prof[phase]; read with vslam_debug_ba_prof().  Never compiled into the product library.
#ifdef VSLAM_BA_PROF
__device__ unsigned long long g_ba_prof[32];
#define BA_STAMP(id) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = clock64(); g_ba_prof[id] += t_ - ba_t0; ba_t0 = t_; } } while (0)
#else
#define BA_STAMP(id) do { } while (0)
#endif

// Measurement- and point-indexed fp64 arrays are component-major ("transposed"): component k of item i lives at
// arr[k * max + i], so the 64 lanes of a wave (consecutive i) touch 512 contiguous bytes per load/store instead of
// 64 scattered 8-byte words (the AoS form made pass 2 of Do_LM_Step 48 % of the kernel).
#define MS(arr, k, i) v.arr[(size_t)(k) * v.max_meas + (i)]
#define PT(arr, k, p) v.arr[(size_t)(k) * v.max_pts + (p)]

struct BaConfig { CamModel cam; int max_iterations; double convergence_limit, min_sigma2; int sum_order; /* vslam_params.ba_sum_order */ };

DEVFN double ba_wave_sum(double v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; }
DEVFN int ba_wave_sum_i(int v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; }

// block-wide sum of one double per thread; result returned to every thread. red: LDS [BA_WAVES]
DEVFN double ba_block_sum(double v, double* red) {
  v = ba_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0;
  for (int w = 0; w < BA_WAVES; w++) t += red[w];
  return t;
}
DEVFN int ba_block_sum_i(int v, int* red) {
  v = ba_wave_sum_i(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < BA_WAVES; w++) t += red[w];
  return t;
}

// Slot accessors: component k of slot i of a measurement-indexed fp64 array (component-major, see MS()).
#define SL(arr, k, i) v.arr[(size_t)(k) * v.max_meas + (i)]
#define SL_CAM(info) ((info) & 255)
#define SL_STATE(info) (((info) >> 8) & 3)
#define SL_INFO(cam, st) ((cam) | ((st) << 8))

// Jacobians of one measurement from its stored state (jni/Bundle.cc:262-300): cm = v3Cam, d = sqrt-inv-noise * weight *
// camera derivatives (2x2 row-major), R = rotation of the camera.  Same expressions wherever they are re-derived.
struct MeasState { double cm[3], d[4]; int st; };
DEVFN void ba_load_state(const BaViewG& v, int i, MeasState& m) {      // i < 0: loads slot 0, state forced to erased
  const int ic = i < 0 ? 0 : i;
  m.st = SL_STATE(v.sl_info[ic]);
  m.cm[0] = SL(sl_cm, 0, ic); m.cm[1] = SL(sl_cm, 1, ic); m.cm[2] = SL(sl_cm, 2, ic);
  m.d[0] = SL(sl_d, 0, ic); m.d[1] = SL(sl_d, 1, ic); m.d[2] = SL(sl_d, 2, ic); m.d[3] = SL(sl_d, 3, ic);
  if (i < 0) m.st = MS_ERASED;
}
// the F slot of point p in the adjustable camera of ordinal f, or -1
DEVFN int ba_slot_of(const BaViewG& v, int p, int f) {
  const unsigned long long mk = v.pt_maskF[p];
  if (!((mk >> f) & 1ull)) return -1;
  return v.pt_offF[p] + __popcll(mk & ((1ull << f) - 1ull));
}
// V*^-1 of point p (jni/Bundle.cc:329-347): V with its diagonal scaled by 1 + lambda, inverted; zero if a diagonal element of V is zero
DEVFN void ba_vstar_inv(const BaViewG& v, int p, double lambda, double Vi[9]) {
  const double v00 = PT(pt_V, 0, p), v10 = PT(pt_V, 1, p), v11 = PT(pt_V, 2, p), v20 = PT(pt_V, 3, p), v21 = PT(pt_V, 4, p), v22 = PT(pt_V, 5, p);
  if (v00 * v11 * v22 == 0) { _Pragma("unroll") for (int k = 0; k < 9; k++) Vi[k] = 0.0; return; }
  const double Vs[9] = {v00 * (1.0 + lambda), v10, v20, v10, v11 * (1.0 + lambda), v21, v20, v21, v22 * (1.0 + lambda)};
  inv3(Vs, Vi);
}
DEVFN void ba_jac_A(const double cm[3], const double d[4], double A[12]) {
  const double ooz = 1.0 / cm[2];
  const double cc[3] = {cm[0], cm[1], cm[2]};
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double f0, f1;
    se3_generator_motion(k, cc, ooz, f0, f1);
    A[k] = d[0] * f0 + d[1] * f1; A[6 + k] = d[2] * f0 + d[3] * f1;
  }
}
template <class RP>
DEVFN void ba_jac_B(RP R, const double cm[3], const double d[4], double B[6]) {
  const double ooz = 1.0 / cm[2];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double m0 = R[k], m1 = R[3 + k], m2 = R[6 + k];
    const double f0 = (m0 - cm[0] * m2 * ooz) * ooz, f1 = (m1 - cm[1] * m2 * ooz) * ooz;
    B[k] = d[0] * f0 + d[1] * f1; B[3 + k] = d[2] * f0 + d[3] * f1;
  }
}
// W = A^T B (6x3, :302) of measurement i in adjustable camera with rotation R
template <class RP>
DEVFN void ba_jac_W(const MeasState& m, RP R, double W[18]) {
  double A[12], B[6];
  ba_jac_A(m.cm, m.d, A);
  ba_jac_B(R, m.cm, m.d, B);
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int q = 0; q < 3; q++) W[r * 3 + q] = A[r] * B[q] + A[6 + r] * B[3 + q];
}

// Parallel in-place solve S x = E (n x n, row-major in global memory) by Gaussian elimination with partial
// pivoting (stands in for Eigen's mS.inverse()*vE, jni/Bundle.cc:437).  Result in E.  Returns false if singular.
DEVFN bool ba_block_solve(double* S, double* E, int n, int* ired) {
  for (int k = 0; k < n; k++) {
    if (threadIdx.x == 0) {
      int piv = k; double best = fabs(S[(size_t)k * n + k]);
      for (int r = k + 1; r < n; r++) { const double a = fabs(S[(size_t)r * n + k]); if (a > best) { best = a; piv = r; } }
      ired[0] = best == 0.0 ? -1 : piv;
    }
    __syncthreads();
    const int piv = ired[0];
    if (piv < 0) return false;
    if (piv != k) {
      for (int c = threadIdx.x; c < n; c += blockDim.x) { const double t = S[(size_t)k * n + c]; S[(size_t)k * n + c] = S[(size_t)piv * n + c]; S[(size_t)piv * n + c] = t; }
      if (threadIdx.x == 0) { const double t = E[k]; E[k] = E[piv]; E[piv] = t; }
    }
    __syncthreads();
    const double inv = 1.0 / S[(size_t)k * n + k];
    const int rem = n - k - 1;
    // every row r > k: f = S[r][k] * inv; row r -= f * row k
    for (int t = threadIdx.x; t < rem * (rem + 1); t += blockDim.x) {
      const int r = k + 1 + t / (rem + 1), c = k + 1 + t % (rem + 1);   // c == n -> the right-hand side
      const double f = S[(size_t)r * n + k] * inv;
      if (c < n) S[(size_t)r * n + c] -= f * S[(size_t)k * n + c];
      else E[r] -= f * E[k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int k = n - 1; k >= 0; k--) {
      double s = E[k];
      for (int c = k + 1; c < n; c++) s -= S[(size_t)k * n + c] * E[c];
      E[k] = s / S[(size_t)k * n + k];
    }
  __syncthreads();
  return true;
}

// The same solve with the augmented system held in LDS (n <= BA_LDS_N): pivot search by wave shuffles, elimination by
// the whole workgroup, back-substitution by wave 0.  A: LDS [n][n+1].
DEVFN bool ba_block_solve_lds(const double* S, double* E, int n, double* A, int* ired) {
  const int ld = n + 1, lane = threadIdx.x & 63;
  for (int t = threadIdx.x; t < n * ld; t += blockDim.x) { const int r = t / ld, c = t - r * ld; A[t] = c < n ? S[(size_t)r * n + c] : E[r]; }
  __syncthreads();
  for (int k = 0; k < n; k++) {
    if (threadIdx.x < 64) {                                        // partial pivoting: first row of maximal |A[r][k]|, r >= k
      double best = -1.0; int piv = k;
      for (int r = k + lane; r < n; r += 64) { const double a = fabs(A[r * ld + k]); if (a > best) { best = a; piv = r; } }
      for (int d = 32; d > 0; d >>= 1) {
        const double ob = __shfl_xor(best, d); const int op = __shfl_xor(piv, d);
        if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
      }
      if (lane == 0) ired[0] = best == 0.0 ? -1 : piv;
    }
    __syncthreads();
    const int piv = ired[0];
    if (piv < 0) return false;
    if (piv != k) for (int c = threadIdx.x; c < ld; c += blockDim.x) { const double t = A[k * ld + c]; A[k * ld + c] = A[piv * ld + c]; A[piv * ld + c] = t; }
    __syncthreads();
    const double inv = 1.0 / A[k * ld + k];
    const int rem = n - k - 1, wid = ld - k - 1;                    // columns k+1 .. n (the last one is the right-hand side)
    for (int t = threadIdx.x; t < rem * wid; t += blockDim.x) {
      const int r = k + 1 + t / wid, c = k + 1 + t % wid;
      const double f = A[r * ld + k] * inv;
      A[r * ld + c] -= f * A[k * ld + c];
    }
    __syncthreads();
  }
  if (threadIdx.x < 64) {
    for (int k = n - 1; k >= 0; k--) {
      double s = 0.0;
      for (int c = k + 1 + lane; c < n; c += 64) s += A[k * ld + c] * A[c * ld + n];
      for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
      if (lane == 0) A[k * ld + n] = (A[k * ld + n] - s) / A[k * ld + k];
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < n; t += blockDim.x) E[t] = A[t * ld + n];
  __syncthreads();
  return true;
}

// The reduced camera system of up to BA_WSOLVE_N unknowns (5 adjustable cameras) solved by ONE wavefront with the augmented
// matrix in registers: lane r holds row r, the pivot row is broadcast with v_readlane, no LDS and no barrier inside.
// Same pivoting rule (first row of maximal |A[r][k]|) and the same multiply / subtract per element as ba_block_solve, the
// back-substitution sums in ascending column order like lu_solve_n.  Every index is a compile-time constant after
// unrolling.  Called by wavefront 0 only; returns false if singular.  Writes the solution to E.
#define BA_WSOLVE_N 30
#ifndef VSLAM_BA_WSOLVE
#define VSLAM_BA_WSOLVE 1   // measured (tools/ba_phase_profile*.py, 512 problems): 30 x 30 solve 116 kcycles by one wavefront in registers, 160 by the workgroup in LDS; 24 x 24: 103 / 108
#endif
DEVFN double ba_readlane_d(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
BA_PHASE_FN bool ba_solve_wave(const BaView& v_, int n) {
  const BaViewG v = ba_g(v_);
  constexpr int N = BA_WSOLVE_N;
  const int lane = threadIdx.x & 63;
  double a[N + 1];                                                   // a[0..N-1]: the row; a[N]: right-hand side
  _Pragma("unroll") for (int c = 0; c < N; c++) a[c] = (lane < n && c < n) ? v.S[(size_t)lane * n + c] : (c == lane ? 1.0 : 0.0);
  a[N] = lane < n ? v.E[lane] : 0.0;
  bool ok = true;
  _Pragma("unroll") for (int k = 0; k < N; k++) {
    if (k < n && ok) {
      double best = (lane >= k && lane < n) ? fabs(a[k]) : -1.0;
      int piv = lane;
      _Pragma("unroll") for (int d = 32; d > 0; d >>= 1) {
        const double ob = __shfl_xor(best, d); const int op = __shfl_xor(piv, d);
        if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
      }
      piv = __builtin_amdgcn_readfirstlane(piv);
      if (best == 0.0) ok = false;
      else {
        double rowk[N + 1];
        _Pragma("unroll") for (int c = k; c <= N; c++) {
          const double vk = ba_readlane_d(a[c], k), vp = ba_readlane_d(a[c], piv);
          a[c] = lane == k ? vp : (lane == piv ? vk : a[c]);         // rows k and piv change places (a no-op when piv == k)
          rowk[c] = vp;
        }
        const double inv = 1.0 / rowk[k];
        if (lane > k && lane < n) {
          const double f = a[k] * inv;
          _Pragma("unroll") for (int c = k + 1; c <= N; c++) a[c] -= f * rowk[c];
        }
      }
    }
  }
  if (!ok) return false;
  double x[N];
  _Pragma("unroll") for (int k = N - 1; k >= 0; k--) {
    x[k] = 0.0;
    if (k < n) {
      double s = a[N];
      _Pragma("unroll") for (int c = k + 1; c < N; c++) if (c < n) s -= a[c] * x[c];
      const double xk = ba_readlane_d(s / a[k], k);
      x[k] = xk;
      if (lane == k) v.E[k] = xk;
    }
  }
  return true;
}


// ---------------------------------------------------------------------------------------------------------------------
// Slot layout, built once per Compute from the caller's list + lut.
// ---------------------------------------------------------------------------------------------------------------------
#define SL_FORD(info) (((info) >> 16) & 255)                 // ordinal of the (adjustable) camera, 255 for a fixed one
#define SL_MAKE(cam, st, ford) ((cam) | ((st) << 8) | ((ford) << 16))
#define SL_HASF (1 << 24)                                     // X slot of a point that adjustable cameras measure too (its V / epsilon_b sums continue the F sweep's)
#define SL_HAS_F(info) (((info) >> 24) & 1)
#define SL_WITH_STATE(info, st) (((info) & ~(3 << 8)) | ((st) << 8))

// a[0..n) counts -> exclusive offsets, a[n] = total (returned); all threads call
DEVFN int ba_block_excl_scan(int AS1* a, int n, int* ired) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (n + BA_THREADS - 1) / BA_THREADS;
  const int lo = min((int)threadIdx.x * per, n), hi = min(lo + per, n);
  int sum = 0;
  for (int i = lo; i < hi; i++) sum += a[i];
  int inc = sum;
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
  __syncthreads();
  if (lane == 63) ired[wave] = inc;
  __syncthreads();
  int run = inc - sum, total = 0;
  for (int w = 0; w < BA_WAVES; w++) { if (w < wave) run += ired[w]; total += ired[w]; }
  for (int i = lo; i < hi; i++) { const int t = a[i]; a[i] = run; run += t; }
  if (threadIdx.x == 0) a[n] = total;
  __syncthreads();
  return total;
}

#define BA_LAYOUT_CB 12   // cameras whose LUT entries of a point are loaded together (one memory round trip instead of one per camera)
BA_PHASE_FN void ba_build_layout(const BaView& v_, int nc, int np, int* ired, int* lds_i /* LDS, 2 * (max_pts + 1) ints */) {
  const BaViewG v = ba_g(v_);
  // 1. per point: how many adjustable / fixed cameras measure it, which adjustable ones; V and epsilon_b start at zero
  for (int p = threadIdx.x; p < np; p += BA_THREADS) {
    int nF = 0, nX = 0; unsigned long long mk = 0;
    for (int c0 = 0; c0 < nc; c0 += BA_LAYOUT_CB) {
      int li[BA_LAYOUT_CB];
      _Pragma("unroll") for (int u = 0; u < BA_LAYOUT_CB; u++) li[u] = c0 + u < nc ? v.lut[(size_t)(c0 + u) * v.max_pts + p] : -1;
      _Pragma("unroll") for (int u = 0; u < BA_LAYOUT_CB; u++) {
        if (li[u] < 0) continue;
        const int c = c0 + u;
        if (v.cam_fixed[c]) nX++; else { nF++; mk |= 1ull << (v.cam_row[c] / 6); }
      }
    }
    v.pt_offF[p] = nF; v.pt_offX[p] = nX; v.pt_maskF[p] = mk;
    _Pragma("unroll") for (int k = 0; k < 6; k++) PT(pt_V, k, p) = 0.0;
    _Pragma("unroll") for (int k = 0; k < 3; k++) PT(pt_eb, k, p) = 0.0;
  }
  __syncthreads();
  const int MF = ba_block_excl_scan(v.pt_offF, np, ired);
  const int MX = ba_block_excl_scan(v.pt_offX, np, ired);
  // 2. the slots: adjustable cameras in ordinal order, then (region X) fixed cameras in index order
  for (int p = threadIdx.x; p < np; p += BA_THREADS) {
    int kf = v.pt_offF[p], kx = MF + v.pt_offX[p];
    const bool hasF = v.pt_offF[p + 1] - kf > 0;
    for (int c0 = 0; c0 < nc; c0 += BA_LAYOUT_CB) {
      int li[BA_LAYOUT_CB];
      _Pragma("unroll") for (int u = 0; u < BA_LAYOUT_CB; u++) li[u] = c0 + u < nc ? v.lut[(size_t)(c0 + u) * v.max_pts + p] : -1;
      double f0[BA_LAYOUT_CB], f1[BA_LAYOUT_CB], sn[BA_LAYOUT_CB];       // the measurements' static data, all requested before the first is stored
      _Pragma("unroll") for (int u = 0; u < BA_LAYOUT_CB; u++) {
        const int i = li[u] < 0 ? 0 : li[u];
        f0[u] = MS(ms_found, 0, i); f1[u] = MS(ms_found, 1, i); sn[u] = v.ms_sin[i];
      }
      _Pragma("unroll") for (int u = 0; u < BA_LAYOUT_CB; u++) {
        const int i = li[u];
        if (i < 0) continue;
        const int c = c0 + u;
        const bool fx = v.cam_fixed[c] != 0;
        const int s = fx ? kx++ : kf++;
        v.sl_info[s] = SL_MAKE(c, MS_OK, fx ? 255 : v.cam_row[c] / 6) | (fx && hasF ? SL_HASF : 0);
        v.sl_pt[s] = p; v.sl_logical[s] = i;
        SL(sl_found, 0, s) = f0[u]; SL(sl_found, 1, s) = f1[u];
        v.sl_sin[s] = sn[u];
      }
    }
  }
  // 3. chunk tables of the step sweep: consecutive whole points, at most 64 slots (one lane each).  Wavefront 0 cuts region F,
  //    wavefront 1 region X: 64 lanes look at the next 64 points at once, the first whose end leaves the chunk's 64 slots is the cut.
  int* oF = lds_i; int* oX = lds_i + (np + 1);
  for (int p = threadIdx.x; p <= np; p += BA_THREADS) { oF[p] = v.pt_offF[p]; oX[p] = v.pt_offX[p]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 2) {
    const int* o = wave == 0 ? oF : oX;
    int AS1* ch = wave == 0 ? v.chF : v.chX;
    int k = 0, start = 0;
    if (np > 0 && lane == 0) ch[0] = o[0];
    while (start < np) {
      // the cut is the first point p > start with o[p + 1] - o[start] > 64 (a point with more than 64 slots stands alone)
      const int ostart = o[start];
      int cut = -1;
      for (int base = start + 1; base < np && cut < 0; base += 64) {
        const int pp = base + lane;
        const bool over = pp < np && o[pp + 1] - ostart > 64;
        const unsigned long long bm = __ballot(over);
        if (bm) cut = base + (int)__ffsll((long long)bm) - 1;
      }
      if (cut < 0) break;
      k++;
      if (lane == 0) ch[k] = o[cut];
      start = cut;
    }
    if (np > 0) { k++; if (lane == 0) ch[k] = o[np]; }
    if (lane == 0) {
      v.ch_n[wave] = k;
      if (wave == 0) { v.ch_n[2] = MF; v.ch_n[3] = MF + MX; }
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// FindNewError (jni/Bundle.cc:537-561) at the trial state, or -- trial == 0 -- pass 1 of Do_LM_Step (:209-215) at the committed
// state: this thread's share of the objective.  It also prepares the median of the next Do_LM_Step: the squared error of every
// measurement that stays in the list (state OK now, in front of the camera) goes to `scratch`, +inf for the others, and their
// number is returned.  Reads the static 32 B of a slot, writes 8 B.
// ---------------------------------------------------------------------------------------------------------------------
struct BaNewError { double ne; int nvalid; };
BA_PHASE_FN BaNewError ba_find_new_error(const BaView& v_, const BaConfig& cfg_, int M, double sigma2, int trial, double* lds_ /* LDS: 12 doubles per camera */) {
  const BaViewG v = ba_g(v_);
  const BaConfig cfg = cfg_;
  const Pose AS1* camsG = trial ? v.cam_new : v.cam_pose;
  double AS3* camL = (double AS3*)lds_;                                // the poses out of LDS: one dependent global round trip less per slot
  {
    const int nc = v.res->n_cams;
    __syncthreads();
    for (int t = threadIdx.x; t < nc * 12; t += BA_THREADS) camL[t] = ((const double AS1*)camsG)[t];
    __syncthreads();
  }
  const double AS1* pts = trial ? v.pt_new : v.pt_pos;
  double ne = 0.0;
  int nv = 0;
  // The slot fields of the NEXT batch are requested together with this batch's points (which hang on this batch's slot fields):
  // one exposed memory round trip per batch instead of two.  (Deeper pipelining does not survive the compiler: with the stores to
  // `scratch` in the loop, loads and stores share vmcnt and every first use waits for zero.)
  struct SlotBatch { int info[BA_ILP_PROJ], mp[BA_ILP_PROJ]; double f0[BA_ILP_PROJ], f1[BA_ILP_PROJ], sn[BA_ILP_PROJ]; };
  auto load_batch = [&](int i0, SlotBatch& b) {
    _Pragma("unroll") for (int u = 0; u < BA_ILP_PROJ; u++) {
      const int i = i0 + u * BA_THREADS, ic = i < M ? i : M - 1;
      b.info[u] = v.sl_info[ic]; b.mp[u] = v.sl_pt[ic];
      if (i >= M) b.info[u] = SL_WITH_STATE(b.info[u], MS_ERASED);
      b.f0[u] = SL(sl_found, 0, ic); b.f1[u] = SL(sl_found, 1, ic); b.sn[u] = v.sl_sin[ic];
    }
  };
  SlotBatch cb;
  if ((int)threadIdx.x < M) load_batch(threadIdx.x, cb);
  for (int i0 = threadIdx.x; i0 < M; i0 += BA_ILP_PROJ * BA_THREADS) {
    Pose T[BA_ILP_PROJ]; double X[BA_ILP_PROJ][3];
    _Pragma("unroll") for (int u = 0; u < BA_ILP_PROJ; u++) {
      { const int cj = SL_CAM(cb.info[u]); _Pragma("unroll") for (int q = 0; q < 9; q++) T[u].R[q] = camL[cj * 12 + q]; _Pragma("unroll") for (int q = 0; q < 3; q++) T[u].t[q] = camL[cj * 12 + 9 + q]; }
      _Pragma("unroll") for (int k = 0; k < 3; k++) X[u][k] = pts[3 * cb.mp[u] + k];
    }
    SlotBatch nb = cb;
    const int in = i0 + BA_ILP_PROJ * BA_THREADS;
    if (in < M) load_batch(in, nb);
    _Pragma("unroll") for (int u = 0; u < BA_ILP_PROJ; u++) {
      const int st = SL_STATE(cb.info[u]);
      if (st == MS_ERASED) continue;
      const int i = i0 + u * BA_THREADS;
      double c[3];
      pose_xform(T[u], X[u], c);
      if (c[2] <= 0) { ne += 1.0; v.scratch[i] = __builtin_huge_val(); continue; }
      const CamProj pr = cam_project(cfg.cam, c[0] / c[2], c[1] / c[2]);
      const double e0 = (cb.f0[u] - pr.im[0]) * cb.sn[u], e1 = (cb.f1[u] - pr.im[1]) * cb.sn[u];
      const double e2 = e0 * e0 + e1 * e1;
      ne += tukey_objective(e2, sigma2);
      const bool stays = st == MS_OK;                                 // MS_BAD ones are erased at the end of this step
      v.scratch[i] = stays ? e2 : __builtin_huge_val();
      nv += stays ? 1 : 0;
    }
    cb = nb;
  }
  BaNewError r; r.ne = ne; r.nvalid = nv;
  return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// The step sweep: passes 1 and 2 of Do_LM_Step (jni/Bundle.cc:209-321) over the slots of one region, one lane per slot, a chunk
// of whole points per wavefront trip.  Per slot: projection, camera derivatives, Tukey weight (the median is known: the
// squared errors are those FindNewError left), weighted epsilon / derivatives stored for the consumers of this step; the
// objective.  V and epsilon_b (:49-56, :312-316): every lane leaves its 9 products in LDS and the first lane of each point adds
// the point's entries in slot order (region F: the adjustable cameras in order; region X continues the sum with the fixed
// ones).  U and epsilon_a (:40-47, :306-311; region F, up to BA_MFMA_FREE adjustable cameras): 27 products per lane, summed per
// camera by masked wavefront reductions and carried in registers over the chunks; left per wavefront in `ured`.
// Returns this thread's share of the objective (pass 2's dCurrentError).
// ---------------------------------------------------------------------------------------------------------------------
#define BA_MFMA_FREE 5    // adjustable cameras of one pass of the matrix-core form of the reduced camera system
#define BA_FAST_FREE 10   // adjustable cameras up to which U / epsilon_a are summed inside the sweep and the map update runs over the dense F slots
#define BA_U_PAIRS ((BA_FAST_FREE * 27 + 63) / 64)
#define BA_MAX_CAMS_LDS 128  // cameras whose poses the step sweep keeps in LDS (= the most keyframes a map holds)
#define BA_SWEEP_STAGE 27   // doubles per lane of a wavefront's staging area in the step sweep (V / epsilon_b use 9 of them)
BA_PHASE_FN double ba_step_sweep(const BaView& v_, const BaConfig& cfg_, int region, double sigma2, int nfree,
                                                        double* stg_ /* LDS [BA_WAVES][64][BA_SWEEP_STAGE] */, double* ured_ /* unused: a wavefront leaves its U / epsilon_a sums at the head of its own staging area, [camera][27] */,
                                                        const double* cams_ /* LDS: the committed camera poses, 12 doubles each */) {
  const BaViewG v = ba_g(v_);
  const BaConfig cfg = cfg_;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double AS3* stg = (double AS3*)stg_ + wave * 64 * BA_SWEEP_STAGE;
  const double AS3* camL = (const double AS3*)cams_;
  const int nch = v.ch_n[region];
  const int AS1* ch = region ? v.chX : v.chF;
  const int base = region ? v.ch_n[2] : 0;
  const bool fastU = region == 0 && nfree <= BA_FAST_FREE;
  double cur = 0.0;
  double uacc[BA_U_PAIRS];                                             // lane's sums: (camera, value) pairs lane, lane + 64, lane + 128
  _Pragma("unroll") for (int k = 0; k < BA_U_PAIRS; k++) uacc[k] = 0.0;
  // The static part of a slot (32 B) is loaded a chunk ahead: the chunk table entry of the next chunk at the top of this one,
  // its slots' fields behind this chunk's projection -- a wavefront has at most one other wavefront on its SIMD to hide a
  // dependent chain of global loads behind.  The camera poses come out of LDS.
  struct SlotIn { int info, pt; double f0, f1, sn; };
  auto load_in = [&](int a, int n, SlotIn& r) {
    const int s = a + (lane < n ? lane : 0);
    r.info = v.sl_info[s]; r.pt = v.sl_pt[s]; r.f0 = SL(sl_found, 0, s); r.f1 = SL(sl_found, 1, s); r.sn = v.sl_sin[s];
  };
  int cs0 = 0, cs1 = 0;
  SlotIn cin; cin.info = 0; cin.pt = 0; cin.f0 = 0; cin.f1 = 0; cin.sn = 0;
  double cX[3] = {0, 0, 0};                                            // the point of this lane's slot: requested at the end of the previous chunk
  if (wave < nch) {
    cs0 = ch[wave]; cs1 = ch[wave + 1]; load_in(base + cs0, min(64, cs1 - cs0), cin);
    _Pragma("unroll") for (int q = 0; q < 3; q++) cX[q] = v.pt_pos[3 * cin.pt + q];
  }
  // What a slot leaves for the later phases of the step: its state (every region); the weighted camera derivatives d of an F slot
  // (the reduced camera system and the map update re-derive W from them; v3Cam is re-derived there from the point and the pose);
  // v3Cam and epsilon too when more than BA_FAST_FREE cameras are adjusted (the wave-per-block forms read them).  An X slot is
  // consumed here and now: nothing else is stored for it.
  const bool storeD = region == 0, storeAll = region == 0 && nfree > BA_FAST_FREE;
  for (int k = wave; k < nch; k += BA_WAVES) {
    const int a0 = base + cs0, ntot = cs1 - cs0;                       // <= 64 slots, or ONE point with more (a keyframe-rich map: fixed cameras)
    const int kn = k + BA_WAVES;
    int ns0 = 0, ns1 = 0;
    if (kn < nch) { ns0 = ch[kn]; ns1 = ch[kn + 1]; }
    SlotIn nin = cin;
    const int ntrip = (ntot + 63) >> 6;
    double carry[9];                                                   // lane 0: the running sums of a point that spans several trips
    _Pragma("unroll") for (int q = 0; q < 9; q++) carry[q] = 0.0;
    for (int trip = 0; trip < (ntrip > 0 ? ntrip : 1); trip++) {
    const int a = a0 + 64 * trip, n = min(64, ntot - 64 * trip);
    const bool act = lane < n;
    const int s = a + (act ? lane : 0);
    SlotIn in = cin;
    if (trip > 0) load_in(a, n, in);
    const int info = in.info, pt = in.pt;
    const double f0 = in.f0, f1 = in.f1, sn = in.sn;
    const int cam = SL_CAM(info);
    Pose T;
    _Pragma("unroll") for (int q = 0; q < 9; q++) T.R[q] = camL[cam * 12 + q];
    _Pragma("unroll") for (int q = 0; q < 3; q++) T.t[q] = camL[cam * 12 + 9 + q];
    double X[3];
    _Pragma("unroll") for (int q = 0; q < 3; q++) X[q] = cX[q];
    if (trip > 0) { _Pragma("unroll") for (int q = 0; q < 3; q++) X[q] = v.pt_pos[3 * pt + q]; }
    // region X continues the sums the F sweep left for the point: loaded now (every lane of the point, one broadcast access),
    // used by the point's first lane after the projection
    const bool cont = region == 1 && SL_HAS_F(info);
    double vprev[9];
    _Pragma("unroll") for (int q = 0; q < 6; q++) vprev[q] = cont ? PT(pt_V, q, pt) : 0.0;
    _Pragma("unroll") for (int q = 0; q < 3; q++) vprev[6 + q] = cont ? PT(pt_eb, q, pt) : 0.0;
    int st = act ? SL_STATE(info) : MS_ERASED;
    bool valid = false;
    double c[3] = {0, 0, 1}, d[4] = {0, 0, 0, 0}, e0 = 0, e1 = 0;
    if (st != MS_ERASED) {
      pose_xform(T, X, c);
      if (storeAll) { SL(sl_cm, 0, s) = c[0]; SL(sl_cm, 1, s) = c[1]; SL(sl_cm, 2, s) = c[2]; }
      if (c[2] <= 0) { st = MS_BAD; cur += 1.0; }                       // pass 1: bBad (:186-189); pass 2: :243-246
      else {
        const CamProj pr = cam_project(cfg.cam, c[0] / c[2], c[1] / c[2]);
        double dd[4];
        cam_derivs(cfg.cam, pr, dd);
        e0 = (f0 - pr.im[0]) * sn; e1 = (f1 - pr.im[1]) * sn;
        const double e2 = e0 * e0 + e1 * e1;
        const double dWeight = tukey_sqrt_weight(e2, sigma2);
        e0 *= dWeight; e1 *= dWeight;
        if (storeAll) { SL(sl_eps, 0, s) = e0; SL(sl_eps, 1, s) = e1; }
        if (dWeight == 0) { st = MS_BAD; cur += 1.0; }
        else {
          st = MS_OK; valid = true;
          cur += tukey_objective(e2, sigma2);
          _Pragma("unroll") for (int q = 0; q < 4; q++) { d[q] = sn * (dWeight * dd[q]); if (storeD) SL(sl_d, q, s) = d[q]; }   // weighted from here on
        }
      }
      v.sl_info[s] = SL_WITH_STATE(info, st);
    }
    // ---- V, epsilon_b ----
    {
      double pr9[9];
      _Pragma("unroll") for (int q = 0; q < 9; q++) pr9[q] = 0.0;
      if (valid) {
        double B[6];
        ba_jac_B(T.R, c, d, B);
        int q = 0;
        _Pragma("unroll") for (int r = 0; r < 3; r++) for (int cc = 0; cc <= r; cc++) pr9[q++] = B[r] * B[cc] + B[3 + r] * B[3 + cc];   // :49-56 LL triangle
        _Pragma("unroll") for (int r = 0; r < 3; r++) pr9[6 + r] = B[r] * e0 + B[3 + r] * e1;
      }
      _Pragma("unroll") for (int q = 0; q < 9; q++) stg[lane * 9 + q] = pr9[q];
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      if (kn < nch && trip == (ntrip > 0 ? ntrip : 1) - 1) load_in(base + ns0, min(64, ns1 - ns0), nin);   // the next chunk's slots, behind the sums below
      const int ptprev = __shfl_up(pt, 1);
      const bool leader = act && (lane == 0 || ptprev != pt);          // first slot of the point in this region (or in this trip of its slots)
      const unsigned long long lm = __ballot(leader);
      if (leader) {
        const unsigned long long above = lane < 63 ? lm >> (lane + 1) : 0ull;   // the point's slots end where the next point's begin
        const int cnt = above ? (int)__ffsll((long long)above) : n - lane;
        double acc[9];
        if (trip == 0) {
          _Pragma("unroll") for (int q = 0; q < 9; q++) acc[q] = vprev[q];
        } else {
          _Pragma("unroll") for (int q = 0; q < 9; q++) acc[q] = carry[q];
        }
        int j = 0;
        for (; j + 2 <= cnt; j += 2) {                                  // two slots' entries in flight, added in slot order
          double x0[9], x1[9];
          _Pragma("unroll") for (int q = 0; q < 9; q++) { x0[q] = stg[(lane + j) * 9 + q]; x1[q] = stg[(lane + j + 1) * 9 + q]; }
          _Pragma("unroll") for (int q = 0; q < 9; q++) acc[q] = (acc[q] + x0[q]) + x1[q];
        }
        if (j < cnt) { _Pragma("unroll") for (int q = 0; q < 9; q++) acc[q] += stg[(lane + j) * 9 + q]; }
        _Pragma("unroll") for (int q = 0; q < 9; q++) carry[q] = acc[q];
        if (trip == ntrip - 1 || ntrip <= 1) {
          // lower triangle of V: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) -> components 0..5
          _Pragma("unroll") for (int q = 0; q < 6; q++) PT(pt_V, q, pt) = acc[q];
          _Pragma("unroll") for (int q = 0; q < 3; q++) PT(pt_eb, q, pt) = acc[6 + q];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // ---- U, epsilon_a ----
    // Every lane leaves the 27 products of its own camera in LDS, the slots of a camera packed together in slot order (rank by
    // ballots); lane (f, q) then adds value q of camera f's entries in that order.  (The predecessor multiplied the 27 products
    // by a 0 / 1 mask for each of the five cameras and reduced each masked set across the wavefront: five times the products and
    // ~1500 shuffle-address / select / bpermute instructions per chunk.)
    if (fastU) {
      double A[12];
      _Pragma("unroll") for (int q = 0; q < 12; q++) A[q] = 0.0;
      if (valid) ba_jac_A(c, d, A);
      const int ford = valid ? SL_FORD(info) : 255;
      int offf[BA_FAST_FREE], cntf[BA_FAST_FREE];
      int mypos = 0, run = 0;
      _Pragma("unroll") for (int f = 0; f < BA_FAST_FREE; f++) {
        offf[f] = run; cntf[f] = 0;
        if (f < nfree) {                                               // (uniform)
          const unsigned long long mk = __ballot(ford == f);
          const int cf = (int)__popcll(mk);
          if (ford == f) mypos = run + (int)__popcll(mk & ((1ull << lane) - 1ull));
          cntf[f] = cf; run += cf;
        }
      }
      if (ford < BA_FAST_FREE) {
        double AS3* dst = stg + mypos * 27;
        int q = 0;
        _Pragma("unroll") for (int r = 0; r < 6; r++) for (int cc = 0; cc <= r; cc++) dst[q++] = A[r] * A[cc] + A[6 + r] * A[6 + cc];       // :40-47
        _Pragma("unroll") for (int r = 0; r < 6; r++) dst[21 + r] = A[r] * e0 + A[6 + r] * e1;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      _Pragma("unroll") for (int k = 0; k < BA_U_PAIRS; k++) {
        if (64 * k >= nfree * 27) break;                               // (uniform)
        const int pid = lane + 64 * k;
        if (pid < nfree * 27) {
          const int f = pid / 27, q = pid - 27 * f;
          int o = offf[0], cf = cntf[0];
          _Pragma("unroll") for (int g = 1; g < BA_FAST_FREE; g++) if (g < nfree && f == g) { o = offf[g]; cf = cntf[g]; }
          double acc = uacc[k];
          const double AS3* src = stg + o * 27 + q;
          int r = 0;
          for (; r + 8 <= cf; r += 8) {                                 // eight LDS reads in flight, the additions in slot order
            double x[8];
            _Pragma("unroll") for (int u = 0; u < 8; u++) x[u] = src[(r + u) * 27];
            _Pragma("unroll") for (int u = 0; u < 8; u++) acc += x[u];
          }
          {
            double x[8];
            _Pragma("unroll") for (int u = 0; u < 8; u++) x[u] = r + u < cf ? src[(r + u) * 27] : 0.0;
            _Pragma("unroll") for (int u = 0; u < 8; u++) if (r + u < cf) acc += x[u];
          }
          uacc[k] = acc;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    }
    if (kn < nch) { _Pragma("unroll") for (int q = 0; q < 3; q++) cX[q] = v.pt_pos[3 * nin.pt + q]; }   // the next chunk's points (its slot fields have arrived by now)
    cs0 = ns0; cs1 = ns1; cin = nin;
  }
  if (fastU) {
    (void)ured_;
    _Pragma("unroll") for (int k = 0; k < BA_U_PAIRS; k++) {
      const int pid = lane + 64 * k;
      if (pid < nfree * 27) stg[pid] = uacc[k];                        // [camera][27]: the staging area is free after the last chunk
    }
  }
  return cur;
}

// U, epsilon_a for more than BA_MFMA_FREE adjustable cameras: one wavefront per camera, lanes stride over the points
BA_PHASE_FN void ba_accum_U_generic(const BaView& v_, int nfree, int np) {
  const BaViewG v = ba_g(v_);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int f = wave; f < nfree; f += BA_WAVES) {
    const int j = v.free_cams[f];
    double acc[27];
    _Pragma("unroll") for (int k = 0; k < 27; k++) acc[k] = 0.0;
    for (int p = lane; p < np; p += 64) {
      const int s = ba_slot_of(v, p, f);
      MeasState ms;
      ba_load_state(v, s, ms);
      if (ms.st != MS_OK) continue;
      const double e0 = SL(sl_eps, 0, s), e1 = SL(sl_eps, 1, s);
      double A[12];
      ba_jac_A(ms.cm, ms.d, A);
      int q = 0;
      _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) acc[q++] += A[r] * A[c] + A[6 + r] * A[6 + c];
      _Pragma("unroll") for (int r = 0; r < 6; r++) acc[21 + r] += A[r] * e0 + A[6 + r] * e1;
    }
    _Pragma("unroll") for (int k = 0; k < 27; k++) acc[k] = ba_wave_sum(acc[k]);
    if (lane == 0) {
      double AS1* U = v.cam_U + 36 * j;
      int q = 0;
      _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) U[r * 6 + c] = acc[q++];
      _Pragma("unroll") for (int r = 0; r < 6; r++) v.cam_ea[6 * j + r] = acc[21 + r];
    }
  }
}

// S diagonal block + E of one adjustable camera (jni/Bundle.cc:362-396); called by one wavefront.  (More than BA_MFMA_FREE cameras.)
BA_PHASE_FN void ba_task_diag(const BaView& v_, int task, int np, int nS, double lambda) {
  const BaViewG v = ba_g(v_);
  const int lane = threadIdx.x & 63;
  const int j = v.free_cams[task], row = v.cam_row[j];
  double acc[27];
  _Pragma("unroll") for (int k = 0; k < 27; k++) acc[k] = 0.0;
  for (int p = lane; p < np; p += 64) {
    MeasState ms;
    ba_load_state(v, ba_slot_of(v, p, task), ms);
    if (ms.st != MS_OK) continue;
    double Vi[9], W[18], Y[18];
    ba_vstar_inv(v, p, lambda, Vi);
    const double eb[3] = {PT(pt_eb, 0, p), PT(pt_eb, 1, p), PT(pt_eb, 2, p)};
    ba_jac_W(ms, v.cam_pose[j].R, W);
    _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) Y[r * 3 + c] = W[r * 3] * Vi[c] + W[r * 3 + 1] * Vi[3 + c] + W[r * 3 + 2] * Vi[6 + c];
    int q = 0;
    _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) acc[q++] += Y[r * 3] * W[c * 3] + Y[r * 3 + 1] * W[c * 3 + 1] + Y[r * 3 + 2] * W[c * 3 + 2];
    double ve[3];
    _Pragma("unroll") for (int r = 0; r < 3; r++) ve[r] = Vi[r * 3] * eb[0] + Vi[r * 3 + 1] * eb[1] + Vi[r * 3 + 2] * eb[2];
    _Pragma("unroll") for (int r = 0; r < 6; r++) acc[21 + r] += W[r * 3] * ve[0] + W[r * 3 + 1] * ve[1] + W[r * 3 + 2] * ve[2];
  }
  _Pragma("unroll") for (int k = 0; k < 27; k++) acc[k] = ba_wave_sum(acc[k]);
  if (lane == 0) {
    const double AS1* U = v.cam_U + 36 * j;
    int q = 0;
    _Pragma("unroll") for (int r = 0; r < 6; r++)
      for (int c = 0; c <= r; c++) {
        double u = U[r * 6 + c];
        if (r == c) u *= (1.0 + lambda);
        const double val = u - acc[q++];
        v.S[(size_t)(row + r) * nS + row + c] = val; v.S[(size_t)(row + c) * nS + row + r] = val;   // mirrored :431-434
      }
    _Pragma("unroll") for (int r = 0; r < 6; r++) v.E[row + r] = v.cam_ea[6 * j + r] - acc[21 + r];
  }
}

// S off-diagonal block of one pair of adjustable cameras (:400-426); called by one wavefront.
BA_PHASE_FN void ba_task_pair(const BaView& v_, int task, int np, int nS, double lambda) {
  const BaViewG v = ba_g(v_);
  const int lane = threadIdx.x & 63;
  int t = task, fj = 1;
  while (t >= fj) { t -= fj; fj++; }                         // pair (fj > fk): free-camera ordinals
  const int fk = t;
  const int j = v.free_cams[fj], k = v.free_cams[fk];
  const int jrow = v.cam_row[j], krow = v.cam_row[k];
  double acc[36];
  _Pragma("unroll") for (int q = 0; q < 36; q++) acc[q] = 0.0;
  for (int p = lane; p < np; p += 64) {
    MeasState mj, mk;
    ba_load_state(v, ba_slot_of(v, p, fj), mj);
    ba_load_state(v, ba_slot_of(v, p, fk), mk);
    if (mj.st != MS_OK || mk.st != MS_OK) continue;
    double Vi[9], Y[18];
    ba_vstar_inv(v, p, lambda, Vi);
    {
      double Wj[18];
      ba_jac_W(mj, v.cam_pose[j].R, Wj);
      _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) Y[r * 3 + c] = Wj[r * 3] * Vi[c] + Wj[r * 3 + 1] * Vi[3 + c] + Wj[r * 3 + 2] * Vi[6 + c];
    }
    double Wk[18];
    ba_jac_W(mk, v.cam_pose[k].R, Wk);
    _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) acc[r * 6 + c] += Y[r * 3] * Wk[c * 3] + Y[r * 3 + 1] * Wk[c * 3 + 1] + Y[r * 3 + 2] * Wk[c * 3 + 2];
  }
  _Pragma("unroll") for (int q = 0; q < 36; q++) acc[q] = ba_wave_sum(acc[q]);
  if (lane == 0)
    _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
      v.S[(size_t)(jrow + r) * nS + krow + c] = -acc[r * 6 + c];
      v.S[(size_t)(krow + c) * nS + jrow + r] = -acc[r * 6 + c];
    }
}

// The whole reduced camera system of one LM trial, S = U* - sum_p Y_p W_p^T and E = ea - sum_p Y_p eb_p (jni/Bundle.cc:362-434),
// for problems with at most BA_MFMA_FREE adjustable cameras (BundleAdjustRecent: 5) as ONE fp64 matrix product on the
// matrix cores: per point p the 6 n_free x 3 stack Y_p (= W_j V*^-1 of every adjustable camera that measures p, zero rows
// otherwise, plus one row V*^-1 eb_p) times the 3 x 6 n_free stack W_p^T.  One lane per (point, adjustable camera) derives W and Y
// once, BA_MFMA_PPC points per wavefront and trip -- their F slots are contiguous, so the operands are read coalesced --
// staged k-major in LDS and multiplied by v_mfma_f64_16x16x4_f64
// (operand layout: tools/probes/mfma_f64_layout.hip; a = A[l%16][l/16], b = B[l/16][l%16], d[v] = D[l/16+4v][l%16]).
// Wave partials are added in wave order, the lower triangle is mirrored as the reference mirrors it (:431-434).
#ifndef BA_MFMA_PPC
#define BA_MFMA_PPC 12                                  // points per wavefront and trip: 12 x 5 cameras = 60 lanes, K = 36
#endif
#define BA_MFMA_K (3 * BA_MFMA_PPC)
#define BA_MFMA_STAGE (4 * BA_MFMA_K * 16)              // doubles per wavefront: Y rows 0-15 / 16-31, W columns 0-15 / 16-31, each [K][16]
static_assert(BA_MFMA_STAGE >= 32 * 32 && BA_MFMA_K % 4 == 0, "a wavefront's staging area also holds its 32 x 32 partial product");
typedef double ba_v4d __attribute__((ext_vector_type(4)));
// One pass: the block of S whose rows belong to the adjustable cameras of ordinals [r0, r0 + nr) and whose columns to those of
// [c0, c0 + ncl), nr, ncl <= BA_MFMA_FREE.  r0 == c0 (a diagonal block; then nr == ncl): lower triangle + mirror, U* on the cameras'
// own 6 x 6 blocks, and E of these cameras out of row 30.  r0 != c0 (an off-diagonal block of a problem with up to 2 * BA_MFMA_FREE
// adjustable cameras, e.g. BASELINE configs[3]'s 10-keyframe window): the whole 32 x 32 product, written with its transpose.
template <bool DIAG>
BA_PHASE_FN void ba_schur_mfma(const BaView& v_, int np, int nS, double lambda, double* lds_, int r0, int nr, int c0, int ncl) {
  const BaViewG v = ba_g(v_);
  double AS3* lds = (double AS3*)lds_;                    // the staging buffer is LDS: ds_read / ds_write, not flat
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double AS3* Y0 = lds + wave * BA_MFMA_STAGE; double AS3* Y1 = Y0 + BA_MFMA_K * 16; double AS3* W0 = Y1 + BA_MFMA_K * 16; double AS3* W1 = W0 + BA_MFMA_K * 16;
  for (int t = lane; t < BA_MFMA_STAGE; t += 64) Y0[t] = 0.0;      // rows / columns no lane ever writes stay zero
  constexpr bool diag = DIAG;                            // r0 == c0
  const int ncam = diag ? nr : nr + ncl;                 // cameras a trip's lanes cover: a diagonal pass derives Y and W of a camera in one lane
  const int PPC = diag ? BA_MFMA_PPC : (64 / ncam < BA_MFMA_PPC ? 64 / ncam : BA_MFMA_PPC);   // points per wavefront and trip (a diagonal pass: 64 / 5 = 12 at least)
  const bool active = lane < PPC * ncam;
  const int pl = active ? lane / ncam : 0, fi = active ? lane - pl * ncam : 0;
  const bool isrow = diag || fi < nr, iscol = diag || fi >= nr;
  const int frow = fi, fcol = diag ? fi : fi - nr;        // position of the camera inside the row / column group
  const int f = isrow ? r0 + frow : c0 + fcol;            // its ordinal among the adjustable cameras
  const int j = v.free_cams[f];
  double Rj[9], tj[3];
  _Pragma("unroll") for (int k = 0; k < 9; k++) Rj[k] = v.cam_pose[j].R[k];
  _Pragma("unroll") for (int k = 0; k < 3; k++) tj[k] = v.cam_pose[j].t[k];
  // the state of a slot as this phase needs it: the stored weighted derivatives; v3Cam = camera j applied to the point (the same
  // expression, the same bits as in the sweep that produced d)
  auto load_state_x = [&](int slot, int pc, MeasState& m) {
    const int ic = slot < 0 ? 0 : slot;
    m.st = slot < 0 ? MS_ERASED : SL_STATE(v.sl_info[ic]);
    _Pragma("unroll") for (int q = 0; q < 4; q++) m.d[q] = SL(sl_d, q, ic);
    const double X[3] = {v.pt_pos[3 * pc], v.pt_pos[3 * pc + 1], v.pt_pos[3 * pc + 2]};
    m.cm[0] = tj[0] + (Rj[0] * X[0] + Rj[1] * X[1] + Rj[2] * X[2]);
    m.cm[1] = tj[1] + (Rj[3] * X[0] + Rj[4] * X[1] + Rj[5] * X[2]);
    m.cm[2] = tj[2] + (Rj[6] * X[0] + Rj[7] * X[1] + Rj[8] * X[2]);
  };
  ba_v4d d00 = {0, 0, 0, 0}, d10 = {0, 0, 0, 0}, d11 = {0, 0, 0, 0}, d01 = {0, 0, 0, 0};
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  // software pipeline over the trips: the operands of trip t+1 are in flight while trip t is derived, staged and multiplied
  const int STRIDE = BA_WAVES * PPC;
  const int pfirst = wave * PPC;
  const int ksteps = diag ? BA_MFMA_K / 4 : (3 * PPC + 3) / 4;
  // A slot's index hangs on two words of its point (mask and offset of its F slots), its state on the index: two dependent round
  // trips.  The pipeline is two deep accordingly: the point words of trip t+2 and the slot state of trip t+1 are requested in trip t
  // (the index of t+1 is formed from the words requested a trip earlier), so that no request waits for one issued in the same trip.
  auto slot_words = [&](int p0, unsigned long long& mk, int& off) {
    const int p = p0 + pl, pc = p < np ? p : np - 1;
    mk = v.pt_maskF[pc]; off = v.pt_offF[pc];
  };
  auto slot_from = [&](int p0, unsigned long long mk, int off) -> int {
    const int p = p0 + pl;
    if (!(active && p < np) || !((mk >> f) & 1ull)) return -1;
    return off + __popcll(mk & ((1ull << f) - 1ull));                  // ba_slot_of
  };
  MeasState ms_n; double V_n[6], eb_n[3];
  unsigned long long mk_nn; int off_nn;
  {
    const int pc = pfirst + pl < np ? pfirst + pl : np - 1;
    unsigned long long mk0; int off0;
    slot_words(pfirst, mk0, off0);
    slot_words(pfirst + STRIDE, mk_nn, off_nn);
    load_state_x(slot_from(pfirst, mk0, off0), pc, ms_n);
    _Pragma("unroll") for (int k = 0; k < 6; k++) V_n[k] = PT(pt_V, k, pc);
    _Pragma("unroll") for (int k = 0; k < 3; k++) eb_n[k] = PT(pt_eb, k, pc);
  }
  for (int p0 = pfirst; p0 < np; p0 += STRIDE) {
    const int p = p0 + pl;
    const MeasState ms = ms_n;
    double Vl[6], eb[3];
    _Pragma("unroll") for (int k = 0; k < 6; k++) Vl[k] = V_n[k];
    _Pragma("unroll") for (int k = 0; k < 3; k++) eb[k] = eb_n[k];
    {
      const int pn = p0 + STRIDE + pl, pc = pn < np ? pn : np - 1;
      const int slot_n = slot_from(p0 + STRIDE, mk_nn, off_nn);
      load_state_x(slot_n, pc, ms_n);
      _Pragma("unroll") for (int k = 0; k < 6; k++) V_n[k] = PT(pt_V, k, pc);
      _Pragma("unroll") for (int k = 0; k < 3; k++) eb_n[k] = PT(pt_eb, k, pc);
      slot_words(p0 + 2 * STRIDE, mk_nn, off_nn);
    }
    double Vi[9];                                                       // V*^-1 (:329-347)
    if (Vl[0] * Vl[2] * Vl[5] == 0) { _Pragma("unroll") for (int k = 0; k < 9; k++) Vi[k] = 0.0; }
    else {
      const double Vs[9] = {Vl[0] * (1.0 + lambda), Vl[1], Vl[3], Vl[1], Vl[2] * (1.0 + lambda), Vl[4], Vl[3], Vl[4], Vl[5] * (1.0 + lambda)};
      inv3(Vs, Vi);
    }
    double W[18], Y[18];
    if (ms.st == MS_OK) {
      ba_jac_W(ms, Rj, W);
      _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) Y[r * 3 + c] = W[r * 3] * Vi[c] + W[r * 3 + 1] * Vi[3 + c] + W[r * 3 + 2] * Vi[6 + c];
    } else {
      _Pragma("unroll") for (int q = 0; q < 18; q++) { W[q] = 0.0; Y[q] = 0.0; }
    }
    if (active) {
      _Pragma("unroll") for (int r = 0; r < 6; r++) {
        if (isrow) {
          const int row = 6 * frow + r;
          double AS3* yd = (row < 16 ? Y0 + row : Y1 + (row - 16)) + 3 * pl * 16;
          _Pragma("unroll") for (int c = 0; c < 3; c++) yd[c * 16] = Y[r * 3 + c];
        }
        if (iscol) {
          const int col = 6 * fcol + r;
          double AS3* wd = (col < 16 ? W0 + col : W1 + (col - 16)) + 3 * pl * 16;
          _Pragma("unroll") for (int c = 0; c < 3; c++) wd[c * 16] = W[r * 3 + c];
        }
      }
      if (diag && fi == 0) {                                        // row 30 of the left operand: V*^-1 eb_p, so that D[30][.] = sum_p W (V*^-1 eb) (:388-396)
        _Pragma("unroll") for (int c = 0; c < 3; c++) Y1[(3 * pl + c) * 16 + 14] = p < np ? Vi[c * 3] * eb[0] + Vi[c * 3 + 1] * eb[1] + Vi[c * 3 + 2] * eb[2] : 0.0;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0) only: the prefetched global loads stay in flight
    __builtin_amdgcn_wave_barrier();
    auto mfma_step = [&](int ks) {
      const int o = (ks * 4 + (lane >> 4)) * 16 + (lane & 15);
      const double a0 = Y0[o], a1 = Y1[o], b0 = W0[o], b1 = W1[o];
      d00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d00, 0, 0, 0);
      d10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, d10, 0, 0, 0);
      d11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, d11, 0, 0, 0);
      if (!diag) d01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, d01, 0, 0, 0);   // a diagonal block's tile above the diagonal is not needed
    };
    if constexpr (diag) { _Pragma("unroll") for (int ks = 0; ks < BA_MFMA_K / 4; ks++) mfma_step(ks); }
    else { for (int ks = 0; ks < ksteps; ks++) mfma_step(ks); }
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0) only: the prefetched global loads stay in flight
    __builtin_amdgcn_wave_barrier();
  }
  // the wave's 32 x 32 partial product, row-major, into its own staging area
  __syncthreads();
  _Pragma("unroll") for (int q = 0; q < 4; q++) {
    const int r = (lane >> 4) + 4 * q, c = lane & 15;
    Y0[r * 32 + c] = d00[q]; Y0[(16 + r) * 32 + c] = d10[q]; Y0[(16 + r) * 32 + 16 + c] = d11[q]; Y0[r * 32 + 16 + c] = d01[q];
  }
  __syncthreads();
  const int nrow = 6 * nr, ncol = 6 * ncl;
  for (int t = threadIdx.x; t < 32 * 32; t += BA_THREADS) {
    const int r = t >> 5, c = t & 31;
    if (c >= ncol) continue;
    if (diag ? (!(r < nrow || r == 30) || (r < nrow && c > r)) : r >= nrow) continue;
    double sum = 0.0;
    for (int w = 0; w < BA_WAVES; w++) sum += lds[w * BA_MFMA_STAGE + t];
    if (r == 30) { const int fk = c0 + c / 6, kk = v.free_cams[fk]; v.E[6 * c0 + c] = v.cam_ea[6 * kk + (c % 6)] - sum; continue; }   // rows of S follow the adjustable cameras in order (cam_row == 6 * ordinal)
    const int R = 6 * r0 + r, C = 6 * c0 + c;
    double u = 0.0;
    if (diag && c / 6 == r / 6) { const int jj = v.free_cams[r0 + r / 6]; u = v.cam_U[36 * jj + (r % 6) * 6 + (c % 6)]; if (r == c) u *= (1.0 + lambda); }
    const double val = u - sum;
    v.S[(size_t)R * nS + C] = val; v.S[(size_t)C * nS + R] = val;
  }
  __syncthreads();
}

// map updates (jni/Bundle.cc:440-462, :484): trial point positions; returns this thread's share of |update|^2.
// Up to BA_MFMA_FREE adjustable cameras: the lane mapping of ba_schur_mfma (one lane per (point, camera), contiguous F slots);
// the first lane of a point adds the cameras' terms in camera order.
BA_PHASE_FN double ba_map_update(const BaView& v_, int nfree, int np, double lambda, double* lds_ /* LDS: staging [BA_WAVES][64][BA_SWEEP_STAGE], then poses and updates */) {
  const BaViewG v = ba_g(v_);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double ssq = 0.0;
  if (nfree > 0 && nfree <= BA_FAST_FREE) {
    // One lane per F slot, a chunk of whole points per wavefront trip (the step sweep's tables), the slot's state loaded a chunk
    // ahead; W^T * (camera update) of every slot goes to LDS and the first lane of each point adds its slots in order, then
    // solves for the point (:440-462, :484).  Poses and camera updates come out of LDS.
    double AS3* stg = (double AS3*)lds_ + wave * 64 * BA_SWEEP_STAGE;
    double AS3* camL = (double AS3*)lds_ + BA_WAVES * 64 * BA_SWEEP_STAGE;
    double AS3* cuL = camL + 12 * BA_MAX_CAMS_LDS;
    const int nc = v.res->n_cams;
    for (int t = threadIdx.x; t < nc * 12; t += BA_THREADS) camL[t] = ((const double AS1*)v.cam_pose)[t];
    for (int t = threadIdx.x; t < nfree * 6; t += BA_THREADS) cuL[t] = v.cam_up[v.cam_row[v.free_cams[t / 6]] + t % 6];
    __syncthreads();
    const int nch = v.ch_n[0];
    const int AS1* ch = v.chF;
    struct SIn { int info, pt; double d[4]; };
    auto load_in = [&](int a, int n, SIn& r) {
      const int s = a + (lane < n ? lane : 0);
      r.info = v.sl_info[s]; r.pt = v.sl_pt[s];
      _Pragma("unroll") for (int q = 0; q < 4; q++) r.d[q] = SL(sl_d, q, s);
    };
    int cs0 = 0, cs1 = 0;
    SIn cin; cin.info = 0; cin.pt = 0;
    _Pragma("unroll") for (int q = 0; q < 4; q++) cin.d[q] = 0;
    double cX[3] = {0, 0, 0};                                          // the slot's point: v3Cam is re-derived from it, and the point's first lane updates it
    if (wave < nch) {
      cs0 = ch[wave]; cs1 = ch[wave + 1]; load_in(cs0, cs1 - cs0, cin);
      _Pragma("unroll") for (int q = 0; q < 3; q++) cX[q] = v.pt_pos[3 * cin.pt + q];
    }
    for (int k = wave; k < nch; k += BA_WAVES) {
      const int n = cs1 - cs0;                                         // <= 64: a point has at most BA_MFMA_FREE slots in region F
      const int kn = k + BA_WAVES;
      int ns0 = 0, ns1 = 0;
      if (kn < nch) { ns0 = ch[kn]; ns1 = ch[kn + 1]; }
      SIn nin = cin;
      const bool act = lane < n;
      const int pt = cin.pt;
      // the point's V and epsilon_b for its first lane: requested by every lane of the point (one broadcast access) at the top of
      // the trip, so that they are under way while the slot's W^T * update is formed
      double eb[3], vq[6];
      _Pragma("unroll") for (int q = 0; q < 3; q++) eb[q] = PT(pt_eb, q, pt);
      _Pragma("unroll") for (int q = 0; q < 6; q++) vq[q] = PT(pt_V, q, pt);
      double t[3] = {0, 0, 0};
      if (act && SL_STATE(cin.info) == MS_OK) {
        MeasState ms;
        _Pragma("unroll") for (int q = 0; q < 4; q++) ms.d[q] = cin.d[q];
        ms.st = MS_OK;
        const int cam = SL_CAM(cin.info), f = SL_FORD(cin.info);
        double Rj[9], cu[6], W[18];
        _Pragma("unroll") for (int q = 0; q < 9; q++) Rj[q] = camL[cam * 12 + q];
        _Pragma("unroll") for (int q = 0; q < 3; q++) ms.cm[q] = camL[cam * 12 + 9 + q] + (Rj[3 * q] * cX[0] + Rj[3 * q + 1] * cX[1] + Rj[3 * q + 2] * cX[2]);   // pose_xform
        _Pragma("unroll") for (int q = 0; q < 6; q++) cu[q] = cuL[f * 6 + q];
        ba_jac_W(ms, Rj, W);
        _Pragma("unroll") for (int c = 0; c < 3; c++) { double sx = 0; for (int r = 0; r < 6; r++) sx += W[r * 3 + c] * cu[r]; t[c] = sx; }
      }
      _Pragma("unroll") for (int c = 0; c < 3; c++) stg[lane * 3 + c] = t[c];
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      if (kn < nch) load_in(ns0, ns1 - ns0, nin);
      const int ptprev = __shfl_up(pt, 1);
      const bool leader = act && (lane == 0 || ptprev != pt);
      const unsigned long long lm = __ballot(leader);
      if (leader) {
        const unsigned long long above = lane < 63 ? lm >> (lane + 1) : 0ull;
        const int cnt = above ? (int)__ffsll((long long)above) : n - lane;
        double sum[3] = {0, 0, 0};
        for (int j = 0; j < cnt; j++) { _Pragma("unroll") for (int c = 0; c < 3; c++) sum[c] += stg[(lane + j) * 3 + c]; }
        double Vi[9];
        if (vq[0] * vq[2] * vq[5] == 0) { _Pragma("unroll") for (int q = 0; q < 9; q++) Vi[q] = 0.0; }   // ba_vstar_inv on the values in hand
        else {
          const double Vs[9] = {vq[0] * (1.0 + lambda), vq[1], vq[3], vq[1], vq[2] * (1.0 + lambda), vq[4], vq[3], vq[4], vq[5] * (1.0 + lambda)};
          inv3(Vs, Vi);
        }
        const double x[3] = {eb[0] - sum[0], eb[1] - sum[1], eb[2] - sum[2]};
        _Pragma("unroll") for (int r = 0; r < 3; r++) {
          const double u = Vi[r * 3] * x[0] + Vi[r * 3 + 1] * x[1] + Vi[r * 3 + 2] * x[2];
          ssq += u * u;
          v.pt_new[3 * pt + r] = cX[r] + u;                            // :484
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (kn < nch) { _Pragma("unroll") for (int q = 0; q < 3; q++) cX[q] = v.pt_pos[3 * nin.pt + q]; }
      cs0 = ns0; cs1 = ns1; cin = nin;
    }
    for (int p = threadIdx.x; p < np; p += BA_THREADS) {               // points no adjustable camera measures: the update is V*^-1 epsilon_b
      if (v.pt_offF[p + 1] - v.pt_offF[p] > 0) continue;
      const double eb[3] = {PT(pt_eb, 0, p), PT(pt_eb, 1, p), PT(pt_eb, 2, p)};
      double Vi[9];
      ba_vstar_inv(v, p, lambda, Vi);
      _Pragma("unroll") for (int r = 0; r < 3; r++) {
        const double u = Vi[r * 3] * eb[0] + Vi[r * 3 + 1] * eb[1] + Vi[r * 3 + 2] * eb[2];
        ssq += u * u;
        v.pt_new[3 * p + r] = v.pt_pos[3 * p + r] + u;
      }
    }
    return ssq;
  }
  for (int p = threadIdx.x; p < np; p += BA_THREADS) {               // any number of adjustable cameras: one lane per point, its F slots in order
    double sum[3] = {0, 0, 0};
    const int s0 = v.pt_offF[p], s1 = v.pt_offF[p + 1];
    for (int s = s0; s < s1; s++) {
      MeasState ms;
      ba_load_state(v, s, ms);
      if (ms.st != MS_OK) continue;
      const int jj = SL_CAM(v.sl_info[s]);
      double W[18];
      ba_jac_W(ms, v.cam_pose[jj].R, W);
      const double AS1* cu = v.cam_up + v.cam_row[jj];
      _Pragma("unroll") for (int c = 0; c < 3; c++) { double sx = 0; for (int r = 0; r < 6; r++) sx += W[r * 3 + c] * cu[r]; sum[c] += sx; }
    }
    const double eb[3] = {PT(pt_eb, 0, p), PT(pt_eb, 1, p), PT(pt_eb, 2, p)};
    double Vi[9];
    ba_vstar_inv(v, p, lambda, Vi);
    const double x[3] = {eb[0] - sum[0], eb[1] - sum[1], eb[2] - sum[2]};
    _Pragma("unroll") for (int r = 0; r < 3; r++) {
      const double u = Vi[r * 3] * x[0] + Vi[r * 3 + 1] * x[1] + Vi[r * 3 + 2] * x[2];
      ssq += u * u;
      v.pt_new[3 * p + r] = v.pt_pos[3 * p + r] + u;
    }
  }
  return ssq;
}

// Erase the outliers of this step in LIST order (jni/Bundle.cc:517-528).  The slots are point-major, the reference's list is
// whatever order AddMeas was called in: the bad slots mark their list index in an LDS bit map, then the map is expanded in
// index order (popcount scan) into the (p, c) pairs.  Returns the new total of outlier measurements.
BA_PHASE_FN int ba_erase_outliers(const BaView& v_, int M, int nm, int nout, unsigned* bits /* LDS [(max_meas + 31) / 32] */, int* ired) {
  const BaViewG v = ba_g(v_);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwords = (nm + 31) >> 5;
  for (int t = threadIdx.x; t < nwords; t += BA_THREADS) bits[t] = 0u;
  __syncthreads();
  int mine = 0;
  constexpr int EI = 8;                                               // slots whose state words are requested together: few slots are bad, the loop is all latency
  for (int s0 = threadIdx.x; s0 < M; s0 += EI * BA_THREADS) {
    int info[EI];
    _Pragma("unroll") for (int u = 0; u < EI; u++) { const int s = s0 + u * BA_THREADS; info[u] = s < M ? v.sl_info[s] : 0; }
    _Pragma("unroll") for (int u = 0; u < EI; u++) {
      const int s = s0 + u * BA_THREADS;
      if (s >= M || SL_STATE(info[u]) != MS_BAD) continue;
      v.sl_info[s] = SL_WITH_STATE(info[u], MS_ERASED);
      v.scratch[s] = __builtin_huge_val();
      const int i = v.sl_logical[s];
      atomicOr(&bits[i >> 5], 1u << (i & 31));
      atomicAdd((int*)&v.pt_nout[v.sl_pt[s]], 1);
      mine++;
    }
  }
  const int total = ba_block_sum_i(mine, ired);                     // (barriers inside: the bit map is complete)
  if (total == 0) return nout;
  int base = nout;
  for (int w0 = 0; w0 < nwords; w0 += BA_THREADS) {
    const int w = w0 + threadIdx.x;
    unsigned m = w < nwords ? bits[w] : 0u;
    const int cnt = __popc(m);
    int inc = cnt;
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    __syncthreads();
    if (lane == 63) ired[wave] = inc;
    __syncthreads();
    int off = base + inc - cnt;
    for (int ww = 0; ww < BA_WAVES; ww++) { if (ww < wave) off += ired[ww]; base += ired[ww]; }
    while (m) {
      const int b = __ffs((int)m) - 1;
      m &= m - 1;
      const int i = (w << 5) + b;
      v.outl[2 * off] = v.ms_p[i]; v.outl[2 * off + 1] = v.ms_c[i];
      off++;
    }
  }
  __syncthreads();
  return base;
}

// Bundle::Compute.  Called by all BA_THREADS threads of one workgroup.
DEVFN void ba_compute(const BaView& v_, const BaConfig& cfg) {
  const BaViewG v = ba_g(v_);
  __shared__ double red[BA_WAVES];
  __shared__ int ired[BA_WAVES];
  __shared__ int hist[768];
  __shared__ unsigned long long sel[1];
  __shared__ double sh_lambda, sh_factor, sh_sigma2, sh_cur_err, sh_new_err;
  constexpr int LDS_MFMA = BA_WAVES * BA_MFMA_STAGE, LDS_SOLVE = BA_LDS_N * (BA_LDS_N + 1);
  constexpr int LDS_SWEEP = BA_WAVES * 64 * BA_SWEEP_STAGE + 12 * BA_MAX_CAMS_LDS + 6 * BA_FAST_FREE;
  constexpr int LDS_LAYOUT = (2 * 4097 * (int)sizeof(int) + 7) / 8;
  constexpr int LDS_A = LDS_MFMA > LDS_SOLVE ? LDS_MFMA : LDS_SOLVE, LDS_B = LDS_SWEEP > LDS_LAYOUT ? LDS_SWEEP : LDS_LAYOUT;
  __shared__ double lds_buf[LDS_A > LDS_B ? LDS_A : LDS_B];
  double* lds_A = lds_buf;
  __shared__ int sh_converged, sh_hitmax, sh_counter, sh_accepted, sh_error, sh_nout, sh_cache_valid, sh_next_nvalid;
  BaResult AS1* R = v.res;
  const int nc = R->n_cams, np = R->n_pts, nm = R->n_meas;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    int nf = 0, row = 0;
    for (int j = 0; j < nc; j++) {            // AddCamera start rows, jni/Bundle.cc:79-85
      if (!v.cam_fixed[j]) { v.cam_row[j] = row; row += 6; v.free_cams[nf++] = j; } else v.cam_row[j] = -999999999;
    }
    R->n_free = nf;
    sh_lambda = 0.0001; sh_factor = 2.0;      // :144-145
    sh_converged = 0; sh_hitmax = 0; sh_counter = 0; sh_accepted = 0; sh_error = 0; sh_nout = 0; sh_sigma2 = 0; sh_cache_valid = 0;
    R->trials = 0;
  }
  __syncthreads();
  const int nfree = R->n_free, nS = nfree * 6;
  if (nfree > 64) { if (threadIdx.x == 0) { R->accepted = -1; R->converged = 0; R->hit_max = 0; R->counter = 0; R->n_outlier_meas = 0; } __syncthreads(); return; }   // pt_maskF holds 64 ordinals
#ifdef VSLAM_BA_PROF
  unsigned long long ba_t0 = clock64();
#endif
  static_assert(sizeof(lds_buf) >= 2 * 4097 * sizeof(int) && sizeof(lds_buf) >= (65536 / 32) * sizeof(unsigned) &&
                sizeof(lds_buf) >= (size_t)LDS_SWEEP * sizeof(double), "the LDS buffer serves the layout, the step sweep and the erase");
  ba_build_layout(v_, nc, np, ired, (int*)lds_buf);
  const int M = v.ch_n[3];                                            // slots = measurements of the list
  BA_STAMP(0);

  while (!sh_converged && !sh_hitmax && !sh_error) {             // :153 (no abort signal: the map-maker runs synchronously)
    // ================= Do_LM_Step =================
    // pass 1 (:209-215): the squared errors of the measurements still in the list, for the median
    const bool cached = sh_cache_valid != 0;                       // the previous step was accepted: FindNewError has left them
    int nvalid;
    if (cached) nvalid = sh_next_nvalid;
    else nvalid = ba_block_sum_i(ba_find_new_error(v_, cfg, M, 1.0, 0, lds_buf).nvalid, ired);
    BA_STAMP(1);
    if (nvalid == 0) { if (threadIdx.x == 0) sh_error = 1; __syncthreads(); break; }
    {                                                              // :220-227 Tukey sigma, clamped
      const double med = M > 4096 ? block_radix_select<16>(v.scratch, M, nvalid / 2, hist, sel)   // (big problems: 16 values per lane in flight, 64 per lane and sweep)
                                  : block_radix_select<8>(v.scratch, M, nvalid / 2, hist, sel);
      double s2 = tukey_sigma_squared(med, (unsigned long)nvalid);
      if (s2 < cfg.min_sigma2) s2 = cfg.min_sigma2;
      if (threadIdx.x == 0) sh_sigma2 = s2;
      __syncthreads();
    }
    const double sigma2 = sh_sigma2;
    BA_STAMP(2);
    // passes 1 + 2 (:209-321) in one sweep: weights, objective, V / epsilon_b, U / epsilon_a; A, B, W are re-derived by their consumers
    double* stg = lds_buf; double* ured = nullptr; double* camsL = lds_buf + BA_WAVES * 64 * BA_SWEEP_STAGE;
    for (int t = threadIdx.x; t < nc * 12; t += BA_THREADS) camsL[t] = ((const double AS1*)v.cam_pose)[t];   // Pose = R[9], t[3]
    __syncthreads();
    double cur = ba_step_sweep(v_, cfg, 0, sigma2, nfree, stg, ured, camsL);
    __syncthreads();
    if (nfree <= BA_FAST_FREE) {
      for (int t = threadIdx.x; t < nfree * 27; t += BA_THREADS) {     // wave partials in wave order
        const int f = t / 27, q = t - 27 * f, j = v.free_cams[f];
        double x = 0.0;
        for (int w = 0; w < BA_WAVES; w++) x += lds_buf[w * 64 * BA_SWEEP_STAGE + f * 27 + q];
        if (q < 21) { int r = 0, qq = q; while (qq > r) { qq -= r + 1; r++; } v.cam_U[36 * j + r * 6 + qq] = x; }
        else v.cam_ea[6 * j + (q - 21)] = x;
      }
      __syncthreads();                                                 // the sums sit in the staging areas the next sweep writes
    }
    BA_STAMP(3);
    cur += ba_step_sweep(v_, cfg, 1, sigma2, nfree, stg, ured, camsL);
    cur = ba_block_sum(cur, red);
    if (threadIdx.x == 0) sh_cur_err = cur;
    __syncthreads();
    BA_STAMP(4);
    if (nfree > BA_FAST_FREE) { ba_accum_U_generic(v_, nfree, np); __syncthreads(); }
    BA_STAMP(5);
    // ---- inner loop over lambda (:326-501) ----
    if (threadIdx.x == 0) sh_new_err = sh_cur_err + 9999;
    __syncthreads();
    while (sh_new_err > sh_cur_err && !sh_converged && !sh_hitmax && !sh_error) {
      const double lambda = sh_lambda;
      BA_STAMP(6);
      // S: diagonal blocks + E (:362-396) and off-diagonal blocks (:400-426); V*^-1 (:329-347) is formed where it is used
      if (nfree == 0) { }                                              // only fixed cameras: no camera unknowns, the points move alone
      else if (nfree <= BA_MFMA_FREE) ba_schur_mfma<true>(v_, np, nS, lambda, lds_buf, 0, nfree, 0, nfree);
      else if (nfree <= 2 * BA_MFMA_FREE) {                           // two groups of cameras: two diagonal blocks and the block between them
        const int g0 = (nfree + 1) / 2, g1 = nfree - g0;
        ba_schur_mfma<true>(v_, np, nS, lambda, lds_buf, 0, g0, 0, g0);
        ba_schur_mfma<true>(v_, np, nS, lambda, lds_buf, g0, g1, g0, g1);
        ba_schur_mfma<false>(v_, np, nS, lambda, lds_buf, g0, g1, 0, g0);
      }
      else {
        for (int t = threadIdx.x; t < nS * nS; t += BA_THREADS) v.S[t] = 0.0;
        __syncthreads();
        const int ntask = nfree + nfree * (nfree - 1) / 2;
        for (int task = wave; task < ntask; task += BA_WAVES) {
          if (task < nfree) ba_task_diag(v_, task, np, nS, lambda);
          else ba_task_pair(v_, task - nfree, np, nS, lambda);
        }
      }
      __syncthreads();
      BA_STAMP(7);
      bool solved = true;
      if (nS > 0 && nS <= BA_WSOLVE_N && VSLAM_BA_WSOLVE) {           // one wavefront, registers (the BundleAdjustRecent size)
        if (wave == 0) { const bool okw = ba_solve_wave(v_, nS); if (lane == 0) ired[0] = okw ? 1 : 0; }
        __syncthreads();
        solved = ired[0] != 0;
        __syncthreads();
      } else if (nS > 0) solved = nS <= BA_LDS_N ? ba_block_solve_lds((const double*)v.S, (double*)v.E, nS, lds_A, ired) : ba_block_solve((double*)v.S, (double*)v.E, nS, ired);
      if (!solved) { if (threadIdx.x == 0) sh_error = 1; __syncthreads(); break; }
      for (int t = threadIdx.x; t < nS; t += BA_THREADS) v.cam_up[t] = v.E[t];
      __syncthreads();
      BA_STAMP(8);
      // map updates (:440-462)
      double ssq = ba_map_update(v_, nfree, np, lambda, lds_buf);
      for (int t = threadIdx.x; t < nS; t += BA_THREADS) ssq += v.cam_up[t] * v.cam_up[t];
      ssq = ba_block_sum(ssq, red);                                    // :467-470
      for (int j = threadIdx.x; j < nc; j += BA_THREADS) {             // :476-482
        const Pose Tj = ba_load_pose(v.cam_pose + j);
        if (v.cam_fixed[j]) ba_store_pose(v.cam_new + j, Tj);
        else {
          double mu[6];
          _Pragma("unroll") for (int k = 0; k < 6; k++) mu[k] = v.cam_up[v.cam_row[j] + k];
          ba_store_pose(v.cam_new + j, pose_mul(se3_exp(mu), Tj));
        }
      }
      __syncthreads();
      BA_STAMP(9);
      // FindNewError (:537-561)
      const BaNewError fne = ba_find_new_error(v_, cfg, M, sigma2, 1, lds_buf);
      const double ne = ba_block_sum(fne.ne, red);
      const int nv_next = ba_block_sum_i(fne.nvalid, ired);
      BA_STAMP(10);
      if (threadIdx.x == 0) {
        sh_next_nvalid = nv_next;
        if (ssq < cfg.convergence_limit) sh_converged = 1;
        sh_new_err = ne;
        if (ne > sh_cur_err) { sh_lambda = sh_lambda * sh_factor; sh_factor = sh_factor * 2; }   // ModifyLambda_BadStep :614-617
        sh_counter++; R->trials++;
        if (sh_counter >= cfg.max_iterations) sh_hitmax = 1;           // :498-500
      }
      __syncthreads();
    }
    if (sh_error) break;
    if (sh_new_err < sh_cur_err) {                                     // :503-514
      for (int j = threadIdx.x; j < nc; j += BA_THREADS) ba_store_pose(v.cam_pose + j, ba_load_pose(v.cam_new + j));
      for (int t = threadIdx.x; t < 3 * np; t += BA_THREADS) v.pt_pos[t] = v.pt_new[t];
      if (threadIdx.x == 0) { sh_factor = 2.0; sh_lambda *= 0.3; sh_accepted++; sh_cache_valid = 1; }   // ModifyLambda_GoodStep :609-612
    } else if (threadIdx.x == 0) sh_cache_valid = 0;
    __syncthreads();
    BA_STAMP(11);
    {                                                                  // erase the outliers in list order (:517-528)
      const int no = ba_erase_outliers(v_, M, nm, sh_nout, (unsigned*)lds_buf, ired);
      __syncthreads();
      if (threadIdx.x == 0) sh_nout = no;
      __syncthreads();
    }
    BA_STAMP(12);
  }
  if (threadIdx.x == 0) {
    R->accepted = sh_error ? -1 : sh_accepted;                         // :170-177
    R->converged = sh_converged; R->hit_max = sh_hitmax; R->counter = sh_counter;
    R->sigma2 = sh_sigma2; R->lambda = sh_lambda; R->lambda_factor = sh_factor; R->n_outlier_meas = sh_nout;
  }
  __syncthreads();
}
