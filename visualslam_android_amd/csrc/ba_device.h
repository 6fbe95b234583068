// Bundle::Compute (jni/Bundle.cc:136-178) / Do_LM_Step (:202-532) as ONE persistent workgroup per problem:
// the whole Levenberg-Marquardt loop runs inside one launch, no host round-trip.
//
// Data layout (per problem, struct-of-arrays in HBM, fp64):
//   cameras: pose, trial pose, fixed flag, start row, U (6x6 lower), epsilon_a
//   points : position, trial position, V (3x3 lower), epsilon_b, V*^-1
//   measurements in AddMeas order: (p, c), found, sqrt-inv-noise, state, v3Cam, weighted camera derivatives, epsilon.
//   The Jacobians A (2x6), B (2x3) and W = A^T B (6x3) are NOT stored: every consumer re-derives them from v3Cam, the
//   weighted derivatives and the camera rotation (about 150 flops instead of 288 B written and up to 1 KB re-read per
//   measurement and LM trial).  At 256-512 concurrent problems the kernel is bound by dependent-load latency and fp64
//   issue at 2 waves per SIMD, not by HBM or flops (profiles/README.md): hence the batched loads and the per-phase
//   functions below.
//   lut[c][p] -> measurement index (GenerateMeasLUTs :566-575)
// Reductions are deterministic: "segmented" per-camera / per-camera-pair sums are taken by one wavefront each
// (lanes stride over the points, then __shfl_xor butterflies), per-point sums by one lane in camera order.
#pragma once
#include "dev_math.h"

#define BA_THREADS 256  // 4 waves; with amdgpu_waves_per_eu(2,2) on the kernel two problems share a CU (measured: 512 x 1 -4 %, 128 x 4 -4 %)
#define BA_LDS_N 60      // reduced camera systems up to 60 x 60 (10 adjustable cameras) are solved in LDS
#define BA_WAVES (BA_THREADS / 64)
#define BA_ILP_PROJ 4   // projection passes: 4 measurements in flight (1 -> 4: -31 % on FindNewError once the view pointers were global and scalar; 6 spills: 5x slower)
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
  double* pt_pos; double* pt_new; double* pt_V; double* pt_eb; double* pt_Vinv; int* pt_nmeas; int* pt_nout;
  int* ms_p; int* ms_c; int* ms_state; double* ms_found; double* ms_sin; double* ms_cam; double* ms_eps; double* ms_err2;
  double* ms_derivs;
  double* ms_tcam; double* ms_tfac; double* ms_teps;   // FindNewError's projection of the trial state, reused by pass 1 after an accepted step
  int* lut;              // [max_cams][max_pts]
  double* S; double* E; double* cam_up; double* map_up;
  double* scratch;       // [max_meas]
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
  double AS1* pt_pos; double AS1* pt_new; double AS1* pt_V; double AS1* pt_eb; double AS1* pt_Vinv; int AS1* pt_nmeas; int AS1* pt_nout;
  int AS1* ms_p; int AS1* ms_c; int AS1* ms_state; double AS1* ms_found; double AS1* ms_sin; double AS1* ms_cam; double AS1* ms_eps; double AS1* ms_err2;
  double AS1* ms_derivs;
  double AS1* ms_tcam; double AS1* ms_tfac; double AS1* ms_teps;
  int AS1* lut;
  double AS1* S; double AS1* E; double AS1* cam_up; double AS1* map_up;
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
  BA_G(pt_pos); BA_G(pt_new); BA_G(pt_V); BA_G(pt_eb); BA_G(pt_Vinv); BA_G(pt_nmeas); BA_G(pt_nout);
  BA_G(ms_p); BA_G(ms_c); BA_G(ms_state); BA_G(ms_found); BA_G(ms_sin); BA_G(ms_cam); BA_G(ms_eps); BA_G(ms_err2); BA_G(ms_derivs);
  BA_G(ms_tcam); BA_G(ms_tfac); BA_G(ms_teps); BA_G(lut); BA_G(S); BA_G(E); BA_G(cam_up); BA_G(map_up); BA_G(scratch); BA_G(outl); BA_G(free_cams);
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

struct BaConfig { CamModel cam; int max_iterations; double convergence_limit, min_sigma2; };

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

// ProjectAndFindSquaredError, jni/Bundle.cc:181-199, on operands already in registers (camera pose T, point X, found
// position f, sqrt-inv-noise sn).  Stores v3Cam, state, derivatives, epsilon, error^2 of measurement i.
DEVFN int ba_project_meas(const BaViewG& v, const BaConfig& cfg, int i, const Pose& T, const double X[3], double f0, double f1, double sn, double& e2) {
  double c[3];
  pose_xform(T, X, c);
  MS(ms_cam, 0, i) = c[0]; MS(ms_cam, 1, i) = c[1]; MS(ms_cam, 2, i) = c[2];
  if (c[2] <= 0) { v.ms_state[i] = MS_BAD; return MS_BAD; }
  v.ms_state[i] = MS_OK;
  const CamProj pr = cam_project(cfg.cam, c[0] / c[2], c[1] / c[2]);
  double dd[4];
  cam_derivs(cfg.cam, pr, dd);
  MS(ms_derivs, 0, i) = dd[0]; MS(ms_derivs, 1, i) = dd[1]; MS(ms_derivs, 2, i) = dd[2]; MS(ms_derivs, 3, i) = dd[3];
  const double e0 = (f0 - pr.im[0]) * sn, e1 = (f1 - pr.im[1]) * sn;
  MS(ms_eps, 0, i) = e0; MS(ms_eps, 1, i) = e1;
  e2 = e0 * e0 + e1 * e1;
  v.ms_err2[i] = e2;
  return MS_OK;
}

// Jacobians of one measurement from its stored state (jni/Bundle.cc:262-300): cm = v3Cam, d = sqrt-inv-noise * weight *
// camera derivatives (2x2 row-major), R = rotation of the camera.  Same expressions wherever they are re-derived.
// The loops over measurements are memory-latency bound (one workgroup per problem, 2 waves per SIMD): each thread
// first loads the operands of BA_ILP (or 2) independent measurements unconditionally -- index clamped, no branch
// between the loads -- and only then computes, so the round trips overlap.  Accumulation order is unchanged.
struct MeasState { double cm[3], d[4]; int st; };
DEVFN void ba_load_state(const BaViewG& v, int i, MeasState& m) {      // i < 0: loads measurement 0, state forced to erased
  const int ic = i < 0 ? 0 : i;
  m.st = v.ms_state[ic];
  m.cm[0] = MS(ms_cam, 0, ic); m.cm[1] = MS(ms_cam, 1, ic); m.cm[2] = MS(ms_cam, 2, ic);
  m.d[0] = MS(ms_derivs, 0, ic); m.d[1] = MS(ms_derivs, 1, ic); m.d[2] = MS(ms_derivs, 2, ic); m.d[3] = MS(ms_derivs, 3, ic);
  if (i < 0) m.st = MS_ERASED;
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
DEVFN double ba_readlane_d(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __attribute__((noinline)) bool ba_solve_wave(const BaView& v_, int n) {
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

// pass 1 of Do_LM_Step (jni/Bundle.cc:209-215): project every measurement still in the list; returns this thread's
// count of valid ones.  Kept out of line (like FindNewError below): the fp64 atan / division sequences of the camera model
// get their own register allocation instead of inheriting the pressure of the Schur-complement tasks.
__device__ __attribute__((noinline)) int ba_pass1_project(const BaView& v_, const BaConfig& cfg_, int nm) {
  const BaViewG v = ba_g(v_);
  const BaConfig cfg = cfg_;
  int nvalid = 0;
  for (int i0 = threadIdx.x; i0 < nm; i0 += BA_ILP_PROJ * BA_THREADS) {
    int st[BA_ILP_PROJ], mc[BA_ILP_PROJ], mp[BA_ILP_PROJ]; double f0[BA_ILP_PROJ], f1[BA_ILP_PROJ], sn[BA_ILP_PROJ];
    _Pragma("unroll") for (int u = 0; u < BA_ILP_PROJ; u++) {
      const int i = i0 + u * BA_THREADS, ic = i < nm ? i : nm - 1;
      st[u] = v.ms_state[ic]; mc[u] = v.ms_c[ic]; mp[u] = v.ms_p[ic];
      f0[u] = MS(ms_found, 0, ic); f1[u] = MS(ms_found, 1, ic); sn[u] = v.ms_sin[ic];
    }
    Pose T[BA_ILP_PROJ]; double X[BA_ILP_PROJ][3];
    _Pragma("unroll") for (int u = 0; u < BA_ILP_PROJ; u++) {
      T[u] = ba_load_pose(v.cam_pose + mc[u]);
      _Pragma("unroll") for (int k = 0; k < 3; k++) X[u][k] = v.pt_pos[3 * mp[u] + k];
    }
    _Pragma("unroll") for (int u = 0; u < BA_ILP_PROJ; u++) {
      const int i = i0 + u * BA_THREADS;
      if (i >= nm) continue;
      double e2 = __builtin_huge_val(), e2p;
      if (st[u] != MS_ERASED && ba_project_meas(v, cfg, i, T[u], X[u], f0[u], f1[u], sn[u], e2p) == MS_OK) { e2 = e2p; nvalid++; }
      v.scratch[i] = e2;
    }
  }
  return nvalid;
}

// Passes 1 and 2 right after an accepted step, in one sweep: the committed cameras / points are the trial state
// FindNewError has just projected, so v3Cam, the radial factor (the atan), the residual and its square are taken from
// its stores instead of being recomputed -- the same values, bit for bit -- and the median of the squared errors (sigma) is
// already known, because FindNewError also left them in `scratch`.  What is left is the camera derivatives and the Tukey
// weights; the unweighted epsilon / derivatives / error^2 of the two-pass form are never written or read back
// (136 instead of 264 B per measurement).  Returns this thread's share of the objective (pass 2's `cur`).
__device__ __attribute__((noinline)) double ba_pass12_cached(const BaView& v_, const BaConfig& cfg_, int nm, double sigma2) {
  const BaViewG v = ba_g(v_);
  const BaConfig cfg = cfg_;
  double cur = 0.0;
  for (int i0 = threadIdx.x; i0 < nm; i0 += BA_ILP_C * BA_THREADS) {
    int st[BA_ILP_C]; double c[BA_ILP_C][3], fac[BA_ILP_C], e0[BA_ILP_C], e1[BA_ILP_C], sn[BA_ILP_C];
    _Pragma("unroll") for (int u = 0; u < BA_ILP_C; u++) {
      const int i = i0 + u * BA_THREADS, ic = i < nm ? i : nm - 1;
      st[u] = v.ms_state[ic];
      c[u][0] = MS(ms_tcam, 0, ic); c[u][1] = MS(ms_tcam, 1, ic); c[u][2] = MS(ms_tcam, 2, ic);
      fac[u] = v.ms_tfac[ic]; e0[u] = MS(ms_teps, 0, ic); e1[u] = MS(ms_teps, 1, ic); sn[u] = v.ms_sin[ic];
    }
    _Pragma("unroll") for (int u = 0; u < BA_ILP_C; u++) {
      const int i = i0 + u * BA_THREADS;
      if (i >= nm || st[u] == MS_ERASED) continue;
      MS(ms_cam, 0, i) = c[u][0]; MS(ms_cam, 1, i) = c[u][1]; MS(ms_cam, 2, i) = c[u][2];
      if (c[u][2] <= 0) { v.ms_state[i] = MS_BAD; cur += 1.0; continue; }          // pass 1: bBad; pass 2 (:243-246)
      CamProj pr;                                                     // cam_project minus its atan
      pr.cam[0] = c[u][0] / c[u][2]; pr.cam[1] = c[u][1] / c[u][2];
      pr.r = sqrt(pr.cam[0] * pr.cam[0] + pr.cam[1] * pr.cam[1]);
      pr.factor = fac[u]; pr.invalid = 0; pr.im[0] = 0; pr.im[1] = 0;
      double dd[4];
      cam_derivs(cfg.cam, pr, dd);
      const double e2 = e0[u] * e0[u] + e1[u] * e1[u];
      const double dWeight = tukey_sqrt_weight(e2, sigma2);
      MS(ms_eps, 0, i) = e0[u] * dWeight; MS(ms_eps, 1, i) = e1[u] * dWeight;
      if (dWeight == 0) { v.ms_state[i] = MS_BAD; cur += 1.0; continue; }
      v.ms_state[i] = MS_OK;
      cur += tukey_objective(e2, sigma2);
      _Pragma("unroll") for (int k = 0; k < 4; k++) MS(ms_derivs, k, i) = sn[u] * (dWeight * dd[k]);
    }
  }
  return cur;
}

// FindNewError (jni/Bundle.cc:537-561): this thread's share of the objective at the trial state.
// It also prepares the next Do_LM_Step in case this trial is accepted: the squared error of every measurement that stays
// in the list (state OK now, in front of the trial camera) goes to `scratch` for the median, +inf for the others, and
// their number is returned.
struct BaNewError { double ne; int nvalid; };
__device__ __attribute__((noinline)) BaNewError ba_find_new_error(const BaView& v_, const BaConfig& cfg_, int nm, double sigma2) {
  const BaViewG v = ba_g(v_);
  const BaConfig cfg = cfg_;
  double ne = 0.0;
  int nv = 0;
  for (int i0 = threadIdx.x; i0 < nm; i0 += BA_ILP_PROJ * BA_THREADS) {
    int st[BA_ILP_PROJ], mc[BA_ILP_PROJ], mp[BA_ILP_PROJ]; double f0[BA_ILP_PROJ], f1[BA_ILP_PROJ], sn[BA_ILP_PROJ];
    _Pragma("unroll") for (int u = 0; u < BA_ILP_PROJ; u++) {
      const int i = i0 + u * BA_THREADS, ic = i < nm ? i : nm - 1;
      st[u] = i < nm ? v.ms_state[ic] : MS_ERASED; mc[u] = v.ms_c[ic]; mp[u] = v.ms_p[ic];
      f0[u] = MS(ms_found, 0, ic); f1[u] = MS(ms_found, 1, ic); sn[u] = v.ms_sin[ic];
    }
    Pose T[BA_ILP_PROJ]; double X[BA_ILP_PROJ][3];
    _Pragma("unroll") for (int u = 0; u < BA_ILP_PROJ; u++) {
      T[u] = ba_load_pose(v.cam_new + mc[u]);
      _Pragma("unroll") for (int k = 0; k < 3; k++) X[u][k] = v.pt_new[3 * mp[u] + k];
    }
    _Pragma("unroll") for (int u = 0; u < BA_ILP_PROJ; u++) {
      if (st[u] == MS_ERASED) continue;
      const int i = i0 + u * BA_THREADS;
      double c[3];
      pose_xform(T[u], X[u], c);
      MS(ms_tcam, 0, i) = c[0]; MS(ms_tcam, 1, i) = c[1]; MS(ms_tcam, 2, i) = c[2];
      if (c[2] <= 0) { ne += 1.0; v.scratch[i] = __builtin_huge_val(); continue; }
      const CamProj pr = cam_project(cfg.cam, c[0] / c[2], c[1] / c[2]);
      const double e0 = (f0[u] - pr.im[0]) * sn[u], e1 = (f1[u] - pr.im[1]) * sn[u];
      v.ms_tfac[i] = pr.factor; MS(ms_teps, 0, i) = e0; MS(ms_teps, 1, i) = e1;
      const double e2 = e0 * e0 + e1 * e1;
      ne += tukey_objective(e2, sigma2);
      const bool stays = st[u] == MS_OK;                              // MS_BAD ones are erased at the end of this step
      v.scratch[i] = stays ? e2 : __builtin_huge_val();
      nv += stays ? 1 : 0;
    }
  }
  BaNewError r; r.ne = ne; r.nvalid = nv;
  return r;
}

// V, epsilon_b (jni/Bundle.cc:49-56, :312-316).  Each phase below is its own function so that it gets its own register
// allocation (see ba_pass1_project).
__device__ __attribute__((noinline)) void ba_accum_V(const BaView& v_, int nc, int np) {
  const BaViewG v = ba_g(v_);
  // V, epsilon_b per point: one lane per point, cameras in id order
  for (int p = threadIdx.x; p < np; p += BA_THREADS) {
    double V[6] = {0, 0, 0, 0, 0, 0}, eb[3] = {0, 0, 0};
    for (int c0 = 0; c0 < nc; c0 += BA_ILP_P) {
      int ii[BA_ILP_P]; MeasState ms[BA_ILP_P]; double e[BA_ILP_P][2];
      _Pragma("unroll") for (int u = 0; u < BA_ILP_P; u++) ii[u] = c0 + u < nc ? v.lut[(size_t)(c0 + u) * v.max_pts + p] : -1;
      _Pragma("unroll") for (int u = 0; u < BA_ILP_P; u++) {
        ba_load_state(v, ii[u], ms[u]);
        const int ic = ii[u] < 0 ? 0 : ii[u];
        e[u][0] = MS(ms_eps, 0, ic); e[u][1] = MS(ms_eps, 1, ic);
      }
      _Pragma("unroll") for (int u = 0; u < BA_ILP_P; u++) {
        if (ms[u].st != MS_OK) continue;
        double B[6];
        ba_jac_B(v.cam_pose[c0 + u].R, ms[u].cm, ms[u].d, B);
        int q = 0;
        _Pragma("unroll") for (int r = 0; r < 3; r++) for (int cc = 0; cc <= r; cc++) V[q++] += B[r] * B[cc] + B[3 + r] * B[3 + cc];   // :49-56 LL triangle
        _Pragma("unroll") for (int r = 0; r < 3; r++) eb[r] += B[r] * e[u][0] + B[3 + r] * e[u][1];
      }
    }
    // lower triangle of V: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) -> components 0..5
    _Pragma("unroll") for (int k = 0; k < 6; k++) PT(pt_V, k, p) = V[k];
    PT(pt_eb, 0, p) = eb[0]; PT(pt_eb, 1, p) = eb[1]; PT(pt_eb, 2, p) = eb[2];
  }
}

__device__ __attribute__((noinline)) void ba_accum_U(const BaView& v_, int nfree, int np) {
  const BaViewG v = ba_g(v_);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // U, epsilon_a per adjustable camera: one wavefront per camera (segmented wave reduction)
  for (int f = wave; f < nfree; f += BA_WAVES) {
    const int j = v.free_cams[f];
    double acc[27];
    _Pragma("unroll") for (int k = 0; k < 27; k++) acc[k] = 0.0;
    for (int p0 = lane; p0 < np; p0 += 64 * BA_ILP) {
      int ii[BA_ILP]; MeasState ms[BA_ILP]; double e[BA_ILP][2];
      _Pragma("unroll") for (int u = 0; u < BA_ILP; u++) ii[u] = p0 + 64 * u < np ? v.lut[(size_t)j * v.max_pts + p0 + 64 * u] : -1;
      _Pragma("unroll") for (int u = 0; u < BA_ILP; u++) {
        ba_load_state(v, ii[u], ms[u]);
        const int ic = ii[u] < 0 ? 0 : ii[u];
        e[u][0] = MS(ms_eps, 0, ic); e[u][1] = MS(ms_eps, 1, ic);
      }
      _Pragma("unroll") for (int u = 0; u < BA_ILP; u++) {
        if (ms[u].st != MS_OK) continue;
        double A[12];
        ba_jac_A(ms[u].cm, ms[u].d, A);
        int q = 0;
        _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) acc[q++] += A[r] * A[c] + A[6 + r] * A[6 + c];       // :40-47
        _Pragma("unroll") for (int r = 0; r < 6; r++) acc[21 + r] += A[r] * e[u][0] + A[6 + r] * e[u][1];
      }
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

// S diagonal block + E of one adjustable camera (jni/Bundle.cc:362-396); called by one wavefront.
__device__ __attribute__((noinline)) void ba_task_diag(const BaView& v_, int task, int np, int nS, double lambda) {
  const BaViewG v = ba_g(v_);
  const int lane = threadIdx.x & 63;
  const int j = v.free_cams[task], row = v.cam_row[j];
  double acc[27];
  _Pragma("unroll") for (int k = 0; k < 27; k++) acc[k] = 0.0;
  for (int p0 = lane; p0 < np; p0 += 64 * BA_ILP_S) {
    int ii[BA_ILP_S]; MeasState ms[BA_ILP_S]; double Vi[BA_ILP_S][9], eb[BA_ILP_S][3];
    _Pragma("unroll") for (int u = 0; u < BA_ILP_S; u++) ii[u] = p0 + 64 * u < np ? v.lut[(size_t)j * v.max_pts + p0 + 64 * u] : -1;
    _Pragma("unroll") for (int u = 0; u < BA_ILP_S; u++) {
      const int p = p0 + 64 * u < np ? p0 + 64 * u : np - 1;
      ba_load_state(v, ii[u], ms[u]);
      _Pragma("unroll") for (int k = 0; k < 9; k++) Vi[u][k] = PT(pt_Vinv, k, p);
      _Pragma("unroll") for (int k = 0; k < 3; k++) eb[u][k] = PT(pt_eb, k, p);
    }
    _Pragma("unroll") for (int u = 0; u < BA_ILP_S; u++) {
      if (ms[u].st != MS_OK) continue;
      double W[18], Y[18];
      ba_jac_W(ms[u], v.cam_pose[j].R, W);
      _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) Y[r * 3 + c] = W[r * 3] * Vi[u][c] + W[r * 3 + 1] * Vi[u][3 + c] + W[r * 3 + 2] * Vi[u][6 + c];
      int q = 0;
      _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) acc[q++] += Y[r * 3] * W[c * 3] + Y[r * 3 + 1] * W[c * 3 + 1] + Y[r * 3 + 2] * W[c * 3 + 2];
      double ve[3];
      _Pragma("unroll") for (int r = 0; r < 3; r++) ve[r] = Vi[u][r * 3] * eb[u][0] + Vi[u][r * 3 + 1] * eb[u][1] + Vi[u][r * 3 + 2] * eb[u][2];
      _Pragma("unroll") for (int r = 0; r < 6; r++) acc[21 + r] += W[r * 3] * ve[0] + W[r * 3 + 1] * ve[1] + W[r * 3 + 2] * ve[2];
    }
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
__device__ __attribute__((noinline)) void ba_task_pair(const BaView& v_, int task, int np, int nS) {
  const BaViewG v = ba_g(v_);
  const int lane = threadIdx.x & 63;
  int t = task, fj = 1;
  while (t >= fj) { t -= fj; fj++; }                         // pair (fj > fk): free-camera ordinals
  const int fk = t;
  const int j = v.free_cams[fj], k = v.free_cams[fk];
  const int jrow = v.cam_row[j], krow = v.cam_row[k];
  double acc[36];
  _Pragma("unroll") for (int q = 0; q < 36; q++) acc[q] = 0.0;
  for (int p0 = lane; p0 < np; p0 += 64 * BA_ILP_S) {
    int ij[BA_ILP_S], ik[BA_ILP_S]; MeasState mj[BA_ILP_S], mk[BA_ILP_S]; double Vi[BA_ILP_S][9];
    _Pragma("unroll") for (int u = 0; u < BA_ILP_S; u++) {
      const bool in = p0 + 64 * u < np;
      ij[u] = in ? v.lut[(size_t)j * v.max_pts + p0 + 64 * u] : -1;
      ik[u] = in ? v.lut[(size_t)k * v.max_pts + p0 + 64 * u] : -1;
    }
    _Pragma("unroll") for (int u = 0; u < BA_ILP_S; u++) {
      const int p = p0 + 64 * u < np ? p0 + 64 * u : np - 1;
      ba_load_state(v, ij[u], mj[u]);
      ba_load_state(v, ik[u], mk[u]);
      _Pragma("unroll") for (int q = 0; q < 9; q++) Vi[u][q] = PT(pt_Vinv, q, p);
    }
    _Pragma("unroll") for (int u = 0; u < BA_ILP_S; u++) {
      if (mj[u].st != MS_OK || mk[u].st != MS_OK) continue;
      double Y[18];
      {
        double Wj[18];
        ba_jac_W(mj[u], v.cam_pose[j].R, Wj);
        _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) Y[r * 3 + c] = Wj[r * 3] * Vi[u][c] + Wj[r * 3 + 1] * Vi[u][3 + c] + Wj[r * 3 + 2] * Vi[u][6 + c];
      }
      double Wk[18];
      ba_jac_W(mk[u], v.cam_pose[k].R, Wk);
      _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) acc[r * 6 + c] += Y[r * 3] * Wk[c * 3] + Y[r * 3 + 1] * Wk[c * 3 + 1] + Y[r * 3 + 2] * Wk[c * 3 + 2];
    }
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
// otherwise, plus one row V*^-1 eb_p) times the 3 x 6 n_free stack W_p^T.  The wave-per-block form above re-derives W_j and Y_j for
// every camera pair (10 x 650 + 5 x 400 fp64 instructions per point); here each (point, camera) is derived once (one lane
// each, BA_MFMA_PPC points per wavefront and trip), staged k-major in LDS and multiplied by v_mfma_f64_16x16x4_f64
// (operand layout: tools/probes/mfma_f64_layout.hip; a = A[l%16][l/16], b = B[l/16][l%16], d[v] = D[l/16+4v][l%16]).
// Wave partials are added in wave order, the lower triangle is mirrored as the reference mirrors it (:431-434).
#define BA_MFMA_FREE 5
#ifndef BA_MFMA_PPC
#define BA_MFMA_PPC 12                                  // points per wavefront and trip: 12 x 5 cameras = 60 lanes, K = 36
#endif
#define BA_MFMA_K (3 * BA_MFMA_PPC)
#define BA_MFMA_STAGE (4 * BA_MFMA_K * 16)              // doubles per wavefront: Y rows 0-15 / 16-31, W columns 0-15 / 16-31, each [K][16]
static_assert(BA_MFMA_STAGE >= 32 * 32 && BA_MFMA_K % 4 == 0, "a wavefront's staging area also holds its 32 x 32 partial product");
typedef double ba_v4d __attribute__((ext_vector_type(4)));
__device__ __attribute__((noinline)) void ba_schur_mfma(const BaView& v_, int nfree, int np, int nS, double lambda, double* lds_) {
  const BaViewG v = ba_g(v_);
  double AS3* lds = (double AS3*)lds_;                    // the staging buffer is LDS: ds_read / ds_write, not flat
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double AS3* Y0 = lds + wave * BA_MFMA_STAGE; double AS3* Y1 = Y0 + BA_MFMA_K * 16; double AS3* W0 = Y1 + BA_MFMA_K * 16; double AS3* W1 = W0 + BA_MFMA_K * 16;
  for (int t = lane; t < BA_MFMA_STAGE; t += 64) Y0[t] = 0.0;      // rows / columns no lane ever writes stay zero
  const bool active = lane < BA_MFMA_PPC * nfree;
  const int pl = active ? lane / nfree : 0, f = active ? lane - pl * nfree : 0;
  const int j = v.free_cams[f];
  double Rj[9];
  _Pragma("unroll") for (int k = 0; k < 9; k++) Rj[k] = v.cam_pose[j].R[k];
  ba_v4d d00 = {0, 0, 0, 0}, d10 = {0, 0, 0, 0}, d11 = {0, 0, 0, 0};
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  // two-deep software pipeline over the trips: the measurement index of trip t+2 and the operands of trip t+1 are in
  // flight while trip t is derived, staged and multiplied (lut -> state is a dependent pair of global round trips)
  constexpr int STRIDE = BA_WAVES * BA_MFMA_PPC;
  const int pfirst = wave * BA_MFMA_PPC;
  auto lut_of = [&](int p0) -> int { const int p = p0 + pl; return (active && p < np) ? v.lut[(size_t)j * v.max_pts + p] : -1; };
  int i_nxt = lut_of(pfirst + STRIDE);
  MeasState ms_n; double Vi_n[9], eb_n[3];
  {
    const int pc = pfirst + pl < np ? pfirst + pl : np - 1;
    ba_load_state(v, lut_of(pfirst), ms_n);
    _Pragma("unroll") for (int k = 0; k < 9; k++) Vi_n[k] = PT(pt_Vinv, k, pc);
    _Pragma("unroll") for (int k = 0; k < 3; k++) eb_n[k] = PT(pt_eb, k, pc);
  }
  for (int p0 = pfirst; p0 < np; p0 += STRIDE) {
    const int p = p0 + pl;
    const MeasState ms = ms_n;
    double Vi[9], eb[3];
    _Pragma("unroll") for (int k = 0; k < 9; k++) Vi[k] = Vi_n[k];
    _Pragma("unroll") for (int k = 0; k < 3; k++) eb[k] = eb_n[k];
    {
      const int i_cur = i_nxt;
      i_nxt = lut_of(p0 + 2 * STRIDE);
      const int pn = p0 + STRIDE + pl, pc = pn < np ? pn : np - 1;
      ba_load_state(v, i_cur, ms_n);
      _Pragma("unroll") for (int k = 0; k < 9; k++) Vi_n[k] = PT(pt_Vinv, k, pc);
      _Pragma("unroll") for (int k = 0; k < 3; k++) eb_n[k] = PT(pt_eb, k, pc);
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
        const int row = 6 * f + r;
        double AS3* yd = (row < 16 ? Y0 + row : Y1 + (row - 16)) + 3 * pl * 16;
        double AS3* wd = (row < 16 ? W0 + row : W1 + (row - 16)) + 3 * pl * 16;
        _Pragma("unroll") for (int c = 0; c < 3; c++) { yd[c * 16] = Y[r * 3 + c]; wd[c * 16] = W[r * 3 + c]; }
      }
      if (f == 0) {                                               // row 30 of the left operand: V*^-1 eb_p, so that D[30][.] = sum_p W (V*^-1 eb) (:388-396)
        _Pragma("unroll") for (int c = 0; c < 3; c++) Y1[(3 * pl + c) * 16 + 14] = p < np ? Vi[c * 3] * eb[0] + Vi[c * 3 + 1] * eb[1] + Vi[c * 3 + 2] * eb[2] : 0.0;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0) only: the prefetched global loads stay in flight
    __builtin_amdgcn_wave_barrier();
    _Pragma("unroll") for (int ks = 0; ks < BA_MFMA_K / 4; ks++) {
      const int o = (ks * 4 + (lane >> 4)) * 16 + (lane & 15);
      const double a0 = Y0[o], a1 = Y1[o], b0 = W0[o], b1 = W1[o];
      d00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d00, 0, 0, 0);   // the tile of rows 0-15 x columns 16-31 lies above the diagonal: not needed
      d10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, d10, 0, 0, 0);
      d11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, d11, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0) only: the prefetched global loads stay in flight
    __builtin_amdgcn_wave_barrier();
  }
  // the wave's 32 x 32 partial product, row-major, into its own staging area
  __syncthreads();
  _Pragma("unroll") for (int q = 0; q < 4; q++) {
    const int r = (lane >> 4) + 4 * q, c = lane & 15;
    Y0[r * 32 + c] = d00[q]; Y0[(16 + r) * 32 + c] = d10[q]; Y0[(16 + r) * 32 + 16 + c] = d11[q];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 32 * 32; t += BA_THREADS) {
    const int r = t >> 5, c = t & 31;
    if (!(r < nS || r == 30) || c >= nS || (r < nS && c > r)) continue;
    double sum = 0.0;
    for (int w = 0; w < BA_WAVES; w++) sum += lds[w * BA_MFMA_STAGE + t];
    if (r == 30) { const int fk = c / 6, kk = v.free_cams[fk]; v.E[c] = v.cam_ea[6 * kk + (c - 6 * fk)] - sum; continue; }   // rows of S follow the adjustable cameras in order (cam_row == 6 * ordinal)
    const int fj = r / 6, jj = v.free_cams[fj];
    double u = 0.0;
    if (c / 6 == fj) { u = v.cam_U[36 * jj + (r - 6 * fj) * 6 + (c - 6 * fj)]; if (r == c) u *= (1.0 + lambda); }
    const double val = u - sum;
    v.S[(size_t)r * nS + c] = val; v.S[(size_t)c * nS + r] = val;
  }
  __syncthreads();
}

// map updates (jni/Bundle.cc:440-462, :484): trial point positions; returns this thread's share of |update|^2.
__device__ __attribute__((noinline)) double ba_map_update(const BaView& v_, int nfree, int np) {
  const BaViewG v = ba_g(v_);
  double ssq = 0.0;
  for (int p = threadIdx.x; p < np; p += BA_THREADS) {
    double sum[3] = {0, 0, 0};
    for (int f0 = 0; f0 < nfree; f0 += BA_ILP_P) {
      int jj[BA_ILP_P]; MeasState ms[BA_ILP_P];
      _Pragma("unroll") for (int u = 0; u < BA_ILP_P; u++) jj[u] = v.free_cams[f0 + u < nfree ? f0 + u : nfree - 1];
      _Pragma("unroll") for (int u = 0; u < BA_ILP_P; u++) ba_load_state(v, f0 + u < nfree ? v.lut[(size_t)jj[u] * v.max_pts + p] : -1, ms[u]);
      _Pragma("unroll") for (int u = 0; u < BA_ILP_P; u++) {
        if (ms[u].st != MS_OK) continue;
        double W[18];
        ba_jac_W(ms[u], v.cam_pose[jj[u]].R, W);
        const double AS1* cu = v.cam_up + v.cam_row[jj[u]];
        _Pragma("unroll") for (int c = 0; c < 3; c++) { double s = 0; for (int r = 0; r < 6; r++) s += W[r * 3 + c] * cu[r]; sum[c] += s; }
      }
    }
    const double eb[3] = {PT(pt_eb, 0, p), PT(pt_eb, 1, p), PT(pt_eb, 2, p)};
    double Vi[9]; _Pragma("unroll") for (int k = 0; k < 9; k++) Vi[k] = PT(pt_Vinv, k, p);
    const double x[3] = {eb[0] - sum[0], eb[1] - sum[1], eb[2] - sum[2]};
    _Pragma("unroll") for (int r = 0; r < 3; r++) {
      const double u = Vi[r * 3] * x[0] + Vi[r * 3 + 1] * x[1] + Vi[r * 3 + 2] * x[2];
      ssq += u * u;
      v.pt_new[3 * p + r] = v.pt_pos[3 * p + r] + u;               // :484
    }
  }
  return ssq;
}

// Bundle::Compute.  Called by all BA_THREADS threads of one workgroup.
DEVFN void ba_compute(const BaView& v_, const BaConfig& cfg) {
  const BaViewG v = ba_g(v_);
  __shared__ double red[BA_WAVES];
  __shared__ int ired[BA_WAVES];
  __shared__ int hist[768];
  __shared__ unsigned long long sel[1];
  __shared__ double sh_lambda, sh_factor, sh_sigma2, sh_cur_err, sh_new_err;
  __shared__ double lds_buf[BA_WAVES * BA_MFMA_STAGE > BA_LDS_N * (BA_LDS_N + 1) ? BA_WAVES * BA_MFMA_STAGE : BA_LDS_N * (BA_LDS_N + 1)];
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
#ifdef VSLAM_BA_PROF
  unsigned long long ba_t0 = clock64();
#endif

  while (!sh_converged && !sh_hitmax && !sh_error) {             // :153 (no abort signal: the map-maker runs synchronously)
    // ================= Do_LM_Step =================
    // pass 1 (:209-215): project every measurement still in the list
    const bool cached = sh_cache_valid != 0;                       // the previous step was accepted: FindNewError has projected this state
    int nvalid;
    if (cached) nvalid = sh_next_nvalid;
    else nvalid = ba_block_sum_i(ba_pass1_project(v_, cfg, nm), ired);
    BA_STAMP(1);
    if (nvalid == 0) { if (threadIdx.x == 0) sh_error = 1; __syncthreads(); break; }
    {                                                              // :220-227 Tukey sigma, clamped
      const double med = block_radix_select(v.scratch, nm, nvalid / 2, hist, sel);
      double s2 = tukey_sigma_squared(med, (unsigned long)nvalid);
      if (s2 < cfg.min_sigma2) s2 = cfg.min_sigma2;
      if (threadIdx.x == 0) sh_sigma2 = s2;
      __syncthreads();
    }
    const double sigma2 = sh_sigma2;
    BA_STAMP(2);
    // pass 2 (:241-321): weights and objective; A, B, W are re-derived by their consumers
    double cur = 0.0;
    if (cached) cur = ba_pass12_cached(v_, cfg, nm, sigma2);
    else for (int i0 = threadIdx.x; i0 < nm; i0 += BA_ILP_W * BA_THREADS) {   // BA_ILP_W measurements in flight per thread
      int stt[BA_ILP_W]; double e2[BA_ILP_W], ep0[BA_ILP_W], ep1[BA_ILP_W], sn[BA_ILP_W], dd[BA_ILP_W][4];
      _Pragma("unroll") for (int u = 0; u < BA_ILP_W; u++) {
        const int i = i0 + u * BA_THREADS, ic = i < nm ? i : nm - 1;   // unconditional loads: no branch between them
        stt[u] = v.ms_state[ic];
        e2[u] = v.ms_err2[ic]; ep0[u] = MS(ms_eps, 0, ic); ep1[u] = MS(ms_eps, 1, ic); sn[u] = v.ms_sin[ic];
        _Pragma("unroll") for (int k = 0; k < 4; k++) dd[u][k] = MS(ms_derivs, k, ic);
        if (i >= nm) stt[u] = MS_ERASED;
      }
      _Pragma("unroll") for (int u = 0; u < BA_ILP_W; u++) {
        const int i = i0 + u * BA_THREADS;
        if (stt[u] == MS_ERASED) continue;
        if (stt[u] == MS_BAD) { cur += 1.0; continue; }
        const double dWeight = tukey_sqrt_weight(e2[u], sigma2);
        MS(ms_eps, 0, i) = ep0[u] * dWeight; MS(ms_eps, 1, i) = ep1[u] * dWeight;
        if (dWeight == 0) { v.ms_state[i] = MS_BAD; cur += 1.0; continue; }
        cur += tukey_objective(e2[u], sigma2);
        _Pragma("unroll") for (int k = 0; k < 4; k++) MS(ms_derivs, k, i) = sn[u] * (dWeight * dd[u][k]);   // weighted from here on
      }
    }
    cur = ba_block_sum(cur, red);
    if (threadIdx.x == 0) sh_cur_err = cur;
    __syncthreads();
    BA_STAMP(3);
    ba_accum_V(v_, nc, np);
    BA_STAMP(4);
    ba_accum_U(v_, nfree, np);
    __syncthreads();
    BA_STAMP(5);
    // ---- inner loop over lambda (:326-501) ----
    if (threadIdx.x == 0) sh_new_err = sh_cur_err + 9999;
    __syncthreads();
    while (sh_new_err > sh_cur_err && !sh_converged && !sh_hitmax && !sh_error) {
      const double lambda = sh_lambda;
      for (int p = threadIdx.x; p < np; p += BA_THREADS) {           // V*^-1 (:329-347)
        const double v00 = PT(pt_V, 0, p), v10 = PT(pt_V, 1, p), v11 = PT(pt_V, 2, p), v20 = PT(pt_V, 3, p), v21 = PT(pt_V, 4, p), v22 = PT(pt_V, 5, p);
        double Vi[9];
        if (v00 * v11 * v22 == 0) { _Pragma("unroll") for (int k = 0; k < 9; k++) Vi[k] = 0.0; }
        else {
          const double Vs[9] = {v00 * (1.0 + lambda), v10, v20, v10, v11 * (1.0 + lambda), v21, v20, v21, v22 * (1.0 + lambda)};
          inv3(Vs, Vi);
        }
        _Pragma("unroll") for (int k = 0; k < 9; k++) PT(pt_Vinv, k, p) = Vi[k];
      }
      for (int t = threadIdx.x; t < nS * nS; t += BA_THREADS) v.S[t] = 0.0;
      __syncthreads();
      BA_STAMP(6);
      // S: diagonal blocks + E (:362-396) and off-diagonal blocks (:400-426); one wavefront per block
      if (nfree <= BA_MFMA_FREE) ba_schur_mfma(v_, nfree, np, nS, lambda, lds_buf);
      else {
        const int ntask = nfree + nfree * (nfree - 1) / 2;
        for (int task = wave; task < ntask; task += BA_WAVES) {
          if (task < nfree) ba_task_diag(v_, task, np, nS, lambda);
          else ba_task_pair(v_, task - nfree, np, nS);
        }
      }
      __syncthreads();
      BA_STAMP(7);
      bool solved = true;
      if (nS > 0 && nS <= BA_WSOLVE_N) {                              // one wavefront, registers (the BundleAdjustRecent size)
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
      double ssq = ba_map_update(v_, nfree, np);
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
      const BaNewError fne = ba_find_new_error(v_, cfg, nm, sigma2);
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
    // erase the outliers in list order (:517-528): ordered compaction of the (p, c) pairs, BA_ERASE_B chunks of the list
    // per barrier pair (all their states are loaded first; one LDS table of per-chunk, per-wavefront counts)
    {
      constexpr int BA_ERASE_B = 8;
      __shared__ int ecnt[BA_ERASE_B * BA_WAVES];
      int base = sh_nout;
      for (int i0 = 0; i0 < nm; i0 += BA_ERASE_B * BA_THREADS) {
        bool bad[BA_ERASE_B]; unsigned long long bm[BA_ERASE_B];
        _Pragma("unroll") for (int u = 0; u < BA_ERASE_B; u++) {
          const int i = i0 + u * BA_THREADS + threadIdx.x;
          bad[u] = v.ms_state[i < nm ? i : nm - 1] == MS_BAD && i < nm;
        }
        _Pragma("unroll") for (int u = 0; u < BA_ERASE_B; u++) { bm[u] = __ballot(bad[u]); if (lane == 0) ecnt[u * BA_WAVES + wave] = __popcll(bm[u]); }
        __syncthreads();
        _Pragma("unroll") for (int u = 0; u < BA_ERASE_B; u++) {
          int off = base;
          for (int w = 0; w < BA_WAVES; w++) { const int c = ecnt[u * BA_WAVES + w]; if (w < wave) off += c; base += c; }
          if (bad[u]) {
            const int i = i0 + u * BA_THREADS + threadIdx.x;
            off += __popcll(bm[u] & ((1ull << lane) - 1ull));
            const int pp = v.ms_p[i], cc = v.ms_c[i];
            v.outl[2 * off] = pp; v.outl[2 * off + 1] = cc;
            v.ms_state[i] = MS_ERASED;
            v.lut[(size_t)cc * v.max_pts + pp] = -1;
            atomicAdd((int*)&v.pt_nout[pp], 1);
          }
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) sh_nout = base;
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
