// Bundle::Compute (jni/Bundle.cc:136-178, Do_LM_Step :202-532) with EVERY floating-point sum taken in the reference's order
// (vslam_params.ba_sum_order = 1): the parity mode of the bundle adjustment.
//
// The fast path (ba_device.h) sums U / epsilon_a, V / epsilon_b, the reduced camera system and the objectives as per-wavefront
// partial sums, tree reductions and matrix-core products: its results agree with the reference's sequential loops to ~1e-10, not
// to the bit, and PTAM's trunc(bilinear) templates turn a last-bit difference of the map into a different template pixel now and
// then.  Here every term is computed in parallel, one lane per measurement slot -- the same expressions as the fast path and as
// the reference -- and left in global memory; each SUM is then walked by one lane in the order of the reference's loops:
//   dCurrentError, dNewError            the measurement list in AddMeas order (jni/Bundle.cc:241-266, 537-561)
//   U_j, epsilon_a_j                    the measurements of camera j in list order (:306-311)
//   V_i, epsilon_b_i                    the measurements of point i in list order (:312-316)
//   S_jj, E_j                           U*_j / epsilon_a_j minus the points' terms, points ascending (:362-396)
//   S_jk                                0 minus the points' terms, points ascending (the scripts :400-426 visit a block once per point)
//   map update of a point               its adjustable cameras ascending (:440-462)
//   dSumSquaredUpdate                   the camera updates, then the point updates, in index order (:467-470)
// A skipped entry adds +0.0, which leaves a sum's bits unchanged (no accumulator here can be -0.0).  The independent sums run side
// by side in different lanes; what is sequential is each lane's chain of additions.  One persistent workgroup per problem, like
// the fast path; layout, median, solve and outlier erase are the fast path's own functions (exact already: integer work, an order
// statistic, and an elimination that is sequential per element).
#pragma once
#include "ba_device.h"

struct BaOrdView {       // per-problem arrays of the parity mode (allocated only with ba_sum_order = 1); per-slot arrays are component-major
  int* of_logical;       // [max_meas]      list index -> slot, or -1
  double* obj;           // [max_meas]      a slot's term of dCurrentError / dNewError
  double* v9;            // [9][max_meas]   B^T B (lower triangle, 6) and B^T epsilon (3)
  double* u27;           // [27][max_meas]  A^T A (lower triangle, 21) and A^T epsilon (6); F slots
  double* W;             // [18][max_meas]  A^T B (6 x 3, row-major); F slots
  double* Y;             // [18][max_meas]  W V*^-1 of the trial; F slots
  double* ve;            // [3 max_pts]     V*^-1 epsilon_b of the trial
  double* up;            // [3 max_pts]     the trial's point updates
};
#define OSL(arr, k, s) o.arr[(size_t)(k) * v.max_meas + (s)]
#define ORD_PF 8         // entries a chain lane requests before it adds them: the additions are the only dependent work

// The projection pass at the committed (trial = 0) or trial state: FindNewError (:537-561) / pass 1 (:209-215).  Leaves each slot's
// term of the error sum in o.obj and the squared error for the next median in v.scratch; returns this thread's count of
// measurements that stay in the list.
DEVFN int ord_errors(const BaViewG& v, const BaConfig& cfg, const BaOrdView& o, int M, double sigma2, int trial, const double* camL) {
  const double AS1* pts = trial ? v.pt_new : v.pt_pos;
  int nv = 0;
  for (int s = threadIdx.x; s < M; s += BA_THREADS) {
    const int info = v.sl_info[s], st = SL_STATE(info);
    if (st == MS_ERASED) { o.obj[s] = 0.0; continue; }
    const int cj = SL_CAM(info), p = v.sl_pt[s];
    Pose T;
    _Pragma("unroll") for (int q = 0; q < 9; q++) T.R[q] = camL[cj * 12 + q];
    _Pragma("unroll") for (int q = 0; q < 3; q++) T.t[q] = camL[cj * 12 + 9 + q];
    const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
    double c[3];
    pose_xform(T, X, c);
    if (c[2] <= 0) { o.obj[s] = 1.0; v.scratch[s] = __builtin_huge_val(); continue; }
    const CamProj pr = cam_project(cfg.cam, c[0] / c[2], c[1] / c[2]);
    const double sn = v.sl_sin[s];
    const double e0 = (SL(sl_found, 0, s) - pr.im[0]) * sn, e1 = (SL(sl_found, 1, s) - pr.im[1]) * sn;
    const double e2 = e0 * e0 + e1 * e1;
    o.obj[s] = tukey_objective(e2, sigma2);
    const bool stays = st == MS_OK;
    v.scratch[s] = stays ? e2 : __builtin_huge_val();
    nv += stays ? 1 : 0;
  }
  return nv;
}

// Passes 1 + 2 of Do_LM_Step (:209-321) per slot: state, weighted derivatives, and the slot's terms of every sum of the step.
DEVFN void ord_sweep(const BaViewG& v, const BaConfig& cfg, const BaOrdView& o, int M, int MF, double sigma2, const double* camL) {
  for (int s = threadIdx.x; s < M; s += BA_THREADS) {
    const int info = v.sl_info[s];
    int st = SL_STATE(info);
    const bool isF = s < MF;
    double obj = 0.0, v9[9], u27[27], W[18], d[4] = {0, 0, 0, 0};
    _Pragma("unroll") for (int q = 0; q < 9; q++) v9[q] = 0.0;
    _Pragma("unroll") for (int q = 0; q < 27; q++) u27[q] = 0.0;
    _Pragma("unroll") for (int q = 0; q < 18; q++) W[q] = 0.0;
    if (st != MS_ERASED) {
      const int cj = SL_CAM(info), p = v.sl_pt[s];
      Pose T;
      _Pragma("unroll") for (int q = 0; q < 9; q++) T.R[q] = camL[cj * 12 + q];
      _Pragma("unroll") for (int q = 0; q < 3; q++) T.t[q] = camL[cj * 12 + 9 + q];
      const double X[3] = {v.pt_pos[3 * p], v.pt_pos[3 * p + 1], v.pt_pos[3 * p + 2]};
      double c[3];
      pose_xform(T, X, c);
      if (c[2] <= 0) { st = MS_BAD; obj = 1.0; }                          // :186-189, :243-246
      else {
        const CamProj pr = cam_project(cfg.cam, c[0] / c[2], c[1] / c[2]);
        double dd[4];
        cam_derivs(cfg.cam, pr, dd);
        const double sn = v.sl_sin[s];
        double e0 = (SL(sl_found, 0, s) - pr.im[0]) * sn, e1 = (SL(sl_found, 1, s) - pr.im[1]) * sn;
        const double e2 = e0 * e0 + e1 * e1;
        const double dWeight = tukey_sqrt_weight(e2, sigma2);
        e0 *= dWeight; e1 *= dWeight;
        if (dWeight == 0) { st = MS_BAD; obj = 1.0; }                     // :262-266
        else {
          st = MS_OK;
          obj = tukey_objective(e2, sigma2);
          _Pragma("unroll") for (int q = 0; q < 4; q++) d[q] = sn * (dWeight * dd[q]);
          double B[6];
          ba_jac_B(T.R, c, d, B);
          int q = 0;
          _Pragma("unroll") for (int r = 0; r < 3; r++) for (int cc = 0; cc <= r; cc++) v9[q++] = B[r] * B[cc] + B[3 + r] * B[3 + cc];   // :49-56
          _Pragma("unroll") for (int r = 0; r < 3; r++) v9[6 + r] = B[r] * e0 + B[3 + r] * e1;
          if (isF) {
            double A[12];
            ba_jac_A(c, d, A);
            q = 0;
            _Pragma("unroll") for (int r = 0; r < 6; r++) for (int cc = 0; cc <= r; cc++) u27[q++] = A[r] * A[cc] + A[6 + r] * A[6 + cc];     // :40-47
            _Pragma("unroll") for (int r = 0; r < 6; r++) u27[21 + r] = A[r] * e0 + A[6 + r] * e1;
            _Pragma("unroll") for (int r = 0; r < 6; r++) for (int cc = 0; cc < 3; cc++) W[r * 3 + cc] = A[r] * B[cc] + A[6 + r] * B[3 + cc];   // :302
          }
        }
      }
      v.sl_info[s] = SL_WITH_STATE(info, st);
    }
    o.obj[s] = obj;
    _Pragma("unroll") for (int q = 0; q < 9; q++) OSL(v9, q, s) = v9[q];
    if (isF) {
      _Pragma("unroll") for (int q = 0; q < 27; q++) OSL(u27, q, s) = u27[q];
      _Pragma("unroll") for (int q = 0; q < 18; q++) OSL(W, q, s) = W[q];
      _Pragma("unroll") for (int q = 0; q < 4; q++) SL(sl_d, q, s) = d[q];
    }
  }
}

// One lane's walk over the measurement list in AddMeas order: task 0 sums the objective terms (o.obj), task 1 + 27 f + q value q of
// U / epsilon_a of the adjustable camera of ordinal f.  Returns the sum.
DEVFN double ord_list_chain(const BaViewG& v, const BaOrdView& o, int nm, int MF, int task) {
  const int f = task > 0 ? (task - 1) / 27 : -1, q = task > 0 ? (task - 1) - 27 * f : 0;
  double acc = 0.0;
  for (int i0 = 0; i0 < nm; i0 += ORD_PF) {
    double x[ORD_PF];
    _Pragma("unroll") for (int u = 0; u < ORD_PF; u++) {
      const int i = i0 + u;
      const int s = i < nm ? o.of_logical[i] : -1;
      x[u] = 0.0;
      if (s >= 0) {
        if (task == 0) x[u] = o.obj[s];
        else if (s < MF && SL_FORD(v.sl_info[s]) == f) x[u] = OSL(u27, q, s);
      }
    }
    _Pragma("unroll") for (int u = 0; u < ORD_PF; u++) acc += x[u];
  }
  return acc;
}

// V_i, epsilon_b_i (:312-316): one lane per point adds the terms of the point's slots in list order (its F slots and its X slots are
// each in camera order; the list may interleave them: selection by the next larger list index).
DEVFN void ord_point_sums(const BaViewG& v, const BaOrdView& o, int np, int MF) {
  for (int p = threadIdx.x; p < np; p += BA_THREADS) {
    const int a0 = v.pt_offF[p], a1 = v.pt_offF[p + 1], b0 = MF + v.pt_offX[p], b1 = MF + v.pt_offX[p + 1];
    const int n = (a1 - a0) + (b1 - b0);
    double acc[9];
    _Pragma("unroll") for (int q = 0; q < 9; q++) acc[q] = 0.0;
    int last = -1;
    for (int it = 0; it < n; it++) {
      int best = 0x7fffffff, bs = -1;
      for (int s = a0; s < a1; s++) { const int l = v.sl_logical[s]; if (l > last && l < best) { best = l; bs = s; } }
      for (int s = b0; s < b1; s++) { const int l = v.sl_logical[s]; if (l > last && l < best) { best = l; bs = s; } }
      if (bs < 0) break;
      _Pragma("unroll") for (int q = 0; q < 9; q++) acc[q] += OSL(v9, q, bs);
      last = best;
    }
    _Pragma("unroll") for (int q = 0; q < 6; q++) PT(pt_V, q, p) = acc[q];
    _Pragma("unroll") for (int q = 0; q < 3; q++) PT(pt_eb, q, p) = acc[6 + q];
  }
}

// The operands of a trial's reduced camera system (:329-347 and the products inside :362-426): V*^-1 epsilon_b per point, W V*^-1 per F slot
DEVFN void ord_trial_operands(const BaViewG& v, const BaOrdView& o, int np, int MF, double lambda) {
  for (int p = threadIdx.x; p < np; p += BA_THREADS) {
    double Vi[9];
    ba_vstar_inv(v, p, lambda, Vi);
    const double eb[3] = {PT(pt_eb, 0, p), PT(pt_eb, 1, p), PT(pt_eb, 2, p)};
    _Pragma("unroll") for (int r = 0; r < 3; r++) o.ve[3 * p + r] = Vi[r * 3] * eb[0] + Vi[r * 3 + 1] * eb[1] + Vi[r * 3 + 2] * eb[2];
  }
  for (int s = threadIdx.x; s < MF; s += BA_THREADS) {
    if (SL_STATE(v.sl_info[s]) != MS_OK) continue;
    double Vi[9], W[18];
    ba_vstar_inv(v, v.sl_pt[s], lambda, Vi);
    _Pragma("unroll") for (int q = 0; q < 18; q++) W[q] = OSL(W, q, s);
    _Pragma("unroll") for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) OSL(Y, r * 3 + c, s) = W[r * 3] * Vi[c] + W[r * 3 + 1] * Vi[3 + c] + W[r * 3 + 2] * Vi[6 + c];
  }
}

// One element of the reduced camera system: task < 27 nfree is value q of the diagonal block / E of the adjustable camera of ordinal
// f = task / 27 (the 21 elements of the lower triangle, then the 6 of E); the tasks behind them are the 36 elements of the blocks
// between two adjustable cameras fj > fk.  The lane walks the points in ascending order, subtracting each point's term.
DEVFN void ord_schur_chain(const BaViewG& v, const BaOrdView& o, int np, int nfree, int nS, double lambda, int task) {
  if (task < 27 * nfree) {
    const int f = task / 27, q = task - 27 * f, j = v.free_cams[f], row = v.cam_row[j];
    int r = 0, c = 0;
    double acc;
    if (q < 21) { int qq = q; while (qq > r) { qq -= r + 1; r++; } c = qq; acc = v.cam_U[36 * j + r * 6 + c]; if (r == c) acc *= (1.0 + lambda); }   // :370-377
    else { r = q - 21; acc = v.cam_ea[6 * j + r]; }
    for (int p0 = 0; p0 < np; p0 += 4) {
      double t[4];
      _Pragma("unroll") for (int u = 0; u < 4; u++) {
        const int p = p0 + u;
        const int s = p < np ? ba_slot_of(v, p, f) : -1;
        t[u] = 0.0;
        if (s >= 0 && SL_STATE(v.sl_info[s]) == MS_OK) {
          if (q < 21) t[u] = OSL(Y, r * 3, s) * OSL(W, c * 3, s) + OSL(Y, r * 3 + 1, s) * OSL(W, c * 3 + 1, s) + OSL(Y, r * 3 + 2, s) * OSL(W, c * 3 + 2, s);
          else t[u] = OSL(W, r * 3, s) * o.ve[3 * p] + OSL(W, r * 3 + 1, s) * o.ve[3 * p + 1] + OSL(W, r * 3 + 2, s) * o.ve[3 * p + 2];
        }
      }
      _Pragma("unroll") for (int u = 0; u < 4; u++) acc -= t[u];
    }
    if (q < 21) { v.S[(size_t)(row + r) * nS + row + c] = acc; v.S[(size_t)(row + c) * nS + row + r] = acc; }   // mirrored :431-434
    else v.E[row + r] = acc;
    return;
  }
  int t = task - 27 * nfree;
  const int e = t % 36; t /= 36;
  int fj = 1;
  while (t >= fj) { t -= fj; fj++; }
  const int fk = t, r = e / 6, c = e - 6 * r;
  const int jrow = v.cam_row[v.free_cams[fj]], krow = v.cam_row[v.free_cams[fk]];
  double acc = 0.0;
  for (int p0 = 0; p0 < np; p0 += 4) {
    double tt[4];
    _Pragma("unroll") for (int u = 0; u < 4; u++) {
      const int p = p0 + u;
      const int sj = p < np ? ba_slot_of(v, p, fj) : -1, sk = p < np ? ba_slot_of(v, p, fk) : -1;
      tt[u] = 0.0;
      if (sj >= 0 && sk >= 0 && SL_STATE(v.sl_info[sj]) == MS_OK && SL_STATE(v.sl_info[sk]) == MS_OK)
        tt[u] = OSL(Y, r * 3, sj) * OSL(W, c * 3, sk) + OSL(Y, r * 3 + 1, sj) * OSL(W, c * 3 + 1, sk) + OSL(Y, r * 3 + 2, sj) * OSL(W, c * 3 + 2, sk);
    }
    _Pragma("unroll") for (int u = 0; u < 4; u++) acc -= tt[u];
  }
  v.S[(size_t)(jrow + r) * nS + krow + c] = acc; v.S[(size_t)(krow + c) * nS + jrow + r] = acc;
}

// map updates (:440-462, :484): one lane per point, its adjustable cameras ascending.  Leaves the update in o.up and the trial position in pt_new.
DEVFN void ord_map_update(const BaViewG& v, const BaOrdView& o, int np, double lambda) {
  for (int p = threadIdx.x; p < np; p += BA_THREADS) {
    double sum[3] = {0, 0, 0};
    for (int s = v.pt_offF[p]; s < v.pt_offF[p + 1]; s++) {
      const int info = v.sl_info[s];
      if (SL_STATE(info) != MS_OK) continue;
      const double AS1* cu = v.cam_up + 6 * SL_FORD(info);
      _Pragma("unroll") for (int c = 0; c < 3; c++) { double sx = 0; for (int r = 0; r < 6; r++) sx += OSL(W, r * 3 + c, s) * cu[r]; sum[c] += sx; }
    }
    const double eb[3] = {PT(pt_eb, 0, p), PT(pt_eb, 1, p), PT(pt_eb, 2, p)};
    double Vi[9];
    ba_vstar_inv(v, p, lambda, Vi);
    const double x[3] = {eb[0] - sum[0], eb[1] - sum[1], eb[2] - sum[2]};
    _Pragma("unroll") for (int r = 0; r < 3; r++) {
      const double u = Vi[r * 3] * x[0] + Vi[r * 3 + 1] * x[1] + Vi[r * 3 + 2] * x[2];
      o.up[3 * p + r] = u;
      v.pt_new[3 * p + r] = v.pt_pos[3 * p + r] + u;
    }
  }
}

// ba_block_solve_lds with the back-substitution of the reference-order mode: the elimination updates every element independently
// (the same multiply / subtract per element as lu_solve, whatever thread does it), but the fast path's back-substitution sums a row's
// products across lanes; here one lane subtracts them in ascending column order.
DEVFN bool ord_solve_lds(const double* S, double* E, int n, double* A, int* ired) {
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
    const int rem = n - k - 1, wid = ld - k - 1;
    for (int t = threadIdx.x; t < rem * wid; t += blockDim.x) {
      const int r = k + 1 + t / wid, c = k + 1 + t % wid;
      const double f = A[r * ld + k] * inv;
      A[r * ld + c] -= f * A[k * ld + c];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int k = n - 1; k >= 0; k--) {
      double s = A[k * ld + n];
      for (int c = k + 1; c < n; c++) s -= A[k * ld + c] * A[c * ld + n];
      A[k * ld + n] = s / A[k * ld + k];
    }
  __syncthreads();
  for (int t = threadIdx.x; t < n; t += blockDim.x) E[t] = A[t * ld + n];
  __syncthreads();
  return true;
}

DEVFN void ba_compute_ordered(const BaView& v_, const BaConfig& cfg, const BaOrdView& o) {
  const BaViewG v = ba_g(v_);
  __shared__ int ired[BA_WAVES];
  __shared__ int hist[768];
  __shared__ unsigned long long sel[1];
  __shared__ double sh_lambda, sh_factor, sh_sigma2, sh_cur_err, sh_new_err, sh_ssq;
  constexpr int LDS_SOLVE = BA_LDS_N * (BA_LDS_N + 1), LDS_LAYOUT = (2 * 4097 * (int)sizeof(int) + 7) / 8;
  __shared__ double lds_buf[LDS_SOLVE > LDS_LAYOUT ? LDS_SOLVE : LDS_LAYOUT];
  __shared__ double camL[12 * 128];                                 // every camera of a problem (at most 128 keyframes per map)
  __shared__ int sh_converged, sh_hitmax, sh_counter, sh_accepted, sh_error, sh_nout, sh_cache_valid, sh_next_nvalid;
  static_assert(sizeof(lds_buf) >= (65536 / 32) * sizeof(unsigned), "the LDS buffer also serves the erase's bit map");
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
  if (nfree > 64) { if (threadIdx.x == 0) { R->accepted = -1; R->converged = 0; R->hit_max = 0; R->counter = 0; R->n_outlier_meas = 0; } __syncthreads(); return; }
  ba_build_layout(v_, nc, np, ired, (int*)lds_buf);
  const int M = v.ch_n[3], MF = v.ch_n[2];
  for (int i = threadIdx.x; i < nm; i += BA_THREADS) o.of_logical[i] = -1;
  __syncthreads();
  for (int s = threadIdx.x; s < M; s += BA_THREADS) o.of_logical[v.sl_logical[s]] = s;
  __syncthreads();
  auto load_cams = [&](const Pose AS1* src) {
    __syncthreads();
    for (int t = threadIdx.x; t < nc * 12; t += BA_THREADS) camL[t] = ((const double AS1*)src)[t];
    __syncthreads();
  };

  while (!sh_converged && !sh_hitmax && !sh_error) {             // :153
    // ================= Do_LM_Step =================
    load_cams(v.cam_pose);
    const bool cached = sh_cache_valid != 0;                       // the previous step was accepted: FindNewError has left the squared errors
    int nvalid;
    if (cached) nvalid = sh_next_nvalid;
    else nvalid = ba_block_sum_i(ord_errors(v, cfg, o, M, 1.0, 0, camL), ired);
    if (nvalid == 0) { if (threadIdx.x == 0) sh_error = 1; __syncthreads(); break; }
    {                                                              // :220-227
      const double med = M > 4096 ? block_radix_select<16>(v.scratch, M, nvalid / 2, hist, sel)   // (big problems: 16 values per lane in flight, 64 per lane and sweep)
                                  : block_radix_select<8>(v.scratch, M, nvalid / 2, hist, sel);
      double s2 = tukey_sigma_squared(med, (unsigned long)nvalid);
      if (s2 < cfg.min_sigma2) s2 = cfg.min_sigma2;
      if (threadIdx.x == 0) sh_sigma2 = s2;
      __syncthreads();
    }
    const double sigma2 = sh_sigma2;
    ord_sweep(v, cfg, o, M, MF, sigma2, camL);
    __syncthreads();
    for (int task = threadIdx.x; task < 1 + 27 * nfree; task += BA_THREADS) {      // dCurrentError; U, epsilon_a
      const double x = ord_list_chain(v, o, nm, MF, task);
      if (task == 0) sh_cur_err = x;
      else {
        const int f = (task - 1) / 27, q = (task - 1) - 27 * f, j = v.free_cams[f];
        if (q < 21) { int r = 0, qq = q; while (qq > r) { qq -= r + 1; r++; } v.cam_U[36 * j + r * 6 + qq] = x; }
        else v.cam_ea[6 * j + (q - 21)] = x;
      }
    }
    ord_point_sums(v, o, np, MF);                                   // V, epsilon_b
    __syncthreads();
    // ---- inner loop over lambda (:326-501) ----
    if (threadIdx.x == 0) sh_new_err = sh_cur_err + 9999;
    __syncthreads();
    while (sh_new_err > sh_cur_err && !sh_converged && !sh_hitmax && !sh_error) {
      const double lambda = sh_lambda;
      ord_trial_operands(v, o, np, MF, lambda);
      __syncthreads();
      const int ntask = 27 * nfree + 36 * (nfree * (nfree - 1) / 2);
      for (int task = threadIdx.x; task < ntask; task += BA_THREADS) ord_schur_chain(v, o, np, nfree, nS, lambda, task);
      __syncthreads();
      bool solved = true;
      if (nS > 0 && nS <= BA_WSOLVE_N) {
        if (wave == 0) { const bool okw = ba_solve_wave(v_, nS); if (lane == 0) ired[0] = okw ? 1 : 0; }
        __syncthreads();
        solved = ired[0] != 0;
        __syncthreads();
      } else if (nS > 0) solved = nS <= BA_LDS_N ? ord_solve_lds((const double*)v.S, (double*)v.E, nS, lds_buf, ired) : ba_block_solve((double*)v.S, (double*)v.E, nS, ired);
      if (!solved) { if (threadIdx.x == 0) sh_error = 1; __syncthreads(); break; }
      for (int t = threadIdx.x; t < nS; t += BA_THREADS) v.cam_up[t] = v.E[t];
      __syncthreads();
      ord_map_update(v, o, np, lambda);
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
      load_cams(v.cam_new);
      const int nv_next = ba_block_sum_i(ord_errors(v, cfg, o, M, sh_sigma2, 1, camL), ired);   // FindNewError's terms (:537-561)
      if (threadIdx.x == 0) {                                          // dSumSquaredUpdate (:467-470): the camera updates, then the points'
        double acc = 0.0;
        for (int t = 0; t < nS; t++) { const double x = v.cam_up[t]; acc += x * x; }
        for (int t0 = 0; t0 < 3 * np; t0 += ORD_PF) {
          double x[ORD_PF];
          _Pragma("unroll") for (int u = 0; u < ORD_PF; u++) x[u] = t0 + u < 3 * np ? o.up[t0 + u] : 0.0;
          _Pragma("unroll") for (int u = 0; u < ORD_PF; u++) acc += x[u] * x[u];
        }
        sh_ssq = acc;
      }
      if (threadIdx.x == 64) sh_new_err = ord_list_chain(v, o, nm, MF, 0);   // dNewError in list order (another wavefront, beside the chain above)
      __syncthreads();
      if (threadIdx.x == 0) {
        sh_next_nvalid = nv_next;
        if (sh_ssq < cfg.convergence_limit) sh_converged = 1;
        if (sh_new_err > sh_cur_err) { sh_lambda = sh_lambda * sh_factor; sh_factor = sh_factor * 2; }   // ModifyLambda_BadStep :614-617
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
    {                                                                  // erase the outliers in list order (:517-528)
      const int no = ba_erase_outliers(v_, M, nm, sh_nout, (unsigned*)lds_buf, ired);
      __syncthreads();
      if (threadIdx.x == 0) sh_nout = no;
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    R->accepted = sh_error ? -1 : sh_accepted;                         // :170-177
    R->converged = sh_converged; R->hit_max = sh_hitmax; R->counter = sh_counter;
    R->sigma2 = sh_sigma2; R->lambda = sh_lambda; R->lambda_factor = sh_factor; R->n_outlier_meas = sh_nout;
  }
  __syncthreads();
}
