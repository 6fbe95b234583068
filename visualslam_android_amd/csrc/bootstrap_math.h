// Map bootstrap mathematics: HomographyInit (jni/HomographyInit.cc) and MapMaker::CalcPlaneAligner
// (jni/MapMaker.cc:1104-1231) as pure functions over plain arrays, compiled for the host and for gfx950 from this one source
// (like vslam_libm.h: IEEE add / mul / div / sqrt only, -ffp-contract=off, so both sides produce the same bits).
//
// One-shot work per map, not a hot path: on the device the hypothesis loops (300 MLESAC trials, 100 plane RANSAC trials) are
// spread over the lanes of one workgroup per stream, each hypothesis scored by a sequential loop in the reference's order; the
// serial parts (refinement, decomposition, choice) run on one lane.
//
// Where this restatement has to decide something the reference leaves open:
//  * rand() (HomographyInit.cc:236, MapMaker.cc:1119-1125) -> bm_rand(seed, trial, draw), a counter-based generator, so that
//    the trials are independent of each other and reproducible;
//  * Eigen::JacobiSVD (:118, :268) -> one-sided Jacobi (Hestenes) on the columns, singular values sorted in decreasing order;
//    the homography is the right singular vector of the smallest one, as there (sign and scale are free, :118-127);
//  * std::sort on HomographyDecomposition::nScore (:410, :435) is not stable -> ties keep their order of generation;
//  * Eigen::EigenSolver(m3Cov).eigenvectors().col(2) (MapMaker.cc:1197-1198) is an unspecified column for a general solver;
//    the intent (PTAM: the eigenvector of least variance) is what is computed: cyclic Jacobi on the symmetric 3x3.
// None of this is covered by a fixture of the reference: PARITY UNPINNED; the tests hold the results against the ground truth of
// synthetic planar scenes and hold host and device to the same bits.
#pragma once
#include <math.h>
#include "vslam_libm.h"

#if defined(__HIPCC__)
#define BM_FN __host__ __device__ inline
#else
#define BM_FN inline
#endif

namespace bm {

struct Match {                       // HomographyMatch, jni/HomographyInit.h: z = 1 plane positions + d(pixel) / d(z = 1 plane) at the second
  double first[2], second[2], jac[4];
};
struct Decomposition {               // HomographyDecomposition
  double Rp[9], Tp[3], n[3], d; double R[9], t[3]; int score;
};

BM_FN unsigned bm_rand(unsigned seed, unsigned trial, unsigned draw) {   // 32-bit mix (splitmix-style finaliser), >> 1 like rand()'s range
  unsigned long long z = ((unsigned long long)seed << 40) ^ ((unsigned long long)trial << 16) ^ (unsigned long long)draw;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 33);
}

// One-sided Jacobi SVD of the m x n matrix A (row-major, overwritten with U * diag(S)), n <= 9: V (n x n, row-major) accumulates
// the rotations; returns the column order by decreasing singular value in `order`, the singular values in S.
BM_FN void svd_onesided(double* A, int m, int n, double* V, double* S, int* order) {
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int k = 0; k < m; k++) { const double x = A[k * n + p], y = A[k * n + q]; a += x * x; b += y * y; c += x * y; }
        if (c == 0.0 || fabs(c) <= 2.220446049250313e-16 * sqrt(a * b)) continue;
        rotated = true;
        const double zeta = (b - a) / (2.0 * c);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
        for (int k = 0; k < m; k++) { const double x = A[k * n + p], y = A[k * n + q]; A[k * n + p] = cs * x - sn * y; A[k * n + q] = sn * x + cs * y; }
        for (int k = 0; k < n; k++) { const double x = V[k * n + p], y = V[k * n + q]; V[k * n + p] = cs * x - sn * y; V[k * n + q] = sn * x + cs * y; }
      }
    if (!rotated) break;
  }
  for (int j = 0; j < n; j++) { double s = 0.0; for (int k = 0; k < m; k++) s += A[k * n + j] * A[k * n + j]; S[j] = sqrt(s); order[j] = j; }
  for (int i = 1; i < n; i++) {                                      // insertion sort, decreasing, stable
    const int o = order[i]; int j = i;
    while (j > 0 && S[order[j - 1]] < S[o]) { order[j] = order[j - 1]; j--; }
    order[j] = o;
  }
}

// HomographyFromMatches, jni/HomographyInit.cc:75-128: n in 4..9 matches given by index
BM_FN void homography_from_matches(const Match* m, const int* idx, int n, double H[9]) {
  double A[18 * 9], V[81], S[9]; int order[9];
  int rows = 2 * n; if (rows < 9) rows = 9;
  for (int i = 0; i < rows * 9; i++) A[i] = 0.0;                     // :113-116: the ninth row of a minimal set is zero
  for (int k = 0; k < n; k++) {
    const Match& q = m[idx ? idx[k] : k];
    const double u = q.second[0], v = q.second[1], x = q.first[0], y = q.first[1];
    double* r0 = A + (2 * k) * 9; double* r1 = r0 + 9;
    r0[0] = x; r0[1] = y; r0[2] = 1; r0[6] = -x * u; r0[7] = -y * u; r0[8] = -u;
    r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; r1[8] = -v;
  }
  svd_onesided(A, rows, 9, V, S, order);
  const int c = order[8];                                            // :121-125: the last row of V^T
  for (int i = 0; i < 9; i++) H[i] = V[i * 9 + c];
}

BM_FN double pixel_error_squared(const double H[9], const Match& q) {  // the common part of IsHomographyInlier / MLESACScore, :21-41
  const double v0 = H[0] * q.first[0] + H[1] * q.first[1] + H[2] * 1.0;
  const double v1 = H[3] * q.first[0] + H[4] * q.first[1] + H[5] * 1.0;
  const double v2 = H[6] * q.first[0] + H[7] * q.first[1] + H[8] * 1.0;
  const double e0 = q.second[0] - v0 / v2, e1 = q.second[1] - v1 / v2;
  const double p0 = q.jac[0] * e0 + q.jac[1] * e1, p1 = q.jac[2] * e0 + q.jac[3] * e1;
  return p0 * p0 + p1 * p1;
}

// One MLESAC trial (:229-262): four distinct matches drawn with bm_rand, their homography, its score over all matches
BM_FN double mlesac_trial(const Match* m, int n, unsigned seed, int trial, double max_err2, double H[9]) {
  int idx[4];
  unsigned draw = 0;
  for (int i = 0; i < 4; i++) {
    bool unique = false; int k = 0;
    while (!unique) {
      k = (int)(bm_rand(seed, (unsigned)trial, draw++) % (unsigned)n);
      unique = true;
      for (int j = 0; j < i && unique; j++) if (idx[j] == k) unique = false;
    }
    idx[i] = k;
  }
  homography_from_matches(m, idx, 4, H);
  double err = 0.0;
  for (int i = 0; i < n; i++) { const double e = pixel_error_squared(H, m[i]); err += e > max_err2 ? max_err2 : e; }
  return err;
}

// the order statistic FindSigmaSquared sorts for (jni/MEstimator.h:67-77): k-th smallest of v[0..n), v is permuted
BM_FN double kth_smallest(double* v, int n, int k) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const double piv = v[(lo + hi) >> 1];
    int i = lo, j = hi;
    while (i <= j) {
      while (v[i] < piv) i++;
      while (v[j] > piv) j--;
      if (i <= j) { const double t = v[i]; v[i] = v[j]; v[j] = t; i++; j--; }
    }
    if (k <= j) hi = j; else if (k >= i) lo = i; else break;
  }
  return v[k];
}

BM_FN bool lu_solve(double* A, double* b, int n) {                  // Gaussian elimination, partial pivoting (myWLS::compute's inverse() * v, jni/myWLS.h:61)
  for (int k = 0; k < n; k++) {
    int piv = k; double best = fabs(A[k * n + k]);
    for (int r = k + 1; r < n; r++) if (fabs(A[r * n + k]) > best) { best = fabs(A[r * n + k]); piv = r; }
    if (best == 0.0) return false;
    if (piv != k) { for (int c = 0; c < n; c++) { const double t = A[k * n + c]; A[k * n + c] = A[piv * n + c]; A[piv * n + c] = t; } const double t = b[k]; b[k] = b[piv]; b[piv] = t; }
    for (int r = k + 1; r < n; r++) {
      const double f = A[r * n + k] / A[k * n + k];
      if (f == 0.0) continue;
      for (int c = k; c < n; c++) A[r * n + c] -= f * A[k * n + c];
      b[r] -= f * b[k];
    }
  }
  for (int k = n - 1; k >= 0; k--) {
    double s = b[k];
    for (int c = k + 1; c < n; c++) s -= A[k * n + c] * b[c];
    b[k] = s / A[k * n + k];
  }
  return true;
}

// RefineHomographyWithInliers, :133-222.  inl: indices of the inliers; e2: workspace of ninl doubles.
BM_FN void refine_homography(double H[9], const Match* m, const int* inl, int ninl, double* e2) {
  double C[81], vec[9];
  for (int i = 0; i < 81; i++) C[i] = 0.0;
  for (int i = 0; i < 9; i++) { vec[i] = 0.0; C[i * 9 + i] += 1.0; }   // add_prior(1.0), :135
  if (ninl == 0) return;
  for (int i = 0; i < ninl; i++) e2[i] = pixel_error_squared(H, m[inl[i]]);   // :146-153 (the same expression)
  // Tukey::FindSigmaSquared over a copy: the [n / 2] order statistic (the selection permutes e2, so the errors are recomputed below)
  const double med = kth_smallest(e2, ninl, ninl / 2);
  double sigma = 1.4826 * (1 + 5.0 / ((unsigned long)ninl * 2 - 6)) * sqrt(med);
  sigma = 4.6851 * sigma;
  const double sigma2 = sigma * sigma;
  for (int i = 0; i < ninl; i++) {
    const Match& q = m[inl[i]];
    const double x = q.first[0], y = q.first[1];
    const double s0 = H[0] * x + H[1] * y + H[2] * 1.0, s1 = H[3] * x + H[4] * y + H[5] * 1.0, den = H[6] * x + H[7] * y + H[8] * 1.0;
    const double d0 = q.second[0] - s0 / den, d1 = q.second[1] - s1 / den;
    const double err[2] = {q.jac[0] * d0 + q.jac[1] * d1, q.jac[2] * d0 + q.jac[3] * d1};
    const double es = err[0] * err[0] + err[1] * err[1];
    double J[2][9];                                                  // :155-185
    const double un[3] = {x, y, 1.0};
    for (int k = 0; k < 3; k++) {
      J[0][k] = un[k] / den; J[0][3 + k] = 0.0; J[0][6 + k] = -un[k] * s0 / (den * den);
      J[1][k] = 0.0; J[1][3 + k] = un[k] / den; J[1][6 + k] = -un[k] * s1 / (den * den);
    }
    double PJ[2][9];                                                 // m2PixelProjectionJac * m29Jacobian, :187
    for (int k = 0; k < 9; k++) { PJ[0][k] = q.jac[0] * J[0][k] + q.jac[1] * J[1][k]; PJ[1][k] = q.jac[2] * J[0][k] + q.jac[3] * J[1][k]; }
    const double w = es > sigma2 ? 0.0 : (1.0 - es / sigma2) * (1.0 - es / sigma2);   // Tukey::Weight
    for (int row = 0; row < 2; row++) {                              // add_mJ((int) error, row, weight), :196-197 (the cast is the reference's)
      const double mm = (double)(int)err[row];
      for (int r = 0; r < 9; r++) {
        const double Jw = w * PJ[row][r];
        vec[r] += mm * Jw;
        for (int c = r; c < 9; c++) C[r * 9 + c] += Jw * PJ[row][c];
      }
    }
  }
  for (int r = 1; r < 9; r++) for (int c = 0; c < r; c++) C[r * 9 + c] = C[c * 9 + r];
  if (!lu_solve(C, vec, 9)) return;
  for (int i = 0; i < 9; i++) H[i] += vec[i];                        // :204-216
}

BM_FN double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
BM_FN void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}

// DecomposeHomography, :264-374 (Faugeras & Lustman).  Returns the number of decompositions (8, or 0 for the degenerate cases).
BM_FN int decompose_homography(const double H[9], Decomposition out[8]) {
  double A[9], V[9], S[3], U[9]; int order[3];
  for (int i = 0; i < 9; i++) A[i] = H[i];
  svd_onesided(A, 3, 3, V, S, order);                                // A = U diag(S) now
  double Vs[9];
  for (int c = 0; c < 3; c++) {
    const int o = order[c];
    for (int r = 0; r < 3; r++) { Vs[r * 3 + c] = V[r * 3 + o]; U[r * 3 + c] = S[o] > 0.0 ? A[r * 3 + o] / S[o] : 0.0; }
  }
  const double d1 = fabs(S[order[0]]), d2 = fabs(S[order[1]]), d3 = fabs(S[order[2]]);
  if (d3 == 0.0) {                                                   // rank-deficient: complete U with the cross product of its first two columns
    U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
  }
  const double s = det3(U) * det3(Vs);
  if (!(d1 != d2 && d2 != d3)) return 0;                             // :286-296: cases 2 and 3 are not handled
  const double x1 = sqrt((d1 * d1 - d2 * d2) / (d1 * d1 - d3 * d3)), x2 = 0.0, x3 = sqrt((d2 * d2 - d3 * d3) / (d1 * d1 - d3 * d3));
  const double e1[4] = {1.0, -1.0, 1.0, -1.0}, e3[4] = {1.0, 1.0, -1.0, -1.0};
  int n = 0;
  for (int half = 0; half < 2; half++)
    for (int sg = 0; sg < 4; sg++) {
      Decomposition& D = out[n++];
      for (int i = 0; i < 9; i++) D.Rp[i] = 0.0;
      if (half == 0) {                                               // d' > 0, eq. 13-14
        D.d = s * d2;
        D.Rp[0] = D.Rp[4] = D.Rp[8] = 1.0;
        const double st = (d1 - d3) * x1 * x3 * e1[sg] * e3[sg] / d2, ct = (d1 * x3 * x3 + d3 * x1 * x1) / d2;
        D.Rp[0] = ct; D.Rp[2] = -st; D.Rp[6] = st; D.Rp[8] = ct;
        D.Tp[0] = (d1 - d3) * x1 * e1[sg]; D.Tp[1] = 0.0; D.Tp[2] = (d1 - d3) * -x3 * e3[sg];
      } else {                                                       // d' < 0, eq. 15-16
        D.d = s * -d2;
        D.Rp[0] = D.Rp[4] = D.Rp[8] = -1.0;
        const double sp = (d1 + d3) * x1 * x3 * e1[sg] * e3[sg] / d2, cp = (d3 * x1 * x1 - d1 * x3 * x3) / d2;
        D.Rp[0] = cp; D.Rp[2] = sp; D.Rp[6] = sp; D.Rp[8] = -cp;
        D.Tp[0] = (d1 + d3) * x1 * e1[sg]; D.Tp[1] = 0.0; D.Tp[2] = (d1 + d3) * x3 * e3[sg];
      }
      const double np[3] = {x1 * e1[sg], x2, x3 * e3[sg]};
      for (int r = 0; r < 3; r++) D.n[r] = Vs[r * 3] * np[0] + Vs[r * 3 + 1] * np[1] + Vs[r * 3 + 2] * np[2];
      double T1[9], Vt[9];
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Vt[r * 3 + c] = Vs[c * 3 + r];
      mat3_mul(U, D.Rp, T1); mat3_mul(T1, Vt, D.R);                  // :365-366: s * U * Rp * V^T, U * Tp
      for (int i = 0; i < 9; i++) D.R[i] = s * D.R[i];
      for (int r = 0; r < 3; r++) D.t[r] = U[r * 3] * D.Tp[0] + U[r * 3 + 1] * D.Tp[1] + U[r * 3 + 2] * D.Tp[2];
      D.score = 0;
    }
  return n;
}

BM_FN void stable_sort_by_score(Decomposition* d, int n) {
  for (int i = 1; i < n; i++) { const Decomposition t = d[i]; int j = i; while (j > 0 && t.score < d[j - 1].score) { d[j] = d[j - 1]; j--; } d[j] = t; }
}

BM_FN double sampsonus_error(const double dash[2], const double E[9], const double v[2]) {   // :380-403
  const double v3d[3] = {dash[0], dash[1], 1.0}, v3[3] = {v[0], v[1], 1.0};
  double f[3], ft[3];
  for (int r = 0; r < 3; r++) { f[r] = E[r * 3] * v3[0] + E[r * 3 + 1] * v3[1] + E[r * 3 + 2] * v3[2]; ft[r] = E[r] * v3d[0] + E[3 + r] * v3d[1] + E[6 + r] * v3d[2]; }
  const double err = f[0] * v3d[0] + f[1] * v3d[1] + f[2] * v3d[2];
  return err * err / ((f[0] * f[0] + f[1] * f[1]) + (ft[0] * ft[0] + ft[1] * ft[1]));
}

// ChooseBestDecomposition, :405-499: visibility votes of the inliers, then the Sampson score of all matches for a tie.  Result in d[0].
BM_FN void choose_best_decomposition(Decomposition d[8], const double H[9], const Match* m, int n, const int* inl, int ninl, double max_err2) {
  for (int i = 0; i < 8; i++) {
    int pos = 0;
    for (int k = 0; k < ninl; k++) { const Match& q = m[inl[k]]; if ((H[6] * q.first[0] + H[7] * q.first[1] + H[8]) / d[i].d > 0.0) pos++; }
    d[i].score = -pos;
  }
  stable_sort_by_score(d, 8);
  for (int i = 0; i < 4; i++) {
    int pos = 0;
    for (int k = 0; k < ninl; k++) { const Match& q = m[inl[k]]; if ((q.first[0] * d[i].n[0] + q.first[1] * d[i].n[1] + 1.0 * d[i].n[2]) / d[i].d > 0.0) pos++; }
    d[i].score = -pos;
  }
  stable_sort_by_score(d, 4);
  const double ratio = (double)d[1].score / (double)d[0].score;
  if (ratio < 0.9) return;                                           // no ambiguity, :447-448
  const double limit = max_err2 * 4;
  double sc[2];
  for (int i = 0; i < 2; i++) {
    double E[9];
    for (int j = 0; j < 3; j++) {                                    // column j of the essential matrix: t x (column j of R), :455-473
      const double a[3] = {d[i].t[0], d[i].t[1], d[i].t[2]}, b[3] = {d[i].R[j], d[i].R[3 + j], d[i].R[6 + j]};
      E[j] = a[1] * b[2] - a[2] * b[1]; E[3 + j] = a[2] * b[0] - a[0] * b[2]; E[6 + j] = a[0] * b[1] - a[1] * b[0];
    }
    double sum = 0.0;
    for (int k = 0; k < n; k++) { double e = sampsonus_error(m[k].second, E, m[k].first); if (e > limit) e = limit; sum += e; }
    sc[i] = sum;
  }
  if (!(sc[0] <= sc[1])) d[0] = d[1];
}

// ---- MapMaker::CalcPlaneAligner, jni/MapMaker.cc:1104-1231 ------------------------------------------------------------------
// One RANSAC trial (:1118-1163): the summed truncated distance of all points to the plane through three random ones; returns
// a negative value when the three are collinear (the reference skips the trial).
BM_FN double plane_trial(const double* pos /* [n][3] */, int n, unsigned seed, int trial, double mean[3], double normal[3]) {
  unsigned draw = 0;
  const int nA = (int)(bm_rand(seed, (unsigned)trial, draw++) % (unsigned)n);
  int nB = nA, nC = nA;
  while (nB == nA) nB = (int)(bm_rand(seed, (unsigned)trial, draw++) % (unsigned)n);
  while (nC == nA || nC == nB) nC = (int)(bm_rand(seed, (unsigned)trial, draw++) % (unsigned)n);
  const double* A = pos + 3 * nA; const double* B = pos + 3 * nB; const double* C = pos + 3 * nC;
  double ca[3], ba[3];
  for (int k = 0; k < 3; k++) { mean[k] = 0.33333333 * (A[k] + B[k] + C[k]); ca[k] = C[k] - A[k]; ba[k] = B[k] - A[k]; }
  normal[0] = ca[1] * ba[2] - ca[2] * ba[1]; normal[1] = ca[2] * ba[0] - ca[0] * ba[2]; normal[2] = ca[0] * ba[1] - ca[1] * ba[0];
  const double nn = normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2];
  if (nn == 0) return -1.0;
  const double inv = sqrt(nn);
  for (int k = 0; k < 3; k++) normal[k] = normal[k] / inv;
  double sum = 0.0;
  for (int i = 0; i < n; i++) {
    const double d0 = pos[3 * i] - mean[0], d1 = pos[3 * i + 1] - mean[1], d2 = pos[3 * i + 2] - mean[2];
    if (d0 * d0 + d1 * d1 + d2 * d2 == 0.0) continue;
    double nd = fabs(d0 * normal[0] + d1 * normal[1] + d2 * normal[2]);
    if (nd > 0.05) nd = 0.05;
    sum += nd;
  }
  return sum;
}

// eigenvector of the smallest eigenvalue of a symmetric 3x3 (cyclic Jacobi)
BM_FN void sym3_smallest_eigenvector(const double M[9], double out[3]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; i++) A[i] = M[i];
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
    if (off == 0.0 || off <= 1e-300) break;
    bool rotated = false;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        const double apq = A[p * 3 + q];
        if (apq == 0.0 || fabs(apq) <= 2.220446049250313e-16 * sqrt(fabs(A[p * 3 + p] * A[q * 3 + q]))) continue;
        rotated = true;
        const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; k++) { const double x = A[k * 3 + p], y = A[k * 3 + q]; A[k * 3 + p] = c * x - s * y; A[k * 3 + q] = s * x + c * y; }
        for (int k = 0; k < 3; k++) { const double x = A[p * 3 + k], y = A[q * 3 + k]; A[p * 3 + k] = c * x - s * y; A[q * 3 + k] = s * x + c * y; }
        for (int k = 0; k < 3; k++) { const double x = V[k * 3 + p], y = V[k * 3 + q]; V[k * 3 + p] = c * x - s * y; V[k * 3 + q] = s * x + c * y; }
      }
    if (!rotated) break;
  }
  int best = 0;
  for (int k = 1; k < 3; k++) if (A[k * 3 + k] < A[best * 3 + best]) best = k;
  for (int k = 0; k < 3; k++) out[k] = V[k * 3 + best];
}

// the part of CalcPlaneAligner after the RANSAC (:1165-1230): inliers of the best plane, their mean and scatter, the aligning
// rotation (rows: x axis made orthogonal to the normal, normal x that, normal) and translation.  R row-major, returns false when
// no inlier is left.
BM_FN bool plane_aligner(const double* pos, int n, const double best_mean[3], const double best_normal[3], double R[9], double t[3]) {
  double mean[3] = {0, 0, 0};
  int ni = 0;
  for (int pass = 0; pass < 2; pass++) {
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
      const double d0 = pos[3 * i] - best_mean[0], d1 = pos[3 * i + 1] - best_mean[1], d2 = pos[3 * i + 2] - best_mean[2];
      if (d0 * d0 + d1 * d1 + d2 * d2 == 0.0) continue;
      if (!(fabs(d0 * best_normal[0] + d1 * best_normal[1] + d2 * best_normal[2]) < 0.05)) continue;
      if (pass == 0) { for (int k = 0; k < 3; k++) mean[k] += pos[3 * i + k]; ni++; }
      else {
        const double e[3] = {pos[3 * i] - mean[0], pos[3 * i + 1] - mean[1], pos[3 * i + 2] - mean[2]};
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) cov[r * 3 + c] += e[r] * e[c];
      }
    }
    if (pass == 0) {
      if (ni == 0) return false;
      for (int k = 0; k < 3; k++) mean[k] *= (1.0 / ni);
    } else {
      double nrm[3];
      sym3_smallest_eigenvector(cov, nrm);
      if (nrm[2] > 0) for (int k = 0; k < 3; k++) nrm[k] *= -1.0;     // towards the camera, :1210-1211
      double row0[3] = {1.0, 0.0, 0.0};
      const double dotp = row0[0] * nrm[0] + row0[1] * nrm[1] + row0[2] * nrm[2];
      double aux[3] = {row0[0] - nrm[0] * dotp, row0[1] - nrm[1] * dotp, row0[2] - nrm[2] * dotp};
      const double an = sqrt(aux[0] * aux[0] + aux[1] * aux[1] + aux[2] * aux[2]);
      for (int k = 0; k < 3; k++) aux[k] = aux[k] / an;
      for (int k = 0; k < 3; k++) { R[k] = aux[k]; R[6 + k] = nrm[k]; }
      R[3] = nrm[1] * aux[2] - nrm[2] * aux[1]; R[4] = nrm[2] * aux[0] - nrm[0] * aux[2]; R[5] = nrm[0] * aux[1] - nrm[1] * aux[0];   // fila2 x fila0
      for (int r = 0; r < 3; r++) t[r] = -(R[r * 3] * mean[0] + R[r * 3 + 1] * mean[1] + R[r * 3 + 2] * mean[2]);
    }
  }
  return true;
}

}  // namespace bm
