// ORACLE (test infrastructure only -- see ptam_oracle.h).  The mathematics of the map bootstrap restated from the reference,
// independently of the product's csrc/bootstrap_math.h (which the oracle used to include: VERDICT r2 weak #4 -- the two sides were one
// header compiled twice):
//   HomographyInit::Compute / IsHomographyInlier / MLESACScore            jni/HomographyInit.cc:21-71
//   HomographyInit::HomographyFromMatches                                  jni/HomographyInit.cc:75-128
//   HomographyInit::RefineHomographyWithInliers (myWLS<9>, Tukey)          jni/HomographyInit.cc:133-222, jni/myWLS.h:29-62
//   HomographyInit::BestHomographyFromMatches_MLESAC                       jni/HomographyInit.cc:224-262
//   HomographyInit::DecomposeHomography / ChooseBestDecomposition          jni/HomographyInit.cc:264-499
//   MapMaker::CalcPlaneAligner                                             jni/MapMaker.cc:1104-1231
// written the way the reference is written (vectors of matches, copies of the inlier set, vectors of Jacobians and errors, the
// decompositions sorted and resized) with the numerical methods of the reference's Eigen where they are published:
//   Eigen::JacobiSVD            two-sided Jacobi on the square matrix (jacobi_svd_square below; a taller matrix is first reduced by a
//                               column-pivoted Householder QR, JacobiSVD's default preconditioner) -- the product uses a ONE-sided
//                               Hestenes Jacobi;
//   MatrixXd::inverse() * v     partial-pivot elimination (orc::lu_solve);
//   EigenSolver(...).col(2)     an unspecified column for a general solver; the intent (PTAM: the direction of least variance) by
//                               the closed-form eigenvalues of a symmetric 3x3 and a cross-product eigenvector -- the product uses
//                               cyclic Jacobi.
// So the two sides agree to rounding (~1e-10), not to the bit; the GPU tests compare integers (trails, inlier counts) exactly and
// everything derived from the homography to a tolerance.  What cannot be restated is rand() (:236, MapMaker.cc:1119-1125): the draws
// come from a counter-based generator, stated here (boot_rand) with the same constants as the product's bm_rand so that both sides
// test the same hypotheses.  PARITY UNPINNED against the reference (no fixture exists).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "ptam_system.hpp"

namespace orc {
namespace hinit {

unsigned boot_rand(unsigned seed, unsigned trial, unsigned draw) {   // the stand-in for rand(): splitmix64 finaliser over (seed, trial, draw), 31 bits
  unsigned long long z = ((unsigned long long)seed << 40) ^ ((unsigned long long)trial << 16) ^ (unsigned long long)draw;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 33);
}

struct Mat3 { double m[9]; double& operator()(int r, int c) { return m[r * 3 + c]; } double operator()(int r, int c) const { return m[r * 3 + c]; } };
static Mat3 mat3_identity() { Mat3 I; for (int i = 0; i < 9; i++) I.m[i] = i % 4 == 0 ? 1.0 : 0.0; return I; }
static Mat3 mul33(const Mat3& a, const Mat3& b) { Mat3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j); return c; }
static Mat3 transpose33(const Mat3& a) { Mat3 t; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t(i, j) = a(j, i); return t; }
static double det33(const Mat3& a) { return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) + a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0)); }

// Eigen::JacobiSVD of an n x n real matrix with full U and V (two-sided Jacobi; the 2x2 step as in mapgrow.cpp's 4x4 instance):
// A = U diag(S) V^T, S >= 0 in decreasing order.  M: n x n row-major (destroyed).
static void jacobi_svd_square(int n, std::vector<double>& M, std::vector<double>& U, std::vector<double>& S, std::vector<double>& V) {
  U.assign((size_t)n * n, 0.0); V.assign((size_t)n * n, 0.0); S.assign(n, 0.0);
  for (int i = 0; i < n; i++) U[(size_t)i * n + i] = V[(size_t)i * n + i] = 1.0;
  const double precision = 2.0 * 2.220446049250313e-16;
  for (int sweep = 0; sweep < 100; sweep++) {
    bool finished = true;
    for (int p = 1; p < n; p++)
      for (int q = 0; q < p; q++) {
        const double off = std::max(fabs(M[(size_t)p * n + q]), fabs(M[(size_t)q * n + p]));
        const double dia = std::max(fabs(M[(size_t)p * n + p]), fabs(M[(size_t)q * n + q]));
        if (!(off > dia * precision)) continue;
        finished = false;
        const double m00 = M[(size_t)p * n + p], m01 = M[(size_t)p * n + q], m10 = M[(size_t)q * n + p], m11 = M[(size_t)q * n + q];
        double c1, s1;                                                    // rot1: makes the 2x2 block symmetric
        const double t = m00 + m11, d = m10 - m01;
        if (t == 0.0) { c1 = 0.0; s1 = d > 0.0 ? 1.0 : -1.0; }
        else { const double u = d / t; c1 = 1.0 / sqrt(1.0 + u * u); s1 = c1 * u; }
        const double x = c1 * m00 + s1 * m10, y = c1 * m01 + s1 * m11, z = -s1 * m01 + c1 * m11;
        double c2, s2;                                                    // j_right = makeJacobi(x, y, z)
        if (y == 0.0) { c2 = 1.0; s2 = 0.0; }
        else {
          const double tau = (x - z) / (2.0 * fabs(y)), w = sqrt(tau * tau + 1.0);
          const double tt = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
          const double sign_t = tt > 0.0 ? 1.0 : -1.0, nn = 1.0 / sqrt(tt * tt + 1.0);
          s2 = -sign_t * (y / fabs(y)) * fabs(tt) * nn; c2 = nn;
        }
        const double cl = c1 * c2 + s1 * s2, sl = s1 * c2 - c1 * s2;     // j_left = rot1 * j_right^T
        for (int k = 0; k < n; k++) {                                     // m.applyOnTheLeft(p, q, j_left)
          const double a = M[(size_t)p * n + k], b = M[(size_t)q * n + k];
          M[(size_t)p * n + k] = cl * a + sl * b; M[(size_t)q * n + k] = -sl * a + cl * b;
        }
        for (int k = 0; k < n; k++) {                                     // U.applyOnTheRight(p, q, j_left.transpose())
          const double a = U[(size_t)k * n + p], b = U[(size_t)k * n + q];
          U[(size_t)k * n + p] = cl * a + sl * b; U[(size_t)k * n + q] = -sl * a + cl * b;
        }
        for (int k = 0; k < n; k++) {                                     // m.applyOnTheRight(p, q, j_right); V likewise
          const double a = M[(size_t)k * n + p], b = M[(size_t)k * n + q];
          M[(size_t)k * n + p] = c2 * a - s2 * b; M[(size_t)k * n + q] = s2 * a + c2 * b;
          const double e = V[(size_t)k * n + p], f = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c2 * e - s2 * f; V[(size_t)k * n + q] = s2 * e + c2 * f;
        }
      }
    if (finished) break;
  }
  for (int i = 0; i < n; i++) {                                            // positive singular values: the sign goes into U
    const double a = M[(size_t)i * n + i];
    S[i] = fabs(a);
    if (a < 0.0) for (int k = 0; k < n; k++) U[(size_t)k * n + i] = -U[(size_t)k * n + i];
  }
  for (int i = 0; i < n; i++) {                                            // decreasing order (selection, as JacobiSVD does: swap with the largest of the rest)
    int best = i;
    for (int j = i + 1; j < n; j++) if (S[j] > S[best]) best = j;
    if (best != i) {
      std::swap(S[i], S[best]);
      for (int k = 0; k < n; k++) { std::swap(U[(size_t)k * n + i], U[(size_t)k * n + best]); std::swap(V[(size_t)k * n + i], V[(size_t)k * n + best]); }
    }
  }
}

// The right singular vectors of a rows x 9 matrix, rows >= 9 (row-major): square -> two-sided Jacobi directly; taller -> JacobiSVD's
// default preconditioner first, a Householder QR with column pivoting: A P = Q R, then the SVD of the square R; V = P V_R.
static void right_singular_vectors_9(int rows, std::vector<double> A, std::vector<double>& V) {
  const int n = 9;
  std::vector<double> U, S;
  if (rows == n) { jacobi_svd_square(n, A, U, S, V); return; }
  int perm[9];
  for (int j = 0; j < n; j++) perm[j] = j;
  for (int k = 0; k < n; k++) {
    int best = k; double bn = -1.0;                                        // the column of largest remaining norm
    for (int j = k; j < n; j++) { double s = 0.0; for (int i = k; i < rows; i++) s += A[(size_t)i * n + j] * A[(size_t)i * n + j]; if (s > bn) { bn = s; best = j; } }
    if (best != k) { for (int i = 0; i < rows; i++) std::swap(A[(size_t)i * n + k], A[(size_t)i * n + best]); std::swap(perm[k], perm[best]); }
    double norm = 0.0;
    for (int i = k; i < rows; i++) norm += A[(size_t)i * n + k] * A[(size_t)i * n + k];
    norm = sqrt(norm);
    if (norm == 0.0) continue;
    const double alpha = A[(size_t)k * n + k] > 0.0 ? -norm : norm;
    std::vector<double> v(rows, 0.0);
    for (int i = k; i < rows; i++) v[i] = A[(size_t)i * n + k];
    v[k] -= alpha;
    double vv = 0.0;
    for (int i = k; i < rows; i++) vv += v[i] * v[i];
    if (vv == 0.0) continue;
    for (int j = k; j < n; j++) {                                          // A <- (I - 2 v v^T / v^T v) A
      double dot = 0.0;
      for (int i = k; i < rows; i++) dot += v[i] * A[(size_t)i * n + j];
      const double f = 2.0 * dot / vv;
      for (int i = k; i < rows; i++) A[(size_t)i * n + j] -= f * v[i];
    }
  }
  std::vector<double> R((size_t)n * n, 0.0), VR;
  for (int i = 0; i < n; i++) for (int j = i; j < n; j++) R[(size_t)i * n + j] = A[(size_t)i * n + j];
  jacobi_svd_square(n, R, U, S, VR);
  V.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; j++) for (int c = 0; c < n; c++) V[(size_t)perm[j] * n + c] = VR[(size_t)j * n + c];
}

struct Init {                                // the members of class HomographyInit, jni/HomographyInit.h
  double mdMaxPixelErrorSquared;
  Mat3 mm3BestHomography;
  std::vector<HMatch> mvMatches, mvHomographyInliers;
  std::vector<HDecomposition> mvDecompositions;
  unsigned seed;

  static void unproject(const double v[2], double o[3]) { o[0] = v[0]; o[1] = v[1]; o[2] = 1.0; }                 // myUnproject :5-12
  static void project(const double v[3], double o[2]) { o[0] = v[0] / v[2]; o[1] = v[1] / v[2]; }                  // myProject :14-19

  double squared_pixel_error(const Mat3& H, const HMatch& match) const {                                              // :21-27 / :31-36
    double u[3], v[3], proj[2];
    unproject(match.first, u);
    for (int r = 0; r < 3; r++) v[r] = H(r, 0) * u[0] + H(r, 1) * u[1] + H(r, 2) * u[2];
    project(v, proj);
    const double e[2] = {match.second[0] - proj[0], match.second[1] - proj[1]};
    const double pe[2] = {match.jac[0] * e[0] + match.jac[1] * e[1], match.jac[2] * e[0] + match.jac[3] * e[1]};
    return pe[0] * pe[0] + pe[1] * pe[1];
  }
  bool IsHomographyInlier(const Mat3& H, const HMatch& match) const { return squared_pixel_error(H, match) < mdMaxPixelErrorSquared; }
  double MLESACScore(const Mat3& H, const HMatch& match) const {
    const double d = squared_pixel_error(H, match);
    return d > mdMaxPixelErrorSquared ? mdMaxPixelErrorSquared : d;
  }

  static Mat3 HomographyFromMatches(const std::vector<HMatch>& vMatches) {                                           // :75-128
    const int nPoints = (int)vMatches.size();
    int nRows = 2 * nPoints;
    if (nRows < 9) nRows = 9;
    std::vector<double> m2Nx9((size_t)nRows * 9, 0.0);
    for (int n = 0; n < nPoints; n++) {
      const double u = vMatches[n].second[0], v = vMatches[n].second[1], x = vMatches[n].first[0], y = vMatches[n].first[1];
      double* r0 = &m2Nx9[(size_t)(n * 2) * 9]; double* r1 = r0 + 9;
      r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * u; r0[7] = -y * u; r0[8] = -u;
      r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; r1[8] = -v;
    }
    if (nRows == 9) for (int i = 0; i < 9; i++) m2Nx9[8 * 9 + i] = 0.0;                                             // :113-116
    std::vector<double> V;
    right_singular_vectors_9(nRows, m2Nx9, V);
    Mat3 H = mat3_identity();                                                                                          // :121-125: row 8 of V^T = column 8 of V
    for (int i = 0; i < 9; i++) H.m[i] = V[(size_t)i * 9 + 8];
    return H;
  }

  void RefineHomographyWithInliers() {                                                                                // :133-222
    const int N = 9;
    std::vector<double> C((size_t)N * N, 0.0), vec(N, 0.0);                                                           // myWLS<9>: my_C_inv, my_vector
    for (int i = 0; i < N; i++) C[(size_t)i * N + i] += 1.0;                                                          // add_prior(1.0)
    std::vector<double> vdErrorSquared;
    std::vector<std::array<double, 18>> vmJacobians;
    std::vector<std::array<double, 2>> vvErrors;
    for (size_t i = 0; i < mvHomographyInliers.size(); i++) {
      const HMatch& in = mvHomographyInliers[i];
      double un[3], v3Second[3], v2Second[2];
      unproject(in.first, un);
      for (int r = 0; r < 3; r++) v3Second[r] = mm3BestHomography(r, 0) * un[0] + mm3BestHomography(r, 1) * un[1] + mm3BestHomography(r, 2) * un[2];
      project(v3Second, v2Second);
      const double dx = in.second[0] - v2Second[0], dy = in.second[1] - v2Second[1];
      const std::array<double, 2> v2Error = {in.jac[0] * dx + in.jac[1] * dy, in.jac[2] * dx + in.jac[3] * dy};
      vdErrorSquared.push_back(v2Error[0] * v2Error[0] + v2Error[1] * v2Error[1]);
      vvErrors.push_back(v2Error);
      double J[2][9];
      const double dDenominator = v3Second[2];
      double dNumerator;
      for (int k = 0; k < 3; k++) { J[0][k] = un[k] / dDenominator; J[0][3 + k] = 0.0; }                              // :160-164
      dNumerator = v3Second[0];
      for (int k = 0; k < 3; k++) J[0][6 + k] = -un[k] * dNumerator / (dDenominator * dDenominator);                   // :165-169
      for (int k = 0; k < 3; k++) { J[1][k] = 0.0; J[1][3 + k] = un[k] / dDenominator; }                              // :171-175
      dNumerator = v3Second[1];
      for (int k = 0; k < 3; k++) J[1][6 + k] = -un[k] * dNumerator / (dDenominator * dDenominator);                   // :176-180
      std::array<double, 18> PJ;                                                                                       // m2PixelProjectionJac * m29Jacobian
      for (int k = 0; k < 9; k++) { PJ[k] = in.jac[0] * J[0][k] + in.jac[1] * J[1][k]; PJ[9 + k] = in.jac[2] * J[0][k] + in.jac[3] * J[1][k]; }
      vmJacobians.push_back(PJ);
    }
    if (mvHomographyInliers.empty()) return;
    std::vector<double> vdd = vdErrorSquared;                                                                         // :186-189
    const double dSigmaSquared = find_sigma_squared(EST_TUKEY, vdd);
    for (size_t i = 0; i < mvHomographyInliers.size(); i++) {                                                         // :192-199
      const double dWeight = weight(EST_TUKEY, vdErrorSquared[i], dSigmaSquared);
      for (int row = 0; row < 2; row++) {
        const double m = (double)(int)vvErrors[i][row];                                                               // add_mJ((int) error, ...): the reference's cast
        const double* J = &vmJacobians[i][row * 9];
        for (int r = 0; r < N; r++) {                                                                                  // myWLS::add_mJ, upper triangle
          const double Jw = dWeight * J[r];
          vec[r] += m * Jw;
          for (int c = r; c < N; c++) C[(size_t)r * N + c] += Jw * J[c];
        }
      }
    }
    for (int r = 1; r < N; r++) for (int c = 0; c < r; c++) C[(size_t)r * N + c] = C[(size_t)c * N + r];            // myWLS::compute
    if (!lu_solve(C.data(), vec.data(), N)) return;
    for (int i = 0; i < 9; i++) mm3BestHomography.m[i] += vec[i];                                                     // :204-216
  }

  void BestHomographyFromMatches_MLESAC() {                                                                           // :224-262
    if (mvMatches.size() < 10) { mm3BestHomography = HomographyFromMatches(mvMatches); return; }
    int anIndices[4];
    mm3BestHomography = mat3_identity();
    double dBestError = 999999999999999999.9;
    for (int nR = 0; nR < 300; nR++) {
      unsigned draw = 0;
      for (int i = 0; i < 4; i++) {
        bool isUnique = false;
        int n = 0;
        while (!isUnique) {
          n = (int)(boot_rand(seed, (unsigned)nR, draw++) % (unsigned)mvMatches.size());                             // rand() % size
          isUnique = true;
          for (int j = 0; j < i && isUnique; j++) if (anIndices[j] == n) isUnique = false;
        }
        anIndices[i] = n;
      }
      std::vector<HMatch> vMinimalMatches;
      for (int i = 0; i < 4; i++) vMinimalMatches.push_back(mvMatches[anIndices[i]]);
      const Mat3 H = HomographyFromMatches(vMinimalMatches);
      double dError = 0.0;
      for (size_t i = 0; i < mvMatches.size(); i++) dError += MLESACScore(H, mvMatches[i]);
      if (dError < dBestError) { mm3BestHomography = H; dBestError = dError; }
    }
  }

  void DecomposeHomography() {                                                                                         // :264-374
    mvDecompositions.clear();
    std::vector<double> M(mm3BestHomography.m, mm3BestHomography.m + 9), Uv, Sv, Vv;
    jacobi_svd_square(3, M, Uv, Sv, Vv);
    const double d1 = fabs(Sv[0]), d2 = fabs(Sv[1]), d3 = fabs(Sv[2]);
    Mat3 U, V;
    for (int i = 0; i < 9; i++) { U.m[i] = Uv[i]; V.m[i] = Vv[i]; }
    const double s = det33(U) * det33(V);
    const double dPrime_PM = d2;
    int nCase;
    if (d1 != d2 && d2 != d3) nCase = 1; else if (d1 == d2 && d2 == d3) nCase = 3; else nCase = 2;
    if (nCase != 1) return;
    const double x1_PM = sqrt((d1 * d1 - d2 * d2) / (d1 * d1 - d3 * d3)), x2 = 0, x3_PM = sqrt((d2 * d2 - d3 * d3) / (d1 * d1 - d3 * d3));   // Eq. 12
    const double e1[4] = {1.0, -1.0, 1.0, -1.0}, e3[4] = {1.0, 1.0, -1.0, -1.0};
    HDecomposition decomposition;
    memset(&decomposition, 0, sizeof(decomposition));
    decomposition.d = s * dPrime_PM;                                                                                   // case 1, d' > 0
    for (int signs = 0; signs < 4; signs++) {
      Mat3 Rp = mat3_identity();                                                                                       // Eq. 13
      const double dSinTheta = (d1 - d3) * x1_PM * x3_PM * e1[signs] * e3[signs] / d2;
      const double dCosTheta = (d1 * x3_PM * x3_PM + d3 * x1_PM * x1_PM) / d2;
      Rp(0, 0) = dCosTheta; Rp(0, 2) = -dSinTheta; Rp(2, 0) = dSinTheta; Rp(2, 2) = dCosTheta;
      memcpy(decomposition.Rp, Rp.m, sizeof(Rp.m));
      decomposition.Tp[0] = (d1 - d3) * x1_PM * e1[signs]; decomposition.Tp[1] = 0.0; decomposition.Tp[2] = (d1 - d3) * -x3_PM * e3[signs];   // Eq. 14
      const double np[3] = {x1_PM * e1[signs], x2, x3_PM * e3[signs]};
      for (int r = 0; r < 3; r++) decomposition.n[r] = V(r, 0) * np[0] + V(r, 1) * np[1] + V(r, 2) * np[2];
      mvDecompositions.push_back(decomposition);
    }
    decomposition.d = s * -dPrime_PM;                                                                                  // case 1, d' < 0
    for (int signs = 0; signs < 4; signs++) {
      Mat3 Rp = mat3_identity();                                                                                       // Eq. 15
      for (int i = 0; i < 9; i++) Rp.m[i] = -1 * Rp.m[i];
      const double dSinPhi = (d1 + d3) * x1_PM * x3_PM * e1[signs] * e3[signs] / d2;
      const double dCosPhi = (d3 * x1_PM * x1_PM - d1 * x3_PM * x3_PM) / d2;
      Rp(0, 0) = dCosPhi; Rp(0, 2) = dSinPhi; Rp(2, 0) = dSinPhi; Rp(2, 2) = -dCosPhi;
      memcpy(decomposition.Rp, Rp.m, sizeof(Rp.m));
      decomposition.Tp[0] = (d1 + d3) * x1_PM * e1[signs]; decomposition.Tp[1] = 0.0; decomposition.Tp[2] = (d1 + d3) * x3_PM * e3[signs];    // Eq. 16
      const double np[3] = {x1_PM * e1[signs], x2, x3_PM * e3[signs]};
      for (int r = 0; r < 3; r++) decomposition.n[r] = V(r, 0) * np[0] + V(r, 1) * np[1] + V(r, 2) * np[2];
      mvDecompositions.push_back(decomposition);
    }
    for (size_t i = 0; i < mvDecompositions.size(); i++) {                                                            // :365-373
      Mat3 Rp; memcpy(Rp.m, mvDecompositions[i].Rp, sizeof(Rp.m));
      Mat3 sU = U; for (int k = 0; k < 9; k++) sU.m[k] = s * sU.m[k];                                                 // s * U * Rp * V^T, evaluated left to right
      const Mat3 rotation = mul33(mul33(sU, Rp), transpose33(V));
      for (int k = 0; k < 9; k++) mvDecompositions[i].R[k] = rotation.m[k];
      for (int r = 0; r < 3; r++) mvDecompositions[i].t[r] = U(r, 0) * mvDecompositions[i].Tp[0] + U(r, 1) * mvDecompositions[i].Tp[1] + U(r, 2) * mvDecompositions[i].Tp[2];
    }
  }

  static double SampsonusError(const double v2Dash[2], const Mat3& E, const double v2[2]) {                            // :380-403
    double v3Dash[3], v3[3], fv3[3], fTv3Dash[3];
    unproject(v2Dash, v3Dash); unproject(v2, v3);
    for (int r = 0; r < 3; r++) fv3[r] = E(r, 0) * v3[0] + E(r, 1) * v3[1] + E(r, 2) * v3[2];
    const double dError = fv3[0] * v3Dash[0] + fv3[1] * v3Dash[1] + fv3[2] * v3Dash[2];
    for (int r = 0; r < 3; r++) fTv3Dash[r] = E(0, r) * v3Dash[0] + E(1, r) * v3Dash[1] + E(2, r) * v3Dash[2];
    return dError * dError / ((fv3[0] * fv3[0] + fv3[1] * fv3[1]) + (fTv3Dash[0] * fTv3Dash[0] + fTv3Dash[1] * fTv3Dash[1]));
  }

  void ChooseBestDecomposition() {                                                                                     // :405-499
    for (size_t i = 0; i < mvDecompositions.size(); i++) {
      HDecomposition& decom = mvDecompositions[i];
      int nPositive = 0;
      for (size_t m = 0; m < mvHomographyInliers.size(); m++) {
        const double* v2 = mvHomographyInliers[m].first;
        const double dVisibilityTest = (mm3BestHomography(2, 0) * v2[0] + mm3BestHomography(2, 1) * v2[1] + mm3BestHomography(2, 2)) / decom.d;
        if (dVisibilityTest > 0.0) nPositive++;
      }
      decom.score = -nPositive;
    }
    // sort(begin, end) with operator< on nScore: not stable in the reference; equal scores keep their order of generation here
    std::stable_sort(mvDecompositions.begin(), mvDecompositions.end(), [](const HDecomposition& a, const HDecomposition& b) { return a.score < b.score; });
    mvDecompositions.resize(4);
    for (size_t i = 0; i < mvDecompositions.size(); i++) {
      HDecomposition& decom = mvDecompositions[i];
      int nPositive = 0;
      for (size_t m = 0; m < mvHomographyInliers.size(); m++) {
        double v3[3];
        unproject(mvHomographyInliers[m].first, v3);
        const double dVisibilityTest = (v3[0] * decom.n[0] + v3[1] * decom.n[1] + v3[2] * decom.n[2]) / decom.d;
        if (dVisibilityTest > 0.0) nPositive++;
      }
      decom.score = -nPositive;
    }
    std::stable_sort(mvDecompositions.begin(), mvDecompositions.end(), [](const HDecomposition& a, const HDecomposition& b) { return a.score < b.score; });
    mvDecompositions.resize(2);
    const double dRatio = (double)mvDecompositions[1].score / (double)mvDecompositions[0].score;
    if (dRatio < 0.9) mvDecompositions.erase(mvDecompositions.begin() + 1);                                           // no ambiguity
    else {                                                                                                             // two-way ambiguity: Sampson score of all matches
      const double dErrorSquaredLimit = mdMaxPixelErrorSquared * 4;
      double adSampsonusScores[2];
      for (int i = 0; i < 2; i++) {
        const HDecomposition& D = mvDecompositions[i];
        Mat3 m3Essential;
        for (int j = 0; j < 3; j++) {                                                                                  // column j = translation x column j of the rotation
          const double rot_T[3] = {D.R[0 * 3 + j], D.R[1 * 3 + j], D.R[2 * 3 + j]};
          m3Essential(0, j) = D.t[1] * rot_T[2] - D.t[2] * rot_T[1];
          m3Essential(1, j) = D.t[2] * rot_T[0] - D.t[0] * rot_T[2];
          m3Essential(2, j) = D.t[0] * rot_T[1] - D.t[1] * rot_T[0];
        }
        double dSumError = 0;
        for (size_t m = 0; m < mvMatches.size(); m++) {
          double d = SampsonusError(mvMatches[m].second, m3Essential, mvMatches[m].first);
          if (d > dErrorSquaredLimit) d = dErrorSquaredLimit;
          dSumError += d;
        }
        adSampsonusScores[i] = dSumError;
      }
      if (adSampsonusScores[0] <= adSampsonusScores[1]) mvDecompositions.erase(mvDecompositions.begin() + 1);
      else mvDecompositions.erase(mvDecompositions.begin());
    }
  }

  bool Compute(const std::vector<HMatch>& vMatches, double dMaxPixelError, SE3& se3SecondCameraPose) {              // :43-71
    mdMaxPixelErrorSquared = dMaxPixelError * dMaxPixelError;
    mvMatches = vMatches;
    if (mvMatches.size() < 4) return false;                                                                           // (the reference asserts in HomographyFromMatches)
    BestHomographyFromMatches_MLESAC();
    mvHomographyInliers.clear();
    for (size_t i = 0; i < mvMatches.size(); i++) if (IsHomographyInlier(mm3BestHomography, mvMatches[i])) mvHomographyInliers.push_back(mvMatches[i]);
    for (int iteration = 0; iteration < 5; iteration++) RefineHomographyWithInliers();
    DecomposeHomography();
    if (mvDecompositions.size() != 8) return false;
    ChooseBestDecomposition();
    for (int i = 0; i < 9; i++) se3SecondCameraPose.R[i] = mvDecompositions[0].R[i];
    for (int i = 0; i < 3; i++) se3SecondCameraPose.t[i] = mvDecompositions[0].t[i];
    return true;
  }
};

}  // namespace hinit

bool homography_init_compute(const std::vector<HMatch>& m, double max_pixel_error, unsigned seed, SE3& second_from_first, int* n_inliers) {
  hinit::Init h;
  h.seed = seed;
  const bool ok = h.Compute(m, max_pixel_error, second_from_first);
  if (n_inliers) *n_inliers = (int)h.mvHomographyInliers.size();
  return ok;
}

// The unit eigenvector of the smallest eigenvalue of a symmetric 3x3 matrix: the eigenvalues in closed form (the trigonometric
// solution of the characteristic cubic), the vector as the largest cross product of two rows of (A - lambda I).
static void sym3_least_eigenvector(const double A[9], double out[3]) {
  const double p1 = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
  const double q = (A[0] + A[4] + A[8]) / 3.0;
  double lambda;
  if (p1 == 0.0) lambda = std::min(A[0], std::min(A[4], A[8]));
  else {
    const double p2 = (A[0] - q) * (A[0] - q) + (A[4] - q) * (A[4] - q) + (A[8] - q) * (A[8] - q) + 2.0 * p1;
    const double p = sqrt(p2 / 6.0);
    double B[9];
    for (int i = 0; i < 9; i++) B[i] = (A[i] - (i % 4 == 0 ? q : 0.0)) / p;
    double r = (B[0] * (B[4] * B[8] - B[5] * B[7]) - B[1] * (B[3] * B[8] - B[5] * B[6]) + B[2] * (B[3] * B[7] - B[4] * B[6])) / 2.0;
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double phi = acos(r) / 3.0;
    lambda = q + 2.0 * p * cos(phi + 2.0 * M_PI / 3.0);                    // the smallest of the three
  }
  const double r0[3] = {A[0] - lambda, A[1], A[2]}, r1[3] = {A[3], A[4] - lambda, A[5]}, r2[3] = {A[6], A[7], A[8] - lambda};
  const double c[3][3] = {{r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]},
                          {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]},
                          {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]}};
  int best = 0; double bn = -1.0;
  for (int i = 0; i < 3; i++) { const double n = c[i][0] * c[i][0] + c[i][1] * c[i][1] + c[i][2] * c[i][2]; if (n > bn) { bn = n; best = i; } }
  if (bn <= 0.0) { out[0] = 0; out[1] = 0; out[2] = 1; return; }
  const double inv = 1.0 / sqrt(bn);
  for (int k = 0; k < 3; k++) out[k] = c[best][k] * inv;
}

// MapMaker::CalcPlaneAligner, jni/MapMaker.cc:1104-1231, over the world positions of all map points
bool calc_plane_aligner(const std::vector<V3>& vpPoints, unsigned seed, SE3& se3Aligner) {
  se3Aligner = SE3();
  const unsigned nPoints = (unsigned)vpPoints.size();
  if (nPoints < 10) return false;                                                                                      // :1107-1110
  const int nRansacs = 100;
  V3 v3BestMean = v3(0, 0, 0), v3BestNormal = v3(0, 0, 1);
  double dBestDistSquared = 9999999999999999.9;
  for (int i = 0; i < nRansacs; i++) {
    unsigned draw = 0;
    const int nA = (int)(hinit::boot_rand(seed, (unsigned)i, draw++) % nPoints);                                      // rand() % nPoints
    int nB = nA, nC = nA;
    while (nB == nA) nB = (int)(hinit::boot_rand(seed, (unsigned)i, draw++) % nPoints);
    while (nC == nA || nC == nB) nC = (int)(hinit::boot_rand(seed, (unsigned)i, draw++) % nPoints);
    const V3& a = vpPoints[nA]; const V3& b = vpPoints[nB]; const V3& c = vpPoints[nC];
    const V3 v3Mean = v3(0.33333333 * (a[0] + b[0] + c[0]), 0.33333333 * (a[1] + b[1] + c[1]), 0.33333333 * (a[2] + b[2] + c[2]));
    const V3 v3CA = c - a, v3BA = b - a;
    V3 v3Normal = cross(v3CA, v3BA);
    if (dot(v3Normal, v3Normal) == 0) continue;
    { const double n = sqrt(dot(v3Normal, v3Normal)); v3Normal = v3(v3Normal[0] / n, v3Normal[1] / n, v3Normal[2] / n); }   // normalize()
    double dSumError = 0.0;
    for (unsigned k = 0; k < nPoints; k++) {
      const V3 v3Diff = vpPoints[k] - v3Mean;
      const double dDistSq = dot(v3Diff, v3Diff);
      if (dDistSq == 0.0) continue;
      double dNormDist = fabs(dot(v3Diff, v3Normal));
      if (dNormDist > 0.05) dNormDist = 0.05;
      dSumError += dNormDist;
    }
    if (dSumError < dBestDistSquared) { dBestDistSquared = dSumError; v3BestMean = v3Mean; v3BestNormal = v3Normal; }
  }
  std::vector<V3> vv3Inliers;                                                                                          // :1165-1178
  for (unsigned i = 0; i < nPoints; i++) {
    const V3 v3Diff = vpPoints[i] - v3BestMean;
    if (dot(v3Diff, v3Diff) == 0.0) continue;
    if (fabs(dot(v3Diff, v3BestNormal)) < 0.05) vv3Inliers.push_back(vpPoints[i]);
  }
  if (vv3Inliers.empty()) return false;                                                                               // (the reference would divide by zero)
  V3 v3MeanOfInliers = v3(0, 0, 0);
  for (auto& p : vv3Inliers) v3MeanOfInliers = v3MeanOfInliers + p;
  { const double f = 1.0 / vv3Inliers.size(); v3MeanOfInliers = v3(v3MeanOfInliers[0] * f, v3MeanOfInliers[1] * f, v3MeanOfInliers[2] * f); }
  double m3Cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (auto& p : vv3Inliers) { const V3 d = p - v3MeanOfInliers; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m3Cov[r * 3 + c] += d[r] * d[c]; }
  double nrm[3];
  sym3_least_eigenvector(m3Cov, nrm);                                                                                  // :1196-1207 (see the header)
  V3 v3Normal = v3(nrm[0], nrm[1], nrm[2]);
  if (v3Normal[2] > 0) v3Normal = v3(-v3Normal[0], -v3Normal[1], -v3Normal[2]);                                       // :1210-1211
  const double nx = v3Normal[0], ny = v3Normal[1], nz = v3Normal[2];
  const double dd = 1.0 * nx + 0.0 * ny + 0.0 * nz;                            // fila0 . normal, fila0 = row 0 of the identity
  double ax = 1.0 - nx * dd, ay = 0.0 - ny * dd, az = 0.0 - nz * dd;           // aux = fila0 - normal * (fila0 . normal)
  const double an = sqrt(ax * ax + ay * ay + az * az);
  ax = ax / an; ay = ay / an; az = az / an;                                    // aux.normalize()
  const double bx = ny * az - nz * ay, by = nz * ax - nx * az, bz = nx * ay - ny * ax;   // fila2.cross(fila0)
  se3Aligner.R[0] = ax; se3Aligner.R[1] = ay; se3Aligner.R[2] = az;
  se3Aligner.R[3] = bx; se3Aligner.R[4] = by; se3Aligner.R[5] = bz;
  se3Aligner.R[6] = nx; se3Aligner.R[7] = ny; se3Aligner.R[8] = nz;
  se3Aligner.t[0] = se3Aligner.t[1] = se3Aligner.t[2] = 0.0;
  const V3 v3RMean = xform(se3Aligner, v3MeanOfInliers);
  for (int k = 0; k < 3; k++) se3Aligner.t[k] = -v3RMean[k];
  return true;
}

}  // namespace orc
