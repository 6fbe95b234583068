// ORACLE (test infrastructure only).  jni/Bundle.cc restated (Tukey M-estimator, jni/Bundle.cc:152).
// The std::list<Meas> of the reference is a vector with an `erased` flag; iteration order is unchanged.
#include "ptam_system.hpp"

namespace orc {

int Bundle::AddCamera(const SE3& pose, bool fixed) {
  // jni/Bundle.cc:71-88
  BCamera c; c.fixed = fixed; c.pose = pose; c.pose_new = pose;
  memset(c.U, 0, sizeof(c.U)); memset(c.ea, 0, sizeof(c.ea));
  if (!fixed) { c.start_row = next_start_row; next_start_row += 6; n_cams_to_update++; }
  else c.start_row = -999999999;
  cams.push_back(c);
  return (int)cams.size() - 1;
}

int Bundle::AddPoint(V3 pos) {
  // jni/Bundle.cc:91-103
  BPoint p;
  if (std::isnan(dot(pos, pos))) pos = v3(0, 0, 0);
  p.pos = pos; p.pos_new = pos;
  memset(p.V, 0, sizeof(p.V)); memset(p.eb, 0, sizeof(p.eb)); memset(p.Vinv, 0, sizeof(p.Vinv));
  pts.push_back(p);
  return (int)pts.size() - 1;
}

void Bundle::AddMeas(int nCam, int nPoint, const double pos[2], double dSigmaSquared) {
  // jni/Bundle.cc:106-117
  pts[nPoint].n_meas++;
  pts[nPoint].cams.insert(nCam);
  BMeas m; m.p = nPoint; m.c = nCam; m.found[0] = pos[0]; m.found[1] = pos[1];
  m.sqrt_inv_noise = sqrt(1.0 / dSigmaSquared);
  memset(m.A, 0, sizeof(m.A)); memset(m.B, 0, sizeof(m.B)); memset(m.W, 0, sizeof(m.W));
  m.eps[0] = m.eps[1] = 0; m.err2 = 0; memset(m.derivs, 0, sizeof(m.derivs)); m.cam = v3(0, 0, 0);
  meas.push_back(m);
}

int Bundle::Compute(const bool* abort) {
  // jni/Bundle.cc:136-178
  lut.assign(cams.size(), std::vector<int>(pts.size(), -1));   // GenerateMeasLUTs :566-575
  for (size_t i = 0; i < meas.size(); i++) if (!meas[i].erased) lut[meas[i].c][meas[i].p] = (int)i;
  for (size_t i = 0; i < pts.size(); i++) {                      // GenerateOffDiagScripts :580-607
    BPoint& p = pts[i];
    p.script.clear();
    for (auto itj = p.cams.begin(); itj != p.cams.end(); ++itj) {
      const int j = *itj;
      if (cams[j].fixed) continue;
      for (auto itk = p.cams.begin(); itk != itj; ++itk) {
        const int k = *itk;
        if (cams[k].fixed) continue;
        p.script.push_back(std::make_pair(j, k));
      }
    }
  }
  lambda = 0.0001; lambda_factor = 2.0;
  converged = false; hit_max = false; counter = 0; accepted = 0;
  while (!converged && !hit_max && !*abort) {
    if (!Do_LM_Step(abort)) return -1;
  }
  return accepted;
}

void Bundle::ProjectAndFindSquaredError(BMeas& m) {
  // jni/Bundle.cc:181-199
  const BCamera& cam = cams[m.c];
  const BPoint& pt = pts[m.p];
  m.cam = xform(cam.pose, pt.pos);
  if (m.cam[2] <= 0) { m.bad = true; return; }
  m.bad = false;
  const Camera::Proj pr = camera.project(m.cam[0] / m.cam[2], m.cam[1] / m.cam[2]);
  camera.derivs(pr, m.derivs);
  m.eps[0] = (m.found[0] - pr.im[0]) * m.sqrt_inv_noise;
  m.eps[1] = (m.found[1] - pr.im[1]) * m.sqrt_inv_noise;
  m.err2 = m.eps[0] * m.eps[0] + m.eps[1] * m.eps[1];
}

bool Bundle::Do_LM_Step(const bool* abort) {
  // jni/Bundle.cc:202-532
  for (auto& p : pts) { memset(p.V, 0, sizeof(p.V)); memset(p.eb, 0, sizeof(p.eb)); }     // ClearAccumulators :120-129
  for (auto& c : cams) { memset(c.U, 0, sizeof(c.U)); memset(c.ea, 0, sizeof(c.ea)); }
  std::vector<double> e2;
  for (auto& m : meas) {                                                                   // :209-215
    if (m.erased) continue;
    ProjectAndFindSquaredError(m);
    if (!m.bad) e2.push_back(m.err2);
  }
  if (e2.empty()) return false;   // (reference asserts; a problem with no valid measurement is an error)
  sigma2 = find_sigma_squared(EST_TUKEY, e2);                                              // :220
  const double dMinSigmaSquared = min_sigma * min_sigma;                                   // :224-227
  if (sigma2 < dMinSigmaSquared) sigma2 = dMinSigmaSquared;

  double dCurrentError = 0.0;
  for (auto& m : meas) {                                                                   // :241-321
    if (m.erased) continue;
    BCamera& cam = cams[m.c];
    BPoint& point = pts[m.p];
    if (m.bad) { dCurrentError += 1.0; continue; }
    const double dWeight = sqrt_weight(EST_TUKEY, m.err2, sigma2);
    m.eps[0] *= dWeight; m.eps[1] *= dWeight;
    if (dWeight == 0) { m.bad = true; dCurrentError += 1.0; continue; }
    dCurrentError += objective(EST_TUKEY, m.err2, sigma2);
    const double d[4] = {dWeight * m.derivs[0], dWeight * m.derivs[1], dWeight * m.derivs[2], dWeight * m.derivs[3]};
    const double ooz = 1.0 / m.cam[2];
    const double v4[4] = {m.cam[0], m.cam[1], m.cam[2], 1.0};
    if (cam.fixed) memset(m.A, 0, sizeof(m.A));
    else
      for (int k = 0; k < 6; k++) {
        double mot[4];
        generator_field(k, v4, mot);
        const double f0 = (mot[0] - v4[0] * mot[2] * ooz) * ooz, f1 = (mot[1] - v4[1] * mot[2] * ooz) * ooz;
        // meas.dSqrtInvNoise * m2CamDerivs * v2CamFrameMotion : (s*D) * v
        m.A[0 * 6 + k] = (m.sqrt_inv_noise * d[0]) * f0 + (m.sqrt_inv_noise * d[1]) * f1;
        m.A[1 * 6 + k] = (m.sqrt_inv_noise * d[2]) * f0 + (m.sqrt_inv_noise * d[3]) * f1;
      }
    for (int k = 0; k < 3; k++) {
      const double mot[3] = {cam.pose.R[0 * 3 + k], cam.pose.R[1 * 3 + k], cam.pose.R[2 * 3 + k]};
      const double f0 = (mot[0] - v4[0] * mot[2] * ooz) * ooz, f1 = (mot[1] - v4[1] * mot[2] * ooz) * ooz;
      m.B[0 * 3 + k] = (m.sqrt_inv_noise * d[0]) * f0 + (m.sqrt_inv_noise * d[1]) * f1;
      m.B[1 * 3 + k] = (m.sqrt_inv_noise * d[2]) * f0 + (m.sqrt_inv_noise * d[3]) * f1;
    }
    if (!cam.fixed) {
      for (int r = 0; r < 6; r++) for (int c = 0; c <= r; c++) cam.U[r * 6 + c] += m.A[r] * m.A[c] + m.A[6 + r] * m.A[6 + c];   // :40-47
      for (int r = 0; r < 6; r++) cam.ea[r] += m.A[r] * m.eps[0] + m.A[6 + r] * m.eps[1];
    }
    for (int r = 0; r < 3; r++) for (int c = 0; c <= r; c++) point.V[r * 3 + c] += m.B[r] * m.B[c] + m.B[3 + r] * m.B[3 + c];    // :49-56
    for (int r = 0; r < 3; r++) point.eb[r] += m.B[r] * m.eps[0] + m.B[3 + r] * m.eps[1];
    if (cam.fixed) memset(m.W, 0, sizeof(m.W));
    else for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) m.W[r * 3 + c] = m.A[r] * m.B[c] + m.A[6 + r] * m.B[3 + c];
  }

  const int nS = n_cams_to_update * 6;
  double dNewError = dCurrentError + 9999;
  std::vector<double> S, E, camUp, mapUp(pts.size() * 3);
  while (dNewError > dCurrentError && !converged && !hit_max && !*abort) {                 // :327-501
    for (auto& point : pts) {                                                              // :329-347
      double Vs[9];
      memcpy(Vs, point.V, sizeof(Vs));
      if (Vs[0] * Vs[4] * Vs[8] == 0) { memset(point.Vinv, 0, sizeof(point.Vinv)); continue; }
      Vs[1] = Vs[3]; Vs[2] = Vs[6]; Vs[5] = Vs[7];
      for (int i = 0; i < 3; i++) Vs[i * 3 + i] *= (1.0 + lambda);
      inv3(Vs, point.Vinv);
    }
    S.assign((size_t)nS * nS, 0.0); E.assign(nS, 0.0);
    for (size_t j = 0; j < cams.size(); j++) {                                             // :362-396
      BCamera& cj = cams[j];
      if (cj.fixed) continue;
      const int row = cj.start_row;
      double m6[36], v6[6];
      for (int r = 0; r < 6; r++) { for (int c = 0; c < r; c++) m6[r * 6 + c] = m6[c * 6 + r] = cj.U[r * 6 + c]; m6[r * 6 + r] = cj.U[r * 6 + r]; }
      for (int n = 0; n < 6; n++) m6[n * 6 + n] *= (1.0 + lambda);
      for (int r = 0; r < 6; r++) v6[r] = cj.ea[r];
      for (size_t i = 0; i < pts.size(); i++) {
        const int mi = lut[j][i];
        if (mi < 0 || meas[mi].bad) continue;
        const BMeas& m = meas[mi];
        const double* Vi = pts[i].Vinv;
        double Y[18];   // W * V*inv  (6x3)
        for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) Y[r * 3 + c] = m.W[r * 3 + 0] * Vi[0 * 3 + c] + m.W[r * 3 + 1] * Vi[1 * 3 + c] + m.W[r * 3 + 2] * Vi[2 * 3 + c];
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) m6[r * 6 + c] -= Y[r * 3 + 0] * m.W[c * 3 + 0] + Y[r * 3 + 1] * m.W[c * 3 + 1] + Y[r * 3 + 2] * m.W[c * 3 + 2];
        double ve[3];   // V*inv * eb
        for (int r = 0; r < 3; r++) ve[r] = Vi[r * 3 + 0] * pts[i].eb[0] + Vi[r * 3 + 1] * pts[i].eb[1] + Vi[r * 3 + 2] * pts[i].eb[2];
        for (int r = 0; r < 6; r++) v6[r] -= m.W[r * 3 + 0] * ve[0] + m.W[r * 3 + 1] * ve[1] + m.W[r * 3 + 2] * ve[2];
      }
      for (int r = 0; r < 6; r++) { for (int c = 0; c < 6; c++) S[(size_t)(row + r) * nS + row + c] = m6[r * 6 + c]; E[row + r] = v6[r]; }
    }
    for (size_t i = 0; i < pts.size(); i++) {                                              // :400-426
      BPoint& p = pts[i];
      int curJ = -1, jrow = -1;
      double Y[18];
      for (auto& e : p.script) {
        const int mik = lut[e.second][i];
        if (mik < 0 || meas[mik].bad) continue;
        if (e.first != curJ) {
          const int mij = lut[e.first][i];
          if (mij < 0 || meas[mij].bad) continue;
          curJ = e.first; jrow = cams[e.first].start_row;
          const double* Wj = meas[mij].W;
          for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) Y[r * 3 + c] = Wj[r * 3 + 0] * p.Vinv[0 * 3 + c] + Wj[r * 3 + 1] * p.Vinv[1 * 3 + c] + Wj[r * 3 + 2] * p.Vinv[2 * 3 + c];
        }
        const int krow = cams[meas[mik].c].start_row;
        const double* Wk = meas[mik].W;
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++)
          S[(size_t)(jrow + r) * nS + krow + c] -= Y[r * 3 + 0] * Wk[c * 3 + 0] + Y[r * 3 + 1] * Wk[c * 3 + 1] + Y[r * 3 + 2] * Wk[c * 3 + 2];
      }
    }
    for (int i = 0; i < nS; i++) for (int j = 0; j < i; j++) S[(size_t)j * nS + i] = S[(size_t)i * nS + j];   // :431-434
    camUp = E;                                                                             // :437 mS.inverse()*vE
    {
      std::vector<double> A = S;
      if (nS > 0 && !lu_solve(A.data(), camUp.data(), nS)) return false;
    }
    for (size_t i = 0; i < pts.size(); i++) {                                              // :440-462
      double sum[3] = {0, 0, 0};
      for (size_t j = 0; j < cams.size(); j++) {
        const BCamera& cam = cams[j];
        if (cam.fixed) continue;
        const int mi = lut[j][i];
        if (mi < 0 || meas[mi].bad) continue;
        const double* W = meas[mi].W;
        for (int c = 0; c < 3; c++) { double s = 0; for (int r = 0; r < 6; r++) s += W[r * 3 + c] * camUp[cam.start_row + r]; sum[c] += s; }
      }
      const double v[3] = {pts[i].eb[0] - sum[0], pts[i].eb[1] - sum[1], pts[i].eb[2] - sum[2]};
      for (int r = 0; r < 3; r++) mapUp[i * 3 + r] = pts[i].Vinv[r * 3 + 0] * v[0] + pts[i].Vinv[r * 3 + 1] * v[1] + pts[i].Vinv[r * 3 + 2] * v[2];
    }
    double dSumSquaredUpdate = 0;                                                          // :467-470
    for (double x : camUp) dSumSquaredUpdate += x * x;
    for (double x : mapUp) dSumSquaredUpdate += x * x;
    if (dSumSquaredUpdate < convergence_limit) converged = true;
    for (auto& c : cams) {                                                                 // :476-485
      if (c.fixed) c.pose_new = c.pose;
      else c.pose_new = mul(se3_exp(&camUp[c.start_row]), c.pose);
    }
    for (size_t i = 0; i < pts.size(); i++) pts[i].pos_new = pts[i].pos + v3(mapUp[i * 3], mapUp[i * 3 + 1], mapUp[i * 3 + 2]);
    dNewError = FindNewError();
    if (dNewError > dCurrentError) { lambda = lambda * lambda_factor; lambda_factor = lambda_factor * 2; }   // :609-617
    counter++; n_trials++;
    if (counter >= max_iterations) hit_max = true;
  }
  if (dNewError < dCurrentError) {                                                         // :503-514
    lambda_factor = 2.0; lambda *= 0.3;
    for (auto& c : cams) c.pose = c.pose_new;
    for (auto& p : pts) p.pos = p.pos_new;
    accepted++;
  }
  for (auto& m : meas) {                                                                   // :517-528
    if (m.erased || !m.bad) continue;
    m.erased = true;
    outlier_meas.push_back(std::make_pair(m.p, m.c));
    pts[m.p].n_outliers++;
    lut[m.c][m.p] = -1;
  }
  return true;
}

double Bundle::FindNewError() {
  // jni/Bundle.cc:537-561
  double dNewError = 0;
  for (auto& m : meas) {
    if (m.erased) continue;
    const V3 c = xform(cams[m.c].pose_new, pts[m.p].pos_new);
    if (c[2] <= 0) { dNewError += 1.0; continue; }
    const Camera::Proj pr = camera.project(c[0] / c[2], c[1] / c[2]);
    const double e0 = (m.found[0] - pr.im[0]) * m.sqrt_inv_noise, e1 = (m.found[1] - pr.im[1]) * m.sqrt_inv_noise;
    dNewError += objective(EST_TUKEY, e0 * e0 + e1 * e1, sigma2);
  }
  return dNewError;
}

std::set<int> Bundle::GetOutliers() const {
  // jni/Bundle.cc:628-637
  std::set<int> s;
  for (size_t i = 0; i < pts.size(); i++) if (pts[i].n_meas > 0 && pts[i].n_meas == pts[i].n_outliers) s.insert((int)i);
  return s;
}

}  // namespace orc
