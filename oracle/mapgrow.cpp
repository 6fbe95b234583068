// ORACLE (test infrastructure only).  Map growth on a new keyframe (first part):
//   KeyFrame::MakeKeyFrame_Rest candidates        jni/KeyFrame.cc:53-95   (oracle/frontend.cpp orc_candidates)
//   MapMaker::ThinCandidates                      jni/MapMaker.cc:393-422
//   MapMaker::AddSomeMapPoints / AddPointEpipolar jni/MapMaker.cc:424-437, 525-703
//   MapMaker::ReprojectPoint                      jni/MapMaker.cc:174-200
//   MapPoint::RefreshPixelVectors                 jni/MapPoint.cc:4-29
//   PatchFinder::MakeTemplateCoarseNoWarp         jni/PatchFinder.cc:130-142
//   MapMaker::ReFind_Common / ReFindInSingleKeyFrame  jni/MapMaker.cc:967-1056  (grow_map bit 1)
//   MapMaker::ReFindNewlyMade / ReFindFromFailureQueue / the idle jobs of run()   jni/MapMaker.cc:94-117, 1061-1100 (below)
// Third-party arithmetic restated (parity unpinned): Eigen::JacobiSVD of the 4x4 triangulation matrix (two-sided Jacobi on A
// itself, below; the sign of the vector cancels in the projective division).
#include "ptam_system.hpp"

namespace orc {

// Eigen::JacobiSVD of a square real matrix (the 4x4 A of ReprojectPoint, jni/MapMaker.cc:191-192), restated from the
// published algorithm of Eigen 3.0-3.1 (the reference's Eigen is an un-vendored dependency, version unpinned: PARITY
// UNPINNED): two-sided Jacobi.  Sweeps over the pairs (p, q), q < p; a pair is rotated while
// max(|m_pq|, |m_qp|) > 2 eps max(|m_pp|, |m_qq|); the 2x2 step (real_2x2_jacobi_svd) first symmetrises the block with a
// left rotation rot1 (t = m_pp + m_qq, d = m_qp - m_pq, u = d / t), then diagonalises it with the Jacobi rotation
// j_right of the symmetric block (JacobiRotation::makeJacobi: tau = (x - z) / (2 |y|), t = 1 / (tau +- sqrt(tau^2 + 1)));
// j_left = rot1 * j_right^T; m <- j_left m j_right, V <- V j_right.  The singular values are the |m_ii|; JacobiSVD sorts
// them in decreasing order, so matrixV().col(3) is the column of V of the smallest one.  Eigen iterates until no pair
// needs a rotation; the cap of 64 sweeps is never reached (4x4 matrices converge in 3-5 sweeps).
void svd4_smallest_right_vector(const double Ain[16], double out[4]) {
  double M[16], V[16];
  for (int i = 0; i < 16; i++) { M[i] = Ain[i]; V[i] = (i % 5 == 0) ? 1.0 : 0.0; }
  const double precision = 2.0 * 2.220446049250313e-16;
  for (int sweep = 0; sweep < 64; sweep++) {
    bool finished = true;
    for (int p = 1; p < 4; p++)
      for (int q = 0; q < p; q++) {
        const double apq = fabs(M[p * 4 + q]), aqp = fabs(M[q * 4 + p]), off = apq > aqp ? apq : aqp;
        const double app = fabs(M[p * 4 + p]), aqq = fabs(M[q * 4 + q]), dia = app > aqq ? app : aqq;
        if (!(off > dia * precision)) continue;
        finished = false;
        const double m00 = M[p * 4 + p], m01 = M[p * 4 + q], m10 = M[q * 4 + p], m11 = M[q * 4 + q];
        double c1, s1;                                        // rot1
        const double t = m00 + m11, d = m10 - m01;
        if (t == 0.0) { c1 = 0.0; s1 = d > 0.0 ? 1.0 : -1.0; }
        else { const double u = d / t; c1 = 1.0 / sqrt(1.0 + u * u); s1 = c1 * u; }
        const double x = c1 * m00 + s1 * m10, y = c1 * m01 + s1 * m11, z = -s1 * m01 + c1 * m11;   // rot1 applied on the left: symmetric [x y; y z]
        double c2, s2;                                        // j_right = makeJacobi(x, y, z)
        if (y == 0.0) { c2 = 1.0; s2 = 0.0; }
        else {
          const double tau = (x - z) / (2.0 * fabs(y)), w = sqrt(tau * tau + 1.0);
          const double tt = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
          const double sign_t = tt > 0.0 ? 1.0 : -1.0, n = 1.0 / sqrt(tt * tt + 1.0);
          s2 = -sign_t * (y / fabs(y)) * fabs(tt) * n; c2 = n;
        }
        const double cl = c1 * c2 + s1 * s2, sl = s1 * c2 - c1 * s2;   // j_left = rot1 * j_right^T
        for (int k = 0; k < 4; k++) {                         // m.applyOnTheLeft(p, q, j_left)
          const double a = M[p * 4 + k], b = M[q * 4 + k];
          M[p * 4 + k] = cl * a + sl * b; M[q * 4 + k] = -sl * a + cl * b;
        }
        for (int k = 0; k < 4; k++) {                         // m.applyOnTheRight(p, q, j_right)
          const double a = M[k * 4 + p], b = M[k * 4 + q];
          M[k * 4 + p] = c2 * a - s2 * b; M[k * 4 + q] = s2 * a + c2 * b;
        }
        for (int k = 0; k < 4; k++) {                         // V.applyOnTheRight(p, q, j_right)
          const double a = V[k * 4 + p], b = V[k * 4 + q];
          V[k * 4 + p] = c2 * a - s2 * b; V[k * 4 + q] = s2 * a + c2 * b;
        }
      }
    if (finished) break;
  }
  int best = 0;
  for (int i = 1; i < 4; i++) if (fabs(M[i * 4 + i]) < fabs(M[best * 4 + best])) best = i;
  for (int k = 0; k < 4; k++) out[k] = V[k * 4 + best];
}

// MapMaker::ReprojectPoint, jni/MapMaker.cc:174-200
V3 reproject_point(const SE3& AfromB, const double v2A[2], const double v2B[2]) {
  double PD[12];                                                // 3x4 [R | t]
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) PD[r * 4 + c] = AfromB.R[r * 3 + c]; PD[r * 4 + 3] = AfromB.t[r]; }
  double A[16] = {-1.0, 0.0, v2B[0], 0.0, 0.0, -1.0, v2B[1], 0.0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = 0; c < 4; c++) { A[8 + c] = v2A[0] * PD[8 + c] - PD[0 + c]; A[12 + c] = v2A[1] * PD[8 + c] - PD[4 + c]; }
  double v[4];
  svd4_smallest_right_vector(A, v);                            // svd.matrixV().block(0, 3, 4, 1)
  if (v[3] == 0.0) v[3] = 0.00001;
  return v3(v[0] / v[3], v[1] / v[3], v[2] / v[3]);
}

// ATANCamera::OnePixelDist, jni/ATANCamera.cc:86-91
static double one_pixel_dist(const Camera& cam) {
  double a[2], b[2];
  cam.unproject(cam.size[0] / 2, cam.size[1] / 2, a);
  cam.unproject(cam.size[0] / 2 + 1, cam.size[1] / 2 + 1, b);
  const double d0 = a[0] - b[0], d1 = a[1] - b[1];
  return sqrt(d0 * d0 + d1 * d1) / sqrt(2.0);
}

void make_keyframe_rest_candidates(KeyFrame& k, double min_score) {
  // jni/KeyFrame.cc:66-95 on the maximal corners make_keyframe_rest_nonmax left
  for (int l = 0; l < 4; l++) {
    const int n = (int)k.maxcorners[l].size();
    k.cand[l].assign(n > 0 ? n : 1, 0); k.cand_score[l].assign(n > 0 ? n : 1, 0.0);
    const int m = orc_candidates(k.im[l].data(), k.w[l], k.h[l], k.w[l], k.maxcorners[l].data(), n, min_score, 10, k.cand[l].data(), k.cand_score[l].data(), n > 0 ? n : 1);
    k.cand[l].resize(m); k.cand_score[l].resize(m);
  }
}

void System::ThinCandidates(KeyFrame& k, int level) {
  // :393-422
  std::vector<double> root; std::vector<int> lev;
  for (auto& it : k.meas) { root.push_back(it.second.root[0]); root.push_back(it.second.root[1]); lev.push_back(it.second.level); }
  const int n = (int)k.cand[level].size();
  std::vector<uint32_t> op(n > 0 ? n : 1); std::vector<double> os(n > 0 ? n : 1);
  const int m = orc_thin_candidates(k.cand[level].data(), k.cand_score[level].data(), n, level, root.data(), lev.data(), (int)lev.size(), op.data(), os.data());
  op.resize(m); os.resize(m);
  k.cand[level] = op; k.cand_score[level] = os;
}

int System::ClosestKeyFrame(int kidx) {
  // :737-758
  double best = 9999999999.9; int n = -1;
  for (int i = 0; i < (int)kfs.size(); i++) {
    if (i == kidx) continue;
    const double d = KeyFrameLinearDist(kfs[kidx]->pose, kfs[i]->pose);
    if (d < best) { best = d; n = i; }
  }
  return n;
}

// MapPoint::RefreshPixelVectors, jni/MapPoint.cc:4-29, with v3Normal_NC = (0, 0, -1)
void refresh_pixel_vectors(MapPoint& p, const KeyFrame& k, const V3& center, const V3& one_right, const V3& one_down) {
  const V3 pc = xform(k.pose, p.pos);
  const double dCamHeight = fabs(-pc[2]);
  const double dPixelRate = fabs(-center[2]), dOneRightRate = fabs(-one_right[2]), dOneDownRate = fabs(-one_down[2]);
  V3 cop, rop, dop;
  for (int i = 0; i < 3; i++) { cop[i] = center[i] * dCamHeight / dPixelRate; rop[i] = one_right[i] * dCamHeight / dOneRightRate; dop[i] = one_down[i] * dCamHeight / dOneDownRate; }
  p.pix_right = rot_inv(k.pose, v3(rop[0] - cop[0], rop[1] - cop[1], rop[2] - cop[2]));
  p.pix_down = rot_inv(k.pose, v3(dop[0] - cop[0], dop[1] - cop[1], dop[2] - cop[2]));
}

V3 unit_ray(const Camera& cam, double ix, double iy) {
  double u[2];
  cam.unproject(ix, iy, u);
  const double n = sqrt(u[0] * u[0] + u[1] * u[1] + 1.0);
  return v3(u[0] / n, u[1] / n, 1.0 / n);                      // myUnproject + normalize()
}

bool System::AddPointEpipolar(int ksrc, int ktgt, int nLevel, int nCandidate) {
  // :525-703
  KeyFrame& kSrc = *kfs[ksrc]; KeyFrame& kTarget = *kfs[ktgt];
  grow_log.push_back(nLevel); grow_log.push_back((int)kSrc.cand[nLevel][nCandidate]); grow_log.push_back(0);
  int& why = grow_log.back();
  const int nLevelScale = level_scale(nLevel);
  const uint32_t cpos = kSrc.cand[nLevel][nCandidate];
  const double irLevelPos[2] = {(double)(cpos & 0xFFFF), (double)(cpos >> 16)};
  const double v2RootPos[2] = {level_zero_pos(irLevelPos[0], nLevel), level_zero_pos(irLevelPos[1], nLevel)};
  const V3 v3Ray_SC = unit_ray(camera, v2RootPos[0], v2RootPos[1]);
  const V3 v3LineDirn_TC = rot(kTarget.pose, rot_inv(kSrc.pose, v3Ray_SC));
  const double dMean = kSrc.depth_mean, dSigma = kSrc.depth_sigma;
  const double dStartDepth = std::max(p.wiggle_scale, dMean - dSigma);
  const double dEndDepth = std::min(40 * p.wiggle_scale, dMean + dSigma);
  const SE3 srcInv = inverse(kSrc.pose);
  const V3 v3CamCenter_TC = xform(kTarget.pose, v3(srcInv.t[0], srcInv.t[1], srcInv.t[2]));
  V3 v3RayStart_TC, v3RayEnd_TC;
  for (int i = 0; i < 3; i++) { v3RayStart_TC[i] = v3CamCenter_TC[i] + dStartDepth * v3LineDirn_TC[i]; v3RayEnd_TC[i] = v3CamCenter_TC[i] + dEndDepth * v3LineDirn_TC[i]; }
  if (v3RayEnd_TC[2] <= v3RayStart_TC[2]) { why = 1; return false; }
  if (v3RayEnd_TC[2] <= 0.0) { why = 1; return false; }
  if (v3RayStart_TC[2] <= 0.0) {
    const double f = 0.001 - v3RayStart_TC[2] / v3LineDirn_TC[2];
    for (int i = 0; i < 3; i++) v3RayStart_TC[i] += v3LineDirn_TC[i] * f;
  }
  const double v2A[2] = {v3RayStart_TC[0] / v3RayStart_TC[2], v3RayStart_TC[1] / v3RayStart_TC[2]};
  const double v2B[2] = {v3RayEnd_TC[0] / v3RayEnd_TC[2], v3RayEnd_TC[1] / v3RayEnd_TC[2]};
  double along[2] = {v2A[0] - v2B[0], v2A[1] - v2B[1]};
  if (along[0] * along[0] + along[1] * along[1] < 0.00000001) { why = 2; return false; }
  { const double n = sqrt(along[0] * along[0] + along[1] * along[1]); along[0] /= n; along[1] /= n; }
  const double normal[2] = {along[1], -along[0]};
  const double dNormDist = v2A[0] * normal[0] + v2A[1] * normal[1];
  if (fabs(dNormDist) > camera.largest_radius) { why = 3; return false; }
  double dMinLen = std::min(along[0] * v2A[0] + along[1] * v2A[1], along[0] * v2B[0] + along[1] * v2B[1]) - 0.05;
  double dMaxLen = std::max(along[0] * v2A[0] + along[1] * v2A[1], along[0] * v2B[0] + along[1] * v2B[1]) + 0.05;
  if (dMinLen < -2.0) dMinLen = -2.0;
  if (dMaxLen < -2.0) dMaxLen = -2.0;
  if (dMinLen > 2.0) dMinLen = 2.0;
  if (dMaxLen > 2.0) dMaxLen = 2.0;

  Finder f;
  f.P = p.patch_size; f.max_ssd = 500 * p.patch_size * p.patch_size;   // jni/PatchFinder.cc:19-20
  f.level = nLevel;                                                    // MakeTemplateCoarseNoWarp :130-142
  const int a = (int)irLevelPos[0], b = (int)irLevelPos[1], bord = f.P / 2 + 1;
  if (!(a >= bord && b >= bord && a < kSrc.w[nLevel] - bord && b < kSrc.h[nLevel] - bord)) { why = 4; return false; }   // TemplateBad
  f.tmpl.resize((size_t)f.P * f.P);
  for (int y = 0; y < f.P; y++) for (int x = 0; x < f.P; x++) f.tmpl[(size_t)y * f.P + x] = kSrc.im[nLevel][(size_t)(b - f.P / 2 + y) * kSrc.w[nLevel] + (a - f.P / 2 + x)];
  f.tsum = 0; f.tsumsq = 0;
  for (size_t i = 0; i < f.tmpl.size(); i++) { f.tsum += f.tmpl[i]; f.tsumsq += f.tmpl[i] * f.tmpl[i]; }   // MakeTemplateSums :152-164

  const std::vector<uint32_t>& vIR = kTarget.corners[nLevel];
  int nBest = -1, nBestZMSSD = f.max_ssd + 1;
  const double dMaxDistDiff = one_pixel_dist(camera) * (4.0 + 1.0 * nLevelScale);
  const double dMaxDistSq = dMaxDistDiff * dMaxDistDiff;
  for (size_t i = 0; i < vIR.size(); i++) {
    const int cx = vIR[i] & 0xFFFF, cy = vIR[i] >> 16;
    // vImplaneCorners (:612-620): UnProject of the level-zero position TRUNCATED to integer pixels (it indexes a cache image)
    double v2Im[2];
    camera.unproject((double)(int)level_zero_pos((double)cx, nLevel), (double)(int)level_zero_pos((double)cy, nLevel), v2Im);
    const double dDistDiff = dNormDist - (v2Im[0] * normal[0] + v2Im[1] * normal[1]);
    if (dDistDiff * dDistDiff > dMaxDistSq) continue;
    const double len = v2Im[0] * along[0] + v2Im[1] * along[1];
    if (len < dMinLen) continue;
    if (len > dMaxLen) continue;
    const int nZMSSD = finder_zmssd(f, kTarget.im[nLevel].data(), kTarget.w[nLevel], kTarget.h[nLevel], kTarget.w[nLevel], cx, cy);
    if (nZMSSD < nBestZMSSD) { nBest = (int)i; nBestZMSSD = nZMSSD; }
  }
  if (nBest == -1) { why = 5; return false; }

  f.coarse[0] = 0; f.coarse[1] = 0;
  finder_make_subpix(f);
  f.subpix[0] = level_zero_pos((double)(vIR[nBest] & 0xFFFF), nLevel);   // SetSubPixPos :661
  f.subpix[1] = level_zero_pos((double)(vIR[nBest] >> 16), nLevel);
  if (!finder_iterate_subpix_to_convergence(f, kTarget, 10)) { why = 6; return false; }

  double uA[2], uB[2];
  camera.unproject(v2RootPos[0], v2RootPos[1], uA);
  camera.unproject(f.subpix[0], f.subpix[1], uB);
  const V3 pB = reproject_point(mul(kSrc.pose, inverse(kTarget.pose)), uA, uB);
  const V3 v3New = xform(inverse(kTarget.pose), pB);

  MapPoint* pNew = new MapPoint();
  pNew->pos = v3New;
  pNew->src_kf = ksrc; pNew->src_level = nLevel; pNew->irx = a; pNew->iry = b;
  pNew->finder.P = p.patch_size; pNew->finder.max_ssd = f.max_ssd;
  refresh_pixel_vectors(*pNew, kSrc, unit_ray(camera, v2RootPos[0], v2RootPos[1]), unit_ray(camera, v2RootPos[0] + nLevelScale, v2RootPos[1]),
                        unit_ray(camera, v2RootPos[0], v2RootPos[1] + nLevelScale));
  pts.push_back(pNew);
  const int pid = (int)pts.size() - 1;
  Measurement m;
  m.source = SRC_ROOT; m.root[0] = v2RootPos[0]; m.root[1] = v2RootPos[1]; m.level = nLevel; m.subpix = true;
  kSrc.meas[pid] = m;
  m.source = SRC_EPIPOLAR; m.root[0] = f.subpix[0]; m.root[1] = f.subpix[1];
  kTarget.meas[pid] = m;
  pNew->meas_kfs.insert(ksrc); pNew->meas_kfs.insert(ktgt);
  new_queue.push_back(pid);                                       // mqNewQueue.push(pNew), :689
  return true;
}

// MapMaker::ReFind_Common, jni/MapMaker.cc:967-1036.  The reference's function-static PatchFinder is `refinder`.
bool System::ReFind_Common(int kidx, int pi) {
  KeyFrame& k = *kfs[kidx];
  MapPoint& pt = *pts[pi];
  if (pt.meas_kfs.count(kidx) || pt.never_retry.count(kidx)) return false;                       // :971-973
  const V3 v3Cam = xform(k.pose, pt.pos);
  if (v3Cam[2] < 0.001) { pt.never_retry.insert(kidx); return false; }                            // :979-982
  const double ip0 = v3Cam[0] / v3Cam[2], ip1 = v3Cam[1] / v3Cam[2];
  if (ip0 * ip0 + ip1 * ip1 > camera.largest_radius * camera.largest_radius) { pt.never_retry.insert(kidx); return false; }
  const Camera::Proj pr = camera.project(ip0, ip1);
  if (pr.invalid) { pt.never_retry.insert(kidx); return false; }
  if (pr.im[0] < 0 || pr.im[1] < 0 || pr.im[0] > k.w[0] || pr.im[1] > k.h[0]) { pt.never_retry.insert(kidx); return false; }
  double d[4];
  camera.derivs(pr, d);
  Finder& f = refinder;
  f.P = p.patch_size; f.max_ssd = p.patch_size * p.patch_size * 500;
  f.have_last = refind_last_point == pi;                                                          // &p == mpLastTemplateMapPoint
  finder_calc_level_and_warp(f, pt, k.pose, d);                                                   // MakeTemplateCoarse, jni/PatchFinder.cc:72-76:
  finder_make_template(f, pt, *kfs[pt.src_kf]);              // a regenerated template's own verdict replaces the bad-scale flag (:112-117)
  refind_last_point = pi;
  if (f.bad) { pt.never_retry.insert(kidx); return false; }                                       // :1004-1007
  if (!finder_find_coarse(f, pr.im, k, 4)) { pt.never_retry.insert(kidx); return false; }         // :1009-1013
  Measurement m;
  m.level = f.level; m.source = SRC_REFIND;
  if (f.level > 0) {
    finder_make_subpix(f);
    finder_iterate_subpix_to_convergence(f, k, 8);                                               // result not looked at, :1022
    m.root[0] = f.subpix[0]; m.root[1] = f.subpix[1]; m.subpix = true;
  } else { m.root[0] = f.coarse[0]; m.root[1] = f.coarse[1]; m.subpix = false; }
  k.meas[pi] = m;
  pt.meas_kfs.insert(kidx);
  return true;
}

// MapMaker::ReFindInSingleKeyFrame, jni/MapMaker.cc:1040-1056.  Bad points stay in the arrays here (HandleBadPoints) and are skipped.
int System::ReFindInSingleKeyFrame(int kidx) {
  int n = 0;
  for (int i = 0; i < (int)pts.size(); i++) if (!pts[i]->bad && ReFind_Common(kidx, i)) n++;
  return n;
}

// MapMaker::ReFindNewlyMade, jni/MapMaker.cc:1061-1081: every point of the new queue against every keyframe
void System::ReFindNewlyMade() {
  for (int pi : new_queue) {
    if (pts[pi]->bad) continue;
    for (int k = 0; k < (int)kfs.size(); k++) if (ReFind_Common(k, pi)) n_refound_new++;
  }
  new_queue.clear();
}

// MapMaker::ReFindFromFailureQueue, jni/MapMaker.cc:1083-1096.  The reference sorts the (KeyFrame*, MapPoint*) pairs by address;
// here by (keyframe index, point index), the build's stand-in for address order.
void System::ReFindFromFailureQueue() {
  if (failure_queue.empty()) return;
  std::sort(failure_queue.begin(), failure_queue.end());
  for (auto& e : failure_queue) if (ReFind_Common(e.first, e.second)) n_refound_failed++;
  failure_queue.clear();
}

// One pass through the idle jobs of MapMaker::run (jni/MapMaker.cc:94-117) with an empty keyframe queue
void System::IdleIteration() { for (int job = 0; job < 4; job++) IdleJob(job); }

// One of the four jobs, so that a test can compare (and re-synchronise) after each.  HandleBadPoints (:117) runs after every job
// here instead of once per pass: it only acts on bBad flags, which nothing between the jobs reads differently.
void System::IdleJob(int job) {
  if (!map_good) return;
  if (job == 0 && !ba_converged_recent) { BundleAdjustRecent(); n_ba_recent_idle++; }                         // :97-98
  if (job == 1 && ba_converged_recent) ReFindNewlyMade();                                                    // :102-103
  if (job == 2 && ba_converged_recent && !ba_converged_full) { BundleAdjustAll(); n_ba_all++; }                // :107-108
  if (job == 3 && ba_converged_recent && ba_converged_full) { if (idle_count++ % 20 == 0) ReFindFromFailureQueue(); }   // :112-113, rand() % 20 == 0 made deterministic
  HandleBadPoints();                                                                                         // :117
}

int System::AddSomeMapPoints(int nLevel) {
  // :424-437
  const int ksrc = (int)kfs.size() - 1;
  const int ktgt = ClosestKeyFrame(ksrc);
  if (ktgt < 0) return 0;
  ThinCandidates(*kfs[ksrc], nLevel);
  int n = 0;
  for (int i = 0; i < (int)kfs[ksrc]->cand[nLevel].size(); i++) if (AddPointEpipolar(ksrc, ktgt, nLevel, i)) n++;
  return n;
}

}  // namespace orc

extern "C" void orc_reproject_point(const double AfromB12[12], const double v2A[2], const double v2B[2], double out3[3]) {
  orc::SE3 T;
  for (int i = 0; i < 9; i++) T.R[i] = AfromB12[i];
  for (int i = 0; i < 3; i++) T.t[i] = AfromB12[9 + i];
  const orc::V3 r = orc::reproject_point(T, v2A, v2B);
  for (int i = 0; i < 3; i++) out3[i] = r[i];
}

extern "C" int orc_sys_get_grow_log(void* sys, int* out3, int cap) {
  orc::System* S = (orc::System*)sys;
  const int n = (int)S->grow_log.size() / 3;
  for (int i = 0; i < n && i < cap; i++) for (int k = 0; k < 3; k++) out3[3 * i + k] = S->grow_log[3 * i + k];
  return n;
}
