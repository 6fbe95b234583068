// ORACLE (test infrastructure): SmallBlurryImage and the rotation prior of the tracker's motion model.
//   SmallBlurryImage::MakeFromKF / MakeJacs / IteratePosRelToTarget / SE3fromSE2   jni/SmallBlurryImage.cc:20-333
//   Tracker::CalcSBIRotation                                                       jni/Tracker.cc:885-893
// Third-party arithmetic restated here (parity unpinned, OpenCV 2.4.x is not in the tree):
//   * cv::resize(level 3 -> half size) (:30): the exact 2:1 area filter (a+b+c+d+2)>>2, as for the pyramid;
//   * cv::GaussianBlur(9x9, sigma 0.75, BORDER_REPLICATE) on CV_32F (:52): separable, fp32 kernel from
//     cv::getGaussianKernel (exp(-x^2/(2 sigma^2)) in double, stored and normalised in float), each pass evaluated in
//     float as k[4]*x[0] + sum_j k[4+j]*(x[+j] + x[-j]) (the symmetric row/column filter form);
//   * Eigen's fixed 4x4 inverse (:207): pivoted Gaussian elimination.
#include "ptam_system.hpp"
#include "ptam_oracle.h"

namespace orc {

struct SE2 { double R[4]; double t[2]; SE2() { R[0] = R[3] = 1; R[1] = R[2] = 0; t[0] = t[1] = 0; } };   // jni/RT.h:488-529
static SE2 se2_mul(const SE2& a, const SE2& b) {                                                          // :516-523
  SE2 r;
  r.R[0] = a.R[0] * b.R[0] + a.R[1] * b.R[2]; r.R[1] = a.R[0] * b.R[1] + a.R[1] * b.R[3];
  r.R[2] = a.R[2] * b.R[0] + a.R[3] * b.R[2]; r.R[3] = a.R[2] * b.R[1] + a.R[3] * b.R[3];
  r.t[0] = a.t[0] + (a.R[0] * b.t[0] + a.R[1] * b.t[1]);
  r.t[1] = a.t[1] + (a.R[2] * b.t[0] + a.R[3] * b.t[1]);
  return r;
}
static SE2 se2_inverse(const SE2& a) {                                                                   // :506-511
  SE2 r;
  r.R[0] = a.R[0]; r.R[1] = a.R[2]; r.R[2] = a.R[1]; r.R[3] = a.R[3];
  r.t[0] = -(r.R[0] * a.t[0] + r.R[1] * a.t[1]);
  r.t[1] = -(r.R[2] * a.t[0] + r.R[3] * a.t[1]);
  return r;
}

void sbi_gauss_kernel9(double sigma, float k[9]) {   // cv::getGaussianKernel(9, sigma, CV_32F)
  const double scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < 9; i++) { const double x = i - 4.0; k[i] = (float)std::exp(scale2X * x * x); sum += k[i]; }
  sum = 1.0 / sum;
  for (int i = 0; i < 9; i++) k[i] = (float)(k[i] * sum);
}

// MakeFromKF (:20-55), blur <= 2 branch only (the tracker uses 0.75, jni/Tracker.cc:87)
void sbi_make(SBI& s, const uint8_t* l3, int w3, int h3, double blur) {
  s.w = w3 / 2; s.h = h3 / 2;                        // mirSize, :22-25
  s.small.assign((size_t)s.w * s.h, 0);
  orc_halfsample(l3, w3, h3, w3, s.small.data(), s.w);
  unsigned int nSum = 0;
  for (size_t i = 0; i < s.small.size(); i++) nSum += s.small[i];
  const float fMean = ((float)nSum) / (s.h * s.w);
  std::vector<float> t((size_t)s.w * s.h), row((size_t)s.w * s.h);
  for (size_t i = 0; i < t.size(); i++) t[i] = s.small[i] - fMean;
  float k[9];
  sbi_gauss_kernel9(blur, k);
  auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
  for (int y = 0; y < s.h; y++)
    for (int x = 0; x < s.w; x++) {
      const float* r = &t[(size_t)y * s.w];
      float acc = k[4] * r[x];
      for (int j = 1; j <= 4; j++) acc += k[4 + j] * (r[clampi(x + j, s.w - 1)] + r[clampi(x - j, s.w - 1)]);
      row[(size_t)y * s.w + x] = acc;
    }
  s.tmpl.assign(t.size(), 0.f);
  for (int y = 0; y < s.h; y++)
    for (int x = 0; x < s.w; x++) {
      float acc = k[4] * row[(size_t)y * s.w + x];
      for (int j = 1; j <= 4; j++) acc += k[4 + j] * (row[(size_t)clampi(y + j, s.h - 1) * s.w + x] + row[(size_t)clampi(y - j, s.h - 1) * s.w + x]);
      s.tmpl[(size_t)y * s.w + x] = acc;
    }
  s.made_jacs = false;
}

void sbi_make_jacs(SBI& s) {                        // :58-79
  s.jacs.assign((size_t)s.w * s.h * 2, 0.f);
  for (int x = 0; x < s.w; x++)
    for (int y = 0; y < s.h; y++)
      if (x >= 1 && y >= 1 && x < s.w - 1 && y < s.h - 1) {
        s.jacs[((size_t)y * s.w + x) * 2 + 0] = s.tmpl[(size_t)y * s.w + x + 1] - s.tmpl[(size_t)y * s.w + x - 1];
        s.jacs[((size_t)y * s.w + x) * 2 + 1] = s.tmpl[(size_t)(y + 1) * s.w + x] - s.tmpl[(size_t)(y - 1) * s.w + x];
      }
  s.made_jacs = true;
}

// transform_image for CV_32FC1 (jni/vision/ImageHandler.cpp:21-113) with sample(double&) :3-10
static void transform_image_f32(const std::vector<float>& in, int iw, int ih, std::vector<float>& out, int w, int h,
                                const double M[4], const double inOrig[2], const double outOrig[2], double defaultValue) {
  const double across[2] = {M[0], M[2]}, down[2] = {M[1], M[3]};
  double p[2] = {inOrig[0] - (M[0] * outOrig[0] + M[1] * outOrig[1]), inOrig[1] - (M[2] * outOrig[0] + M[3] * outOrig[1])};
  const double cr[2] = {down[0] - w * across[0], down[1] - w * across[1]};
  const float x_bound = iw - 1, y_bound = ih - 1;
  // the "completely inside" fast path of the reference (:55-75) evaluates the same expression per pixel
  for (int i = 0; i < h; ++i, p[0] += cr[0], p[1] += cr[1])
    for (int j = 0; j < w; ++j, p[0] += across[0], p[1] += across[1]) {
      if (0 <= p[0] && 0 <= p[1] && p[0] < x_bound && p[1] < y_bound) {
        double x = p[0], y = p[1];
        const int lx = (int)x, ly = (int)y;
        x -= lx; y -= ly;
        const float* q = &in[(size_t)ly * iw + lx];
        const double r = (double)((1 - y) * ((1 - x) * q[0] + x * q[1]) + y * ((1 - x) * q[iw] + x * q[iw + 1]));
        out[(size_t)i * w + j] = (float)r;
      } else out[(size_t)i * w + j] = (float)defaultValue;
    }
}

// IteratePosRelToTarget (:98-222): ESM alignment of `cur` to `other`; returns the SE2 and the final score
static SE2 sbi_iterate(const SBI& cur, const SBI& other, int nIterations, double* score) {
  SE2 se2CtoC, se2WfromC;
  const double irCenter[2] = {cur.w / 2.0, cur.h / 2.0};          // mirSize / 2 on a double vector
  se2WfromC.t[0] = irCenter[0]; se2WfromC.t[1] = irCenter[1];
  double dMeanOffset = 0.0, dFinalScore = 0.0;
  std::vector<float> warped((size_t)cur.w * cur.h);
  const int W = cur.w, H = cur.h;
  for (int it = 0; it < nIterations; it++) {
    dFinalScore = 0.0;
    double v4Accum[4] = {0, 0, 0, 0}, tri[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const SE2 X = se2_mul(se2_mul(se2WfromC, se2CtoC), se2_inverse(se2WfromC));
    const double zero[2] = {0, 0};
    transform_image_f32(cur.tmpl, W, H, warped, W, H, X.R, X.t, zero, -9e20f);
    for (int i = 0; i < W; i++)
      for (int j = 0; j < H; j++) {
        if (!(i >= 1 && j >= 1 && i < W - 1 && j < H - 1)) continue;
        const float l = warped[(size_t)j * W + i - 1], r = warped[(size_t)j * W + i + 1];
        const float u = warped[(size_t)(j - 1) * W + i], d = warped[(size_t)(j + 1) * W + i], here = warped[(size_t)j * W + i];
        if (l + r + u + d + here < -9999.9) continue;
        const double g0 = r - l, g1 = d - u;                        // float differences widened (Vector2d from floats)
        const double s0 = 0.25 * (g0 + other.jacs[((size_t)j * W + i) * 2 + 0]);
        const double s1 = 0.25 * (g1 + other.jacs[((size_t)j * W + i) * 2 + 1]);
        const double J[4] = {s0, s1, -((double)j - irCenter[1]) * s0 + ((double)i - irCenter[0]) * s1, 1.0};
        const double dDiff = warped[(size_t)j * W + i] - other.tmpl[(size_t)j * W + i] + dMeanOffset;
        dFinalScore += dDiff * dDiff;
        for (int k = 0; k < 4; k++) v4Accum[k] += dDiff * J[k];
        tri[0] += J[0] * J[0]; tri[1] += J[1] * J[0]; tri[2] += J[1] * J[1]; tri[3] += J[2] * J[0]; tri[4] += J[2] * J[1];
        tri[5] += J[2] * J[2]; tri[6] += J[0]; tri[7] += J[1]; tri[8] += J[2]; tri[9] += 1.0;
      }
    double m4[16], upd[4];
    int v = 0;
    for (int j = 0; j < 4; j++) for (int i = 0; i <= j; i++) { m4[j * 4 + i] = m4[i * 4 + j] = tri[v++]; }
    for (int k = 0; k < 4; k++) upd[k] = v4Accum[k];
    if (!lu_solve(m4, upd, 4)) { upd[0] = upd[1] = upd[2] = upd[3] = 0; }
    SE2 U;
    U.t[0] = -upd[0]; U.t[1] = -upd[1];
    const double a = -upd[2];
    U.R[0] = U.R[3] = tcos(a); U.R[2] = tsin(a); U.R[1] = -U.R[2];      // mySO2::exp, jni/RT.h:459-465
    se2CtoC = se2_mul(se2CtoC, U);
    dMeanOffset -= upd[3];
  }
  if (score) *score = dFinalScore;
  return se2CtoC;
}

// SE3fromSE2 (:249-333): the camera rotation that produces the image-plane motion of two points 5 px either side of the centre
static SE3 sbi_se3_from_se2(const SE2& se2, const Camera& full, int w, int h, bool quirk_int_radius) {
  Camera cam;
  cam.init(full.params, w, h, quirk_int_radius);                  // camera.SetImageSize(mirSize)
  const double c[2] = {w / 2.0, h / 2.0};
  const double offs[2][2] = {{5, 0}, {-5, 0}};
  double turned[2][2], orig[2][3];
  for (int k = 0; k < 2; k++) {
    turned[k][0] = c[0] + (se2.t[0] + (se2.R[0] * offs[k][0] + se2.R[1] * offs[k][1]));
    turned[k][1] = c[1] + (se2.t[1] + (se2.R[2] * offs[k][0] + se2.R[3] * offs[k][1]));
    double up[2];
    cam.unproject(c[0] + offs[k][0], c[1] + offs[k][1], up);
    orig[k][0] = up[0]; orig[k][1] = up[1]; orig[k][2] = 1.0;
  }
  SE3 so3;                                                         // rotation only
  for (int it = 0; it < 3; it++) {
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, vec[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) C[i * 3 + i] += 10.0;               // wls.add_prior(10.0)
    for (int k = 0; k < 2; k++) {
      const V3 vc = rot(so3, v3(orig[k][0], orig[k][1], orig[k][2]));
      const Camera::Proj pr = cam.project(vc[0] / vc[2], vc[1] / vc[2]);
      const double err[2] = {turned[k][0] - pr.im[0], turned[k][1] - pr.im[1]};
      double dd[4];
      cam.derivs(pr, dd);
      double J[2][3];
      const double ooz = 1.0 / vc[2];
      for (int m = 0; m < 3; m++) {
        double mot[3] = {0, 0, 0};                                  // mySO3::generator_field, jni/RT.h:70-77
        mot[(m + 1) % 3] = -vc[(m + 2) % 3]; mot[(m + 2) % 3] = vc[(m + 1) % 3];
        const double f0 = (mot[0] - vc[0] * mot[2] * ooz) * ooz, f1 = (mot[1] - vc[1] * mot[2] * ooz) * ooz;
        J[0][m] = dd[0] * f0 + dd[1] * f1; J[1][m] = dd[2] * f0 + dd[3] * f1;
      }
      for (int row = 0; row < 2; row++)                             // add_mJ(err, J, 1.0), jni/myWLS.h:39-50
        for (int r = 0; r < 3; r++) {
          const double Jw = 1.0 * J[row][r];
          vec[r] += err[row] * Jw;
          for (int cc = r; cc < 3; cc++) C[r * 3 + cc] += Jw * J[row][cc];
        }
    }
    for (int r = 1; r < 3; r++) for (int cc = 0; cc < r; cc++) C[r * 3 + cc] = C[cc * 3 + r];
    double mu[3] = {vec[0], vec[1], vec[2]};
    if (!lu_solve(C, mu, 3)) mu[0] = mu[1] = mu[2] = 0;
    double Rn[9];
    so3_exp(mu, Rn);
    SE3 e; for (int i = 0; i < 9; i++) e.R[i] = Rn[i];
    so3 = mul(e, so3);
  }
  return so3;
}

// Tracker::CalcSBIRotation (jni/Tracker.cc:885-893): ln of the SE3 that explains the SBI alignment this <- last
void calc_sbi_rotation(const SBI& cur, SBI& last, const Camera& cam, bool quirk_int_radius, double out6[6], double* score) {
  sbi_make_jacs(last);
  const SE2 r = sbi_iterate(cur, last, 6, score);
  const SE3 adj = sbi_se3_from_se2(r, cam, cur.w, cur.h, quirk_int_radius);
  se3_ln(adj, out6);
}

}  // namespace orc

// ---- C API (unit level) ---------------------------------------------------------------------------------------------
extern "C" int orc_sbi_make(const uint8_t* level3, int w3, int h3, double blur, uint8_t* small_out, float* tmpl_out) {
  orc::SBI s;
  orc::sbi_make(s, level3, w3, h3, blur);
  if (small_out) memcpy(small_out, s.small.data(), s.small.size());
  if (tmpl_out) memcpy(tmpl_out, s.tmpl.data(), s.tmpl.size() * sizeof(float));
  return s.w | (s.h << 16);
}

extern "C" void orc_sbi_rotation(const uint8_t* cur_l3, const uint8_t* last_l3, int w3, int h3, double blur, const double cam5[5],
                                 int quirks, double out6[6], double* score) {
  orc::SBI a, b;
  orc::sbi_make(a, cur_l3, w3, h3, blur);
  orc::sbi_make(b, last_l3, w3, h3, blur);
  orc::Camera cam;
  cam.init(cam5, w3 * 8, h3 * 8, (quirks & ORC_Q_CAM_INT_RADIUS) != 0);
  orc::calc_sbi_rotation(a, b, cam, (quirks & ORC_Q_CAM_INT_RADIUS) != 0, out6, score);
}
