// SmallBlurryImage and the rotation prior of the motion model, one workgroup per stream on the
// front-end stream, one frame ahead of the tracker:
//   SmallBlurryImage::MakeFromKF (jni/SmallBlurryImage.cc:20-55)  level 3 -> half size, zero mean, 9x9 Gaussian sigma 0.75
//   MakeJacs (:58-79)                                              central differences of the template
//   IteratePosRelToTarget (:98-222)                                6 ESM iterations aligning this frame's SBI to the last one's
//   SE3fromSE2 (:249-333) + Tracker::CalcSBIRotation (jni/Tracker.cc:885-893)   -> mv6SBIRot, read by k_motion (ApplyMotionModel)
// cv::resize / cv::GaussianBlur / Eigen's 4x4 inverse are third-party arithmetic, restated exactly as in oracle/sbi.cpp
// (same float expressions in the same order).  The whole stage is bit-exact with the oracle: the sample positions of
// transform_image are accumulated pixel by pixel by two lanes, and the fifteen ESM sums are taken in the reference's
// column-major order by fifteen lanes of wave 0 from per-pixel records the other threads stage in LDS.
#include "vslam_internal.h"

#define SBI_THREADS 256
#define SBI_WAVES (SBI_THREADS / 64)
#define SBI_MAX_PIX 4096          // (w/16) * (h/16) small-image pixels: 1200 at 640x480, 3600 at 1280x720
#define SBI_REC 15                // per-pixel record: dDiff*J[0..3], the ten triangle products, dDiff^2
#define SBI_CHUNK 128              // pixels staged per round of the sequential sums (two buffers: one filled while the other is added)
// dynamic LDS: t0[N] t1[N] floats, then max(2N sample positions, two chunks of records) doubles
static size_t sbi_lds_bytes(int N) {
  const size_t a = (size_t)2 * N * sizeof(float), pos = (size_t)2 * N * sizeof(double), rec = (size_t)2 * SBI_CHUNK * SBI_REC * sizeof(double);
  return ((a + 7) & ~(size_t)7) + (pos > rec ? pos : rec);
}

struct SbiArgs {
  const uint8_t* l3; size_t l3_sstride; int l3_pitch, w3, h3;
  uint8_t* small; float* tmpl; float* jacs; double* rot;          // this frame
  const float* last_tmpl; const float* last_jacs;                 // previous frame (== this frame's on the very first frame)
  float k[9];                                                     // cv::getGaussianKernel(9, 0.75, CV_32F)
  CamModel cam;                                                   // the camera at the small image's size (SE3fromSE2 :254)
};

struct Se2 { double R[4]; double t[2]; };
DEVFN Se2 se2_mul(const Se2& a, const Se2& b) {                    // jni/RT.h:516-523
  Se2 r;
  r.R[0] = a.R[0] * b.R[0] + a.R[1] * b.R[2]; r.R[1] = a.R[0] * b.R[1] + a.R[1] * b.R[3];
  r.R[2] = a.R[2] * b.R[0] + a.R[3] * b.R[2]; r.R[3] = a.R[2] * b.R[1] + a.R[3] * b.R[3];
  r.t[0] = a.t[0] + (a.R[0] * b.t[0] + a.R[1] * b.t[1]);
  r.t[1] = a.t[1] + (a.R[2] * b.t[0] + a.R[3] * b.t[1]);
  return r;
}
DEVFN Se2 se2_inverse(const Se2& a) {                              // :506-511
  Se2 r;
  r.R[0] = a.R[0]; r.R[1] = a.R[2]; r.R[2] = a.R[1]; r.R[3] = a.R[3];
  r.t[0] = -(r.R[0] * a.t[0] + r.R[1] * a.t[1]);
  r.t[1] = -(r.R[2] * a.t[0] + r.R[3] * a.t[1]);
  return r;
}
DEVFN int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

__global__ __launch_bounds__(SBI_THREADS) void k_sbi(SbiArgs a) {
  extern __shared__ double sbi_dyn[];
  __shared__ double sums[16];
  __shared__ unsigned int isum[SBI_WAVES];
  __shared__ Se2 shX;
  __shared__ double sh_mean_off;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = a.w3 / 2, H = a.h3 / 2, N = W * H;
  float* t0 = (float*)sbi_dyn;           // zero-mean small image, later the warped template
  float* t1 = t0 + N;                    // row pass, later this frame's template
  double* wk = sbi_dyn + (2 * N + 1) / 2;   // sample positions [N][2], then the records [2][SBI_CHUNK][SBI_REC]
  const uint8_t* l3 = a.l3 + (size_t)s * a.l3_sstride;
  uint8_t* small = a.small + (size_t)s * N;
  float* tmpl = a.tmpl + (size_t)s * N;
  float* jacs = a.jacs + (size_t)s * N * 2;
  const float* ltmpl = a.last_tmpl + (size_t)s * N;
  const float* ljacs = a.last_jacs + (size_t)s * N * 2;

  // ---- MakeFromKF: cv::resize to half size (2:1 area filter), mean, zero-mean float image ----
  unsigned int part = 0;
  for (int i = tid; i < N; i += SBI_THREADS) {
    const int y = i / W, x = i - y * W;
    const uint8_t* r0 = l3 + (size_t)(2 * y) * a.l3_pitch + 2 * x;
    const unsigned v = ((unsigned)r0[0] + r0[1] + r0[a.l3_pitch] + r0[a.l3_pitch + 1] + 2u) >> 2;
    small[i] = (uint8_t)v;
    t0[i] = (float)v;
    part += v;
  }
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) isum[wave] = part;
  __syncthreads();
  unsigned int nSum = 0;
  for (int w = 0; w < SBI_WAVES; w++) nSum += isum[w];
  const float fMean = ((float)nSum) / (float)(H * W);                // :39
  for (int i = tid; i < N; i += SBI_THREADS) t0[i] = t0[i] - fMean;
  __syncthreads();
  // ---- cv::GaussianBlur 9x9, sigma 0.75, BORDER_REPLICATE: row pass, then column pass (see header) ----
  for (int i = tid; i < N; i += SBI_THREADS) {
    const int y = i / W, x = i - y * W;
    const float* r = t0 + y * W;
    float acc = a.k[4] * r[x];
    for (int j = 1; j <= 4; j++) acc += a.k[4 + j] * (r[clampi(x + j, W - 1)] + r[clampi(x - j, W - 1)]);
    t1[i] = acc;
  }
  __syncthreads();
  for (int i = tid; i < N; i += SBI_THREADS) {
    const int y = i / W, x = i - y * W;
    float acc = a.k[4] * t1[i];
    for (int j = 1; j <= 4; j++) acc += a.k[4 + j] * (t1[clampi(y + j, H - 1) * W + x] + t1[clampi(y - j, H - 1) * W + x]);
    t0[i] = acc;
  }
  __syncthreads();
  for (int i = tid; i < N; i += SBI_THREADS) { t1[i] = t0[i]; tmpl[i] = t0[i]; }   // t1 = mimTemplate of this frame
  __syncthreads();
  // ---- MakeJacs of this frame's template (it is the "last frame" of the next call) ----
  for (int i = tid; i < N; i += SBI_THREADS) {
    const int y = i / W, x = i - y * W;
    float gx = 0.f, gy = 0.f;
    if (x >= 1 && y >= 1 && x < W - 1 && y < H - 1) { gx = t1[i + 1] - t1[i - 1]; gy = t1[i + W] - t1[i - W]; }
    jacs[2 * i] = gx; jacs[2 * i + 1] = gy;
  }
  // on the first frame "last" is this frame itself (jni/Tracker.cc:90-92): wait for our own stores
  __threadfence();
  __syncthreads();

  // ---- IteratePosRelToTarget(last, 6) ----
  const double cx = W / 2.0, cy = H / 2.0;                            // irCenter = mirSize / 2
  Se2 CtoC; CtoC.R[0] = CtoC.R[3] = 1.0; CtoC.R[1] = CtoC.R[2] = 0.0; CtoC.t[0] = CtoC.t[1] = 0.0;   // thread 0's copy is the one used
  Se2 WfromC = CtoC; WfromC.t[0] = cx; WfromC.t[1] = cy;
  if (tid == 0) sh_mean_off = 0.0;
  double final_score = 0.0;
  for (int it = 0; it < 6; it++) {
    if (tid == 0) shX = se2_mul(se2_mul(WfromC, CtoC), se2_inverse(WfromC));
    __syncthreads();
    // transform_image<float> (jni/vision/ImageHandler.cpp:21-113): the reference accumulates the sample position pixel by
    // pixel (p += across, a carriage return per row); lane 0 walks x and lane 1 walks y exactly so, the rest sample.
    const double mean_off = sh_mean_off;
    if (tid < 2) {
      const double across = tid ? shX.R[2] : shX.R[0], down = tid ? shX.R[3] : shX.R[1];
      const double cr = down - W * across;
      double p = shX.t[tid];
      double* o = wk + tid;
      for (int i = 0; i < H; i++, p += cr) {
#pragma unroll 8
        for (int j = 0; j < W; j++, p += across, o += 2) *o = p;
      }
    }
    __syncthreads();
    {
      const float x_bound = (float)(W - 1), y_bound = (float)(H - 1);
      for (int idx = tid; idx < N; idx += SBI_THREADS) {
        const double px = wk[2 * idx], py = wk[2 * idx + 1];
        float v = -9e20f;
        if (0 <= px && 0 <= py && px < x_bound && py < y_bound) {
          double x = px, y = py;
          const int lx = (int)x, ly = (int)y;
          x -= lx; y -= ly;
          const float* q = t1 + ly * W + lx;
          v = (float)((1 - y) * ((1 - x) * q[0] + x * q[1]) + y * ((1 - x) * q[W] + x * q[W + 1]));
        }
        t0[idx] = v;
      }
    }
    __syncthreads();
    // the sums of :133-176 in the reference's order (columns outer, rows inner): the upper two waves stage one pixel's
    // fifteen products each (zeros for a skipped pixel: x + 0.0 == x) into one record buffer while lanes 0..14 of wave 0
    // add the other buffer's records one after the other; the template operands of a chunk are fetched a round ahead
    const int nch = (N + SBI_CHUNK - 1) / SBI_CHUNK;
    const bool producer = tid >= SBI_THREADS - SBI_CHUNK;
    const int pt = tid - (SBI_THREADS - SBI_CHUNK);
    double acc = 0.0;
    float ltn = 0.f, j0n = 0.f, j1n = 0.f;
    auto interior = [&](int q, int& i, int& j, int& idx) {
      i = q / H; j = q - i * H; idx = j * W + i;
      return q < N && i >= 1 && j >= 1 && i < W - 1 && j < H - 1;
    };
    auto fetch = [&](int c) {
      int i, j, idx;
      ltn = j0n = j1n = 0.f;
      if (interior(c * SBI_CHUNK + pt, i, j, idx)) { ltn = ltmpl[idx]; j0n = ljacs[2 * idx]; j1n = ljacs[2 * idx + 1]; }
    };
    if (producer) fetch(0);
    for (int c = 0; c <= nch; c++) {
      if (producer) {
        if (c < nch) {
          const float lt = ltn, lj0 = j0n, lj1 = j1n;
          fetch(c + 1);
          const int q = c * SBI_CHUNK + pt;
          int i, j, idx;
          const bool in = interior(q, i, j, idx);
          double r[SBI_REC];
#pragma unroll
          for (int k = 0; k < SBI_REC; k++) r[k] = 0.0;
          if (in) {
            const float l = t0[idx - 1], rr = t0[idx + 1], u = t0[idx - W], d = t0[idx + W], here = t0[idx];
            if (!(l + rr + u + d + here < -9999.9)) {
              const double g0 = rr - l, g1 = d - u;
              const double s0 = 0.25 * (g0 + lj0), s1 = 0.25 * (g1 + lj1);
              const double J0 = s0, J1 = s1, J2 = -((double)j - cy) * s0 + ((double)i - cx) * s1;
              const double dDiff = here - lt + mean_off;
              r[14] = dDiff * dDiff;
              r[0] = dDiff * J0; r[1] = dDiff * J1; r[2] = dDiff * J2; r[3] = dDiff;
              r[4] = J0 * J0; r[5] = J1 * J0; r[6] = J1 * J1; r[7] = J2 * J0; r[8] = J2 * J1; r[9] = J2 * J2;
              r[10] = J0; r[11] = J1; r[12] = J2; r[13] = 1.0;
            }
          }
          if (q < N) {
            double* o = wk + ((c & 1) * SBI_CHUNK + pt) * SBI_REC;
#pragma unroll
            for (int k = 0; k < SBI_REC; k++) o[k] = r[k];
          }
        }
      } else if (tid < SBI_REC && c > 0) {
        const int cnt = min(SBI_CHUNK, N - (c - 1) * SBI_CHUNK);
        const double* rp = wk + ((c - 1) & 1) * SBI_CHUNK * SBI_REC + tid;
        if (cnt == SBI_CHUNK) {
#pragma unroll
          for (int g = 0; g < SBI_CHUNK / 8; g++) {
            double x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = rp[(g * 8 + k) * SBI_REC];
#pragma unroll
            for (int k = 0; k < 8; k++) acc += x[k];
          }
        } else {
          for (int p = 0; p < cnt; p++) acc += rp[p * SBI_REC];
        }
      }
      __syncthreads();
    }
    if (tid < SBI_REC) sums[tid] = acc;
    __syncthreads();
    if (tid == 0) {
      double v[16];
      for (int k = 0; k < SBI_REC; k++) v[k] = sums[k];
      double m4[16], upd[4] = {v[0], v[1], v[2], v[3]};
      int q = 4;
      for (int j = 0; j < 4; j++) for (int i = 0; i <= j; i++) { m4[j * 4 + i] = v[q]; m4[i * 4 + j] = v[q]; q++; }
      if (!lu_solve_n(m4, upd, 4)) { upd[0] = upd[1] = upd[2] = upd[3] = 0.0; }
      Se2 U;
      U.t[0] = -upd[0]; U.t[1] = -upd[1];
      const double ang = -upd[2];
      U.R[0] = U.R[3] = vlm::vcos(ang); U.R[2] = vlm::vsin(ang); U.R[1] = -U.R[2];   // mySO2::exp, jni/RT.h:459-465
      CtoC = se2_mul(CtoC, U);
      sh_mean_off -= upd[3];
      final_score = v[14];
    }
    __syncthreads();
  }
  if (tid != 0) return;
  // ---- SE3fromSE2 (:249-333) + ln ----
  const double offs[2][2] = {{5, 0}, {-5, 0}};
  double turned[2][2], orig[2][3];
  for (int k = 0; k < 2; k++) {
    turned[k][0] = cx + (CtoC.t[0] + (CtoC.R[0] * offs[k][0] + CtoC.R[1] * offs[k][1]));
    turned[k][1] = cy + (CtoC.t[1] + (CtoC.R[2] * offs[k][0] + CtoC.R[3] * offs[k][1]));
    double up[2];
    cam_unproject(a.cam, cx + offs[k][0], cy + offs[k][1], up);
    orig[k][0] = up[0]; orig[k][1] = up[1]; orig[k][2] = 1.0;
  }
  Pose so3; for (int i = 0; i < 9; i++) so3.R[i] = (i % 4 == 0) ? 1.0 : 0.0; so3.t[0] = so3.t[1] = so3.t[2] = 0.0;
  for (int it = 0; it < 3; it++) {
    double C[9] = {10.0, 0, 0, 0, 10.0, 0, 0, 0, 10.0}, vec[3] = {0, 0, 0};   // wls.add_prior(10.0)
    for (int k = 0; k < 2; k++) {
      double vc[3];
      pose_rot(so3, orig[k], vc);
      const CamProj pr = cam_project(a.cam, vc[0] / vc[2], vc[1] / vc[2]);
      const double err[2] = {turned[k][0] - pr.im[0], turned[k][1] - pr.im[1]};
      double dd[4];
      cam_derivs(a.cam, pr, dd);
      double J[2][3];
      const double ooz = 1.0 / vc[2];
      for (int m = 0; m < 3; m++) {
        double mot[3] = {0, 0, 0};                                    // mySO3::generator_field, jni/RT.h:70-77
        mot[(m + 1) % 3] = -vc[(m + 2) % 3]; mot[(m + 2) % 3] = vc[(m + 1) % 3];
        const double f0 = (mot[0] - vc[0] * mot[2] * ooz) * ooz, f1 = (mot[1] - vc[1] * mot[2] * ooz) * ooz;
        J[0][m] = dd[0] * f0 + dd[1] * f1; J[1][m] = dd[2] * f0 + dd[3] * f1;
      }
      for (int row = 0; row < 2; row++)
        for (int r = 0; r < 3; r++) {
          const double Jw = 1.0 * J[row][r];
          vec[r] += err[row] * Jw;
          for (int c = r; c < 3; c++) C[r * 3 + c] += Jw * J[row][c];
        }
    }
    for (int r = 1; r < 3; r++) for (int c = 0; c < r; c++) C[r * 3 + c] = C[c * 3 + r];
    double mu[3] = {vec[0], vec[1], vec[2]};
    if (!lu_solve_n(C, mu, 3)) mu[0] = mu[1] = mu[2] = 0.0;
    Pose e; so3_exp(mu, e.R); e.t[0] = e.t[1] = e.t[2] = 0.0;
    so3 = pose_mul(e, so3);
  }
  double out6[6];
  se3_ln(so3, out6);
  double* rot = a.rot + (size_t)s * 8;
  for (int i = 0; i < 6; i++) rot[i] = out6[i];
  rot[6] = final_score; rot[7] = 0.0;
}

int fe_sbi(vslam_system* sys, const FrameDev& last) {
  const LevelGeom& g3 = sys->geom[3];
  const int W = g3.w / 2, H = g3.h / 2;
  if (W * H > SBI_MAX_PIX || H > SBI_THREADS) { vslam_set_error("use_sbi: small image %d x %d exceeds %d pixels", W, H, SBI_MAX_PIX); return VSLAM_E_INVALID; }
  SbiArgs a;
  a.l3 = sys->fr.img[3]; a.l3_sstride = sys->fr.img_sstride[3]; a.l3_pitch = sys->fr.img_pitch[3]; a.w3 = g3.w; a.h3 = g3.h;
  a.small = sys->fr.sbi_small; a.tmpl = sys->fr.sbi_tmpl; a.jacs = sys->fr.sbi_jacs; a.rot = sys->fr.sbi_rot;
  a.last_tmpl = last.sbi_tmpl; a.last_jacs = last.sbi_jacs;
  {                                                                  // cv::getGaussianKernel(9, 0.75, CV_32F); gvdSBIBlur, jni/Tracker.cc:87
    const double sigma = 0.75, scale2X = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < 9; i++) { const double x = i - 4.0; a.k[i] = (float)exp(scale2X * x * x); sum += a.k[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < 9; i++) a.k[i] = (float)(a.k[i] * sum);
  }
  cam_fill(a.cam, sys->p.cam, W, H, sys->p.quirks);
  const size_t lds = sbi_lds_bytes(W * H);
  if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_sbi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_sbi, dim3(sys->S), dim3(SBI_THREADS), lds, sys->fe_stream, a);
  HIPCHK(hipGetLastError());
  return VSLAM_OK;
}
