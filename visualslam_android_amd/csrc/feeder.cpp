// Synthetic frame feeder (include/vslam_feeder.h): host-only C++, replaces the Android camera plumbing.
#include "../../include/vslam_feeder.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct Rng {  // xorshift64*
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) { if (!s) s = 1; }
  uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
  double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
  int range(int lo, int hi) { return lo + (int)(uni() * (hi - lo)); }
};

inline uint32_t hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return (uint32_t)x;
}

constexpr int TEX = 4096;          // texels per side
constexpr double TEX_M = 4.0;      // metres per side: plane patch [-2,2]^2
constexpr double PPM = TEX / TEX_M;

}  // namespace

struct vslam_feeder {
  int w, h, noise;
  uint64_t seed;
  double focal[2], center[2], inv_focal[2], ww, winv, two_tan, one_over_2tan;
  std::vector<uint8_t> tex;
  std::vector<float> rays;   // per pixel: z = 1 plane coordinates of the (distorted) pixel
  // trajectory parameters
  double radius, height, dtheta, phase, wob_a, wob_b;

  // ATAN/FOV camera (jni/ATANCamera.cc:133-164, ATANCamera.h:136-150)
  void unproject(double ix, double iy, double out[2]) const {
    const double dx = (ix - center[0]) * inv_focal[0], dy = (iy - center[1]) * inv_focal[1];
    const double dr = std::sqrt(dx * dx + dy * dy);
    const double r = ww == 0.0 ? dr : std::tan(dr * ww) * one_over_2tan;
    const double f = dr > 0.01 ? r / dr : 1.0;
    out[0] = dx * f; out[1] = dy * f;
  }
  void project(double cx, double cy, double im[2]) const {
    const double r = std::sqrt(cx * cx + cy * cy);
    const double fac = (r < 0.001 || ww == 0.0) ? 1.0 : winv * std::atan(r * two_tan) / r;
    im[0] = center[0] + focal[0] * (cx * fac); im[1] = center[1] + focal[1] * (cy * fac);
  }
};

static void pose_at(const vslam_feeder* f, double t, double P[12]) {
  // camera centre on a circle above the plane, optical axis along +Z (plane at depth ~height), small wobble
  const double th = f->phase + f->dtheta * t;
  const double C[3] = {f->radius * std::cos(th), f->radius * std::sin(th), -(f->height + 0.04 * std::sin(0.013 * t + f->wob_a))};
  const double rx = 0.035 * std::sin(0.021 * t + f->wob_a), ry = 0.03 * std::cos(0.017 * t + f->wob_b), rz = 0.05 * std::sin(0.009 * t + f->wob_b);
  const double cx = std::cos(rx), sx = std::sin(rx), cy = std::cos(ry), sy = std::sin(ry), cz = std::cos(rz), sz = std::sin(rz);
  // R = Rz * Ry * Rx
  const double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                       sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                       -sy, cy * sx, cy * cx};
  for (int i = 0; i < 9; i++) P[i] = R[i];
  for (int i = 0; i < 3; i++) P[9 + i] = -(R[i * 3 + 0] * C[0] + R[i * 3 + 1] * C[1] + R[i * 3 + 2] * C[2]);
}

static void render_one(const vslam_feeder* f, const double P[12], uint64_t noise_key, uint8_t* out, size_t stride) {
  // camera centre in world: C = -R^T t ; ray_world = R^T (x, y, 1) ; hit plane z = 0
  const double* R = P; const double* t = P + 9;
  const double C[3] = {-(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]), -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]), -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2])};
  for (int y = 0; y < f->h; y++) {
    uint8_t* row = out + (size_t)y * stride;
    for (int x = 0; x < f->w; x++) {
      const float* ry = &f->rays[2 * ((size_t)y * f->w + x)];
      const double cx = ry[0], cy = ry[1];
      const double dx = R[0] * cx + R[3] * cy + R[6], dy = R[1] * cx + R[4] * cy + R[7], dz = R[2] * cx + R[5] * cy + R[8];
      int v = 128;
      if (dz > 1e-6) {
        const double s = -C[2] / dz;
        const double u = (C[0] + s * dx + TEX_M / 2) * PPM - 0.5, vv = (C[1] + s * dy + TEX_M / 2) * PPM - 0.5;
        if (u >= 0 && vv >= 0 && u < TEX - 1 && vv < TEX - 1) {
          const int iu = (int)u, iv = (int)vv;
          const double fu = u - iu, fv = vv - iv;
          const uint8_t* p = &f->tex[(size_t)iv * TEX + iu];
          v = (int)((1 - fv) * ((1 - fu) * p[0] + fu * p[1]) + fv * ((1 - fu) * p[TEX] + fu * p[TEX + 1]) + 0.5);
        }
      }
      if (f->noise) v += (int)(hash32(noise_key * 0x100000001B3ull + (uint64_t)y * f->w + x) % (2 * f->noise + 1)) - f->noise;
      row[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
}

extern "C" int vslam_feeder_create(int width, int height, const double cam[5], uint64_t seed, int noise_amplitude, vslam_feeder** out) {
  if (!out || !cam || width < 48 || height < 48) return -1;
  vslam_feeder* f = new vslam_feeder;
  f->w = width; f->h = height; f->noise = noise_amplitude; f->seed = seed;
  f->focal[0] = width * cam[0]; f->focal[1] = height * cam[1];
  f->center[0] = width * cam[2] - 0.5; f->center[1] = height * cam[3] - 0.5;
  f->inv_focal[0] = 1.0 / f->focal[0]; f->inv_focal[1] = 1.0 / f->focal[1];
  f->ww = cam[4];
  if (f->ww != 0.0) { f->two_tan = 2.0 * std::tan(f->ww / 2.0); f->one_over_2tan = 1.0 / f->two_tan; f->winv = 1.0 / f->ww; }
  else { f->two_tan = 0; f->one_over_2tan = 0; f->winv = 0; }
  Rng rng(seed);
  // texture: mid-grey value noise + rectangles (corners of rectangles are the FAST features)
  f->tex.assign((size_t)TEX * TEX, 0);
  {
    const int G = 64;  // coarse lattice for value noise
    std::vector<float> lat((size_t)(G + 1) * (G + 1));
    for (auto& v : lat) v = (float)(rng.uni() * 2 - 1);
    for (int y = 0; y < TEX; y++)
      for (int x = 0; x < TEX; x++) {
        const double gx = (double)x * G / TEX, gy = (double)y * G / TEX;
        const int ix = (int)gx, iy = (int)gy;
        const double fx = gx - ix, fy = gy - iy;
        const double n = (1 - fy) * ((1 - fx) * lat[iy * (G + 1) + ix] + fx * lat[iy * (G + 1) + ix + 1]) +
                         fy * ((1 - fx) * lat[(iy + 1) * (G + 1) + ix] + fx * lat[(iy + 1) * (G + 1) + ix + 1]);
        f->tex[(size_t)y * TEX + x] = (uint8_t)(128 + 18 * n);
      }
    int nrect = 1000;   // ~1000 FAST-10 corners per 640x480 frame at level 0 (BASELINE.json configs[1])
    if (const char* e = getenv("VSLAM_FEEDER_NRECT")) nrect = atoi(e);
    for (int i = 0; i < nrect; i++) {
      const int rw = rng.range(20, 160), rh = rng.range(20, 160);
      const int x0 = rng.range(0, TEX - rw), y0 = rng.range(0, TEX - rh);
      const int g = rng.range(20, 236);
      for (int y = y0; y < y0 + rh; y++) memset(&f->tex[(size_t)y * TEX + x0], g, rw);
    }
  }
  f->rays.resize((size_t)2 * width * height);
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      double o[2]; f->unproject(x, y, o);
      f->rays[2 * ((size_t)y * width + x)] = (float)o[0]; f->rays[2 * ((size_t)y * width + x) + 1] = (float)o[1];
    }
  f->radius = 0.30 + 0.1 * rng.uni();
  f->height = 1.0 + 0.1 * rng.uni();
  f->dtheta = 0.0095 * (0.9 + 0.2 * rng.uni());
  f->phase = 6.283185307179586 * rng.uni();
  f->wob_a = 6.283185307179586 * rng.uni(); f->wob_b = 6.283185307179586 * rng.uni();
  *out = f;
  return 0;
}

extern "C" int vslam_feeder_destroy(vslam_feeder* f) { delete f; return 0; }

extern "C" int vslam_feeder_pose(const vslam_feeder* f, double t, double pose12[12]) { if (!f) return -1; pose_at(f, t, pose12); return 0; }

extern "C" int vslam_feeder_render_pose(const vslam_feeder* f, const double pose12[12], uint64_t noise_key, uint8_t* frame, size_t stride) {
  if (!f || !frame) return -1;
  render_one(f, pose12, noise_key ^ (f->seed << 20), frame, stride);
  return 0;
}

extern "C" int vslam_feeder_render(const vslam_feeder* f, int first, int count, uint8_t* frames, size_t stride, size_t frame_stride, int n_threads) {
  if (!f || !frames || count < 0) return -1;
  if (n_threads < 1) n_threads = 1;
  auto work = [&](int tid) {
    for (int i = tid; i < count; i += n_threads) {
      double P[12]; pose_at(f, first + i, P);
      render_one(f, P, (uint64_t)(first + i + 100000) ^ (f->seed << 20), frames + (size_t)i * frame_stride, stride);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_threads; t++) th.emplace_back(work, t);
  work(0);
  for (auto& t : th) t.join();
  return 0;
}

static void xform(const double P[12], const double X[3], double out[3]) {
  for (int i = 0; i < 3; i++) out[i] = P[9 + i] + P[i * 3] * X[0] + P[i * 3 + 1] * X[1] + P[i * 3 + 2] * X[2];
}
static void normalize(double v[3]) { const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= n; v[1] /= n; v[2] /= n; }

extern "C" int vslam_feeder_make_point(const vslam_feeder* f, const double P[12], int level, int cx, int cy, double pos[3],
                                       double pix_right[3], double pix_down[3]) {
  if (!f) return -1;
  const int scale = 1 << level;
  const double root[2] = {(cx + 0.5) * scale - 0.5, (cy + 0.5) * scale - 0.5};   // LevelZeroPos, jni/LevelHelpers.h:23-25
  double c2[2]; f->unproject(root[0], root[1], c2);
  // world point: ray through the pixel hits plane z = 0
  const double* R = P; const double* t = P + 9;
  const double C[3] = {-(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]), -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]), -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2])};
  const double d[3] = {R[0] * c2[0] + R[3] * c2[1] + R[6], R[1] * c2[0] + R[4] * c2[1] + R[7], R[2] * c2[0] + R[5] * c2[1] + R[8]};
  if (d[2] <= 1e-9) return -2;
  const double s = -C[2] / d[2];
  pos[0] = C[0] + s * d[0]; pos[1] = C[1] + s * d[1]; pos[2] = 0.0;
  // jni/MapMaker.cc:652-684: unit rays to the centre, one level-pixel right, one level-pixel down; normal (0,0,-1)
  double cen[3] = {c2[0], c2[1], 1.0}, rgt[3], dwn[3];
  double a[2];
  f->unproject(root[0] + scale, root[1], a); rgt[0] = a[0]; rgt[1] = a[1]; rgt[2] = 1.0;
  f->unproject(root[0], root[1] + scale, a); dwn[0] = a[0]; dwn[1] = a[1]; dwn[2] = 1.0;
  normalize(cen); normalize(dwn); normalize(rgt);
  const double nrm[3] = {0, 0, -1};
  // MapPoint::RefreshPixelVectors, jni/MapPoint.cc:4-29
  double pc[3]; xform(P, pos, pc);
  const double camH = std::fabs(pc[0] * nrm[0] + pc[1] * nrm[1] + pc[2] * nrm[2]);
  const double rate = std::fabs(cen[2]), rrate = std::fabs(rgt[2]), drate = std::fabs(dwn[2]);
  double cop[3], rop[3], dop[3];
  for (int i = 0; i < 3; i++) { cop[i] = cen[i] * camH / rate; rop[i] = rgt[i] * camH / rrate; dop[i] = dwn[i] * camH / drate; }
  const double dr[3] = {rop[0] - cop[0], rop[1] - cop[1], rop[2] - cop[2]}, dd[3] = {dop[0] - cop[0], dop[1] - cop[1], dop[2] - cop[2]};
  for (int i = 0; i < 3; i++) {  // R^T *
    pix_right[i] = R[0 * 3 + i] * dr[0] + R[1 * 3 + i] * dr[1] + R[2 * 3 + i] * dr[2];
    pix_down[i] = R[0 * 3 + i] * dd[0] + R[1 * 3 + i] * dd[1] + R[2 * 3 + i] * dd[2];
  }
  return 0;
}

extern "C" int vslam_feeder_project(const vslam_feeder* f, const double P[12], const double pos[3], int border, double im[2], double* depth) {
  if (!f) return -1;
  double pc[3]; xform(P, pos, pc);
  if (depth) *depth = pc[2];
  if (pc[2] < 0.001) return 0;
  f->project(pc[0] / pc[2], pc[1] / pc[2], im);
  return (im[0] >= border && im[1] >= border && im[0] < f->w - border && im[1] < f->h - border) ? 1 : 0;
}
