// MiniPatch (jni/MiniPatch.cc): 9x9 raw-SSD search at the FAST corners of the current frame's level 0 inside a +-range
// box, one wavefront per trail, batched over all trails of a stream.  Used by the reference's trail tracking
// (jni/Tracker.cc:294-346, forward + backward "married match"); the trail state machine itself is bootstrap code and
// stays on the host (out of scope), these are the two data-parallel primitives it calls.
#include "vslam_internal.h"

#define MP_HALF 4   // MiniPatch::mnHalfPatchSize (jni/MiniPatch.cc:86)
#define MP_SIDE 9
#define MP_PIX 81

// SampleFromImage (:71-83): one lane per pixel
__global__ __launch_bounds__(128) void k_minipatch_sample(const uint8_t* img, int pitch, int w, int h, const int* pos, int n,
                                                          uint8_t* patches, int* ok) {
  const int t = blockIdx.x;
  if (t >= n) return;
  const int x = pos[2 * t], y = pos[2 * t + 1];
  const bool inside = x >= MP_HALF && y >= MP_HALF && x < w - MP_HALF && y < h - MP_HALF;   // in_image_with_border (assert :73)
  if (threadIdx.x == 0) ok[t] = inside;
  if (!inside || threadIdx.x >= MP_PIX) return;
  const int r = threadIdx.x / MP_SIDE, c = threadIdx.x - r * MP_SIDE;
  patches[(size_t)t * MP_PIX + threadIdx.x] = img[(size_t)(y - MP_HALF + r) * pitch + (x - MP_HALF + c)];
}

// FindPatch (:35-68) + SSDAtPoint (:6-30)
__global__ __launch_bounds__(64) void k_minipatch_find(const uint8_t* img, int pitch, int w, int h, const uint32_t* corners,
                                                       const int* rowlut, int ncorners, const uint8_t* patches, int* pos, int n,
                                                       int range, int max_ssd, int* found) {
  const int t = blockIdx.x, lane = threadIdx.x;
  if (t >= n) return;
  __shared__ uint8_t tmpl[MP_PIX + 3];
  __shared__ int cand[64];
  for (int k = lane; k < MP_PIX; k += 64) tmpl[k] = patches[(size_t)t * MP_PIX + k];
  const int px = pos[2 * t], py = pos[2 * t + 1];
  const int L = px - range, R = px + range, T = py - range, B = py + range;
  int best = max_ssd + 1, bestIdx = 0x7fffffff;
  // corners with T <= y <= B: [lut[T], lut[B+1]) -- the reference scans linearly from the first corner with y >= T
  const int y0 = T < 0 ? 0 : T, y1 = B + 1;
  const int i0 = y0 >= h ? ncorners : rowlut[y0];
  const int i1 = y1 >= h ? ncorners : (y1 < 0 ? 0 : rowlut[y1]);
  const int grp = lane >> 3, sub = lane & 7;
  __syncthreads();
  for (int base = i0; base < i1; base += 64) {
    const int ci = base + lane;
    bool ok = false;
    if (ci < i1) { const int cx = corners[ci] & 0xFFFF; ok = !(cx < L || cx > R); }
    const unsigned long long bm = __ballot(ok);
    const int nc = __popcll(bm);
    if (ok) cand[__popcll(bm & ((1ull << lane) - 1ull))] = ci;
    __syncthreads();
    for (int c0 = 0; c0 < nc; c0 += 8) {
      const int k = c0 + grp;
      int ssd = 0x7fffffff, cidx = 0x7fffffff;
      if (k < nc) {
        cidx = cand[k];
        const uint32_t c = corners[cidx];
        const int cx = c & 0xFFFF, cy = c >> 16;
        const bool inside = cx >= MP_HALF && cy >= MP_HALF && cx < w - MP_HALF && cy < h - MP_HALF;
        int s = 0;
        if (inside) {
          const uint8_t* ib = img + (size_t)(cy - MP_HALF) * pitch + (cx - MP_HALF);
          for (int q = sub; q < MP_PIX; q += 8) { const int r = q / MP_SIDE, cc = q - r * MP_SIDE; const int d = (int)ib[r * pitch + cc] - (int)tmpl[q]; s += d * d; }
        }
        for (int d = 1; d < 8; d <<= 1) s += __shfl_xor(s, d);
        ssd = inside ? s : max_ssd + 1;
      }
      for (int d = 8; d < 64; d <<= 1) {
        const int os = __shfl_xor(ssd, d), oi = __shfl_xor(cidx, d);
        if (os < ssd || (os == ssd && oi < cidx)) { ssd = os; cidx = oi; }
      }
      if (ssd < best) { best = ssd; bestIdx = cidx; }   // first strict minimum in raster order (:59)
    }
    __syncthreads();
  }
  if (lane == 0) {
    if (best < max_ssd) { const uint32_t c = corners[bestIdx]; pos[2 * t] = c & 0xFFFF; pos[2 * t + 1] = c >> 16; found[t] = 1; }
    else found[t] = 0;
  }
}

static int mp_buffers(vslam_system* sys, int n, int** d_pos, uint8_t** d_patch, int** d_flag) {
  void *a = nullptr, *b = nullptr, *c = nullptr;
  HIPCHK(hipMalloc(&a, sizeof(int) * 2 * (size_t)n + 64));
  HIPCHK(hipMalloc(&b, (size_t)MP_PIX * n + 64));
  HIPCHK(hipMalloc(&c, sizeof(int) * (size_t)n + 64));
  *d_pos = (int*)a; *d_patch = (uint8_t*)b; *d_flag = (int*)c;
  return VSLAM_OK;
}

extern "C" int vslam_minipatch_sample(vslam_system* sys, int stream, int n, const int* pos_xy, uint8_t* patches, int* ok) {
  if (!sys || stream < 0 || stream >= sys->S || n < 0 || (n && (!pos_xy || !patches || !ok))) { vslam_set_error("minipatch_sample: bad argument"); return VSLAM_E_INVALID; }
  if (!sys->have_frame) { vslam_set_error("minipatch_sample: no current frame"); return VSLAM_E_STATE; }
  if (n == 0) return VSLAM_OK;
  int* d_pos; uint8_t* d_patch; int* d_ok;
  int r = mp_buffers(sys, n, &d_pos, &d_patch, &d_ok); if (r) return r;
  HIPCHK(hipMemcpyAsync(d_pos, pos_xy, sizeof(int) * 2 * n, hipMemcpyHostToDevice, sys->stream));
  hipLaunchKernelGGL(k_minipatch_sample, dim3(n), dim3(128), 0, sys->stream, sys->fr.img[0] + (size_t)stream * sys->fr.img_sstride[0],
                     sys->fr.img_pitch[0], sys->geom[0].w, sys->geom[0].h, d_pos, n, d_patch, d_ok);
  HIPCHK(hipMemcpyAsync(patches, d_patch, (size_t)MP_PIX * n, hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipMemcpyAsync(ok, d_ok, sizeof(int) * n, hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  (void)hipFree(d_pos); (void)hipFree(d_patch); (void)hipFree(d_ok);
  return VSLAM_OK;
}

extern "C" int vslam_minipatch_find(vslam_system* sys, int stream, int n, const uint8_t* patches, int* pos_xy, int range, int max_ssd,
                                    int* found) {
  if (!sys || stream < 0 || stream >= sys->S || n < 0 || range < 0 || (n && (!pos_xy || !patches || !found))) { vslam_set_error("minipatch_find: bad argument"); return VSLAM_E_INVALID; }
  if (!sys->have_frame) { vslam_set_error("minipatch_find: no current frame"); return VSLAM_E_STATE; }
  if (n == 0) return VSLAM_OK;
  int* d_pos; uint8_t* d_patch; int* d_found;
  int r = mp_buffers(sys, n, &d_pos, &d_patch, &d_found); if (r) return r;
  int nc = 0;
  HIPCHK(hipMemcpyAsync(&nc, sys->fr.ncorners + stream * NLEV, sizeof(int), hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipMemcpyAsync(d_pos, pos_xy, sizeof(int) * 2 * n, hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipMemcpyAsync(d_patch, patches, (size_t)MP_PIX * n, hipMemcpyHostToDevice, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  const LevelGeom& g = sys->geom[0];
  hipLaunchKernelGGL(k_minipatch_find, dim3(n), dim3(64), 0, sys->stream, sys->fr.img[0] + (size_t)stream * sys->fr.img_sstride[0],
                     sys->fr.img_pitch[0], g.w, g.h, sys->fr.corners[0] + (size_t)stream * g.cap, sys->fr.rowlut[0] + (size_t)stream * (g.h + 1), nc,
                     d_patch, d_pos, n, range, max_ssd, d_found);
  HIPCHK(hipMemcpyAsync(pos_xy, d_pos, sizeof(int) * 2 * n, hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipMemcpyAsync(found, d_found, sizeof(int) * n, hipMemcpyDeviceToHost, sys->stream));
  HIPCHK(hipStreamSynchronize(sys->stream));
  (void)hipFree(d_pos); (void)hipFree(d_patch); (void)hipFree(d_found);
  return VSLAM_OK;
}
