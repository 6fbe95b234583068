// Internal layout of a vslam_system: every buffer lives in HBM for the lifetime of the handle.
// gfx950 only.  Host-side bookkeeping + device pointers; kernels receive POD views.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/vslam_c.h"

#define NLEV VSLAM_LEVELS

void vslam_set_error(const char* fmt, ...);

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      vslam_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return VSLAM_E_HIP;                                                                    \
    }                                                                                        \
  } while (0)

// Geometry of one pyramid level (same for every stream).
struct LevelGeom {
  int w, h;        // level size: (W >> l, H >> l)            (jni/KeyFrame.cc:21)
  int pitch;       // bytes between rows of the internal level image (multiple of 64)
  int nchunk;      // ceil(w / 64): 64-bit corner-mask words per row
  int cap;         // corner capacity per stream
  int thr;         // FAST threshold                          (jni/KeyFrame.cc:32-39)
};

// Device view of the current frame of all streams (front-end outputs).
struct FrameDev {
  // level images: img[l] + s*img_sstride[l] + y*pitch.  Level 0 may point into caller memory.
  const uint8_t* img[NLEV];
  size_t img_sstride[NLEV];
  int img_pitch[NLEV];
  unsigned long long* cmask[NLEV];  // [S][h][nchunk] corner bit masks (bit i of word c = pixel 64c+i)
  int* rowcnt[NLEV];                // [S][h]
  int* rowlut[NLEV];                // [S][h+1]  rowlut[y] = first corner index with y' >= y; [h] = n
  uint32_t* corners[NLEV];          // [S][cap]  x | y<<16, raster order
  int* ncorners;                    // [S][NLEV]
  int* overflow;                    // [1] set when any corner list hit its capacity
  // non-max products (MakeKeyFrame_Rest)
  int* scores[NLEV];                // [S][cap]
  uint32_t* maxcorners[NLEV];       // [S][cap]
  int* nmax;                        // [S][NLEV]
};

struct vslam_system {
  vslam_params p;
  int S;
  hipStream_t stream;
  LevelGeom geom[NLEV];
  FrameDev fr;                 // device pointers (host copy of the view)
  uint8_t* d_lvl[NLEV];        // owned level images (level 0 = staging copy for host input)
  std::vector<void*> allocs;   // everything to hipFree
  bool have_frame;
};

// frontend.hip
int fe_make_keyframe_lite(vslam_system* sys, const uint8_t* gray, size_t row_stride, size_t stream_stride,
                          int on_device);
int fe_fast_nonmax(vslam_system* sys);
