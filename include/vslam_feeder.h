/*
 * vslam_feeder.h -- synthetic frame feeder: the stand-in for the Android camera / Java plumbing of the
 * reference (src/vision/ar/monoslam/MainActivity.java:113-136 -> jni/jni_part.cpp:132-145), which is out of
 * scope.  Host C++ only (no GPU work); part of libvslam_hip.so so tests and bench.py share one scene.
 *
 * Scene: a textured plane z = 0 (seeded random rectangles + low-amplitude value noise) seen by the
 * reference's ATAN/FOV camera (jni/ATANCamera.cc:20-24) on a smooth seeded trajectory.  Because the map
 * bootstrap (InitFromStereo / HomographyInit) is out of scope, the feeder also provides ground-truth map
 * points: a corner of a source keyframe is back-projected onto the plane and given the patch vectors of
 * MapMaker::AddPointEpipolar (jni/MapMaker.cc:652-684) + MapPoint::RefreshPixelVectors (jni/MapPoint.cc:4-29).
 */
#ifndef VSLAM_FEEDER_H
#define VSLAM_FEEDER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vslam_feeder vslam_feeder;

int vslam_feeder_create(int width, int height, const double cam[5], uint64_t seed, int noise_amplitude,
                        vslam_feeder** out);
int vslam_feeder_destroy(vslam_feeder* f);
/* ground-truth camera-from-world pose (R row-major 9 + t 3) at frame index t (may be negative) */
int vslam_feeder_pose(const vslam_feeder* f, double t, double pose12[12]);
/* render `count` consecutive frames first..first+count-1 into frames[count][height][stride] with n_threads */
int vslam_feeder_render(const vslam_feeder* f, int first, int count, uint8_t* frames, size_t stride,
                        size_t frame_stride, int n_threads);
int vslam_feeder_render_pose(const vslam_feeder* f, const double pose12[12], uint64_t noise_key, uint8_t* frame,
                             size_t stride);
/* ground-truth map point for corner (cx, cy) of pyramid level `level` of a keyframe at pose12 */
int vslam_feeder_make_point(const vslam_feeder* f, const double pose12[12], int level, int cx, int cy,
                            double pos[3], double pix_right[3], double pix_down[3]);
/* project a world point; returns 1 if inside the image with `border` pixels to spare, else 0 */
int vslam_feeder_project(const vslam_feeder* f, const double pose12[12], const double pos[3], int border,
                         double im[2], double* depth);

#ifdef __cplusplus
}
#endif
#endif
