/*
 * ptam_oracle.h -- C API of the CPU ORACLE for the PTAM tracking + local-BA hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (visualslam_android_amd/csrc, libvslam_hip.so) never links or calls anything here.
 *
 * Every function is a scalar, double-precision restatement of one reference function and
 * cites the reference file:line it follows (paths relative to the reference checkout,
 * ahcorde/visualSLAM_Android).
 *
 * Parity status (see DESIGN.md "Oracle pinning"):
 *   - FAST-10 segment test: pinned -- exhaustive equivalence with the reference's decision
 *     tree (jni/vision/cvfast.cpp:6124-9235) over all 3^16 ring states, by a tree-walk of the
 *     reference text (oracle/pin_fast_tree.py; not a build of the reference).
 *   - Tukey/Cauchy/Huber M-estimators: pinned against jni/MEstimator.h compiled verbatim
 *     (oracle/Makefile -> oracle/_ref/libref_mestimator.so).
 *   - Everything else: the reference has no tests, fixtures or golden vectors and cannot be
 *     built here (needs OpenCV 2.4 + Eigen 3) => "parity unpinned" against the reference,
 *     pinned only by analytic known-answer tests (tests/test_oracle_*.py).
 */
#ifndef PTAM_ORACLE_H
#define PTAM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LEVELS 4

/* reference-quirk flags */
#define ORC_Q_CAM_INT_RADIUS         1  /* jni/ATANCamera.cc:70-82 */
#define ORC_Q_POSE_INT_RESIDUAL      2  /* jni/Tracker.cc:766-767  */
#define ORC_Q_NONMAX_RIGHT_NEIGHBOUR 4  /* jni/vision/cvfast.cpp:9282-9285 */

/* ---- frame front-end ------------------------------------------------------------------ */

/* cv::resize(prev, lev, size/2) call at jni/KeyFrame.cc:20-23, restated as the exact 2:1
 * area filter (a+b+c+d+2)>>2 (third-party arithmetic: parity unpinned). dst is (w/2)x(h/2). */
void orc_halfsample(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);

/* cvCornerFast_10, jni/vision/cvfast.cpp:6088-9241.  corners packed x | y<<16, raster order.
 * Returns the total number of corners (may exceed cap; only the first cap are written). */
int orc_fast10(const uint8_t* img, int w, int h, int stride, int threshold,
               uint32_t* corners, int cap);

/* row LUT, jni/KeyFrame.cc:43-49: lut[y] = first index with corner.y >= y, y in [0,h). */
void orc_row_lut(const uint32_t* corners, int n, int h, int* lut);

/* compute_fast_score_old, jni/vision/cvfast.cpp:9337-9393 */
void orc_fast_score(const uint8_t* img, int w, int h, int stride, const uint32_t* corners, int n,
                    int barrier, int* scores);

/* nonmax_suppression, jni/vision/cvfast.cpp:9243-9335. quirk!=0 reproduces the "check right"
 * bug (corners[i-1] row test; skipped for i==0 where the reference reads out of bounds). */
int orc_nonmax(const uint32_t* corners, const int* scores, int n, int quirk, uint32_t* out);

/* FindShiTomasiScoreAtPoint, jni/vision/ImageHandler.cpp:124-155 */
double orc_shi_tomasi(const uint8_t* img, int stride, int nsize, int px, int py);

/* KeyFrame::MakeKeyFrame_Lite, jni/KeyFrame.cc:5-51: 4-level pyramid + FAST-10 + row LUT.
 * lvl_img[l] must hold (w>>l)*(h>>l) bytes (tight pitch); corners[l] cap entries;
 * lut[l] (h>>l) ints.  Returns 0. */
int orc_make_keyframe_lite(const uint8_t* gray, int w, int h, int stride, const int thr[ORC_LEVELS],
                           uint8_t* const lvl_img[ORC_LEVELS], uint32_t* const corners[ORC_LEVELS],
                           int cap, int ncorners[ORC_LEVELS], int* const lut[ORC_LEVELS]);

#ifdef __cplusplus
}
#endif
#endif
