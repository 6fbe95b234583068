// The reference's JNI surface (jni/jni_part.cpp:84-145) as thin aliases of the C ABI, compiled only with -DHAVE_JNI (the build
// image has no JDK: `jni.h` is absent, so the product library is built without this translation unit's body).
//   Java_vision_ar_monoslam_SystemPTAM_native_1createTest   -> vslam_create   (800x480, jni_part.cpp:41; one stream)
//   Java_vision_ar_monoslam_SystemPTAM_native_1disposeTest  -> vslam_destroy
//   Java_vision_ar_monoslam_SystemPTAM_native_1touchScreen  -> vslam_touch    (mbUserPressedSpacebar, jni_part.cpp:49-51)
//   Java_vision_ar_monoslam_SystemPTAM_native_1update       -> vslam_update   (gray cv::Mat* smuggled as jlong, :132-145)
//   Java_vision_ar_monoslam_MainActivity_FindFeatures       -> empty, as in the reference (:84-103, body commented out)
// The Java side (src/vision/ar/monoslam/SystemPTAM.java:8-35) is unchanged.  cv::Mat is only read through data / step, so the
// stand-in of include/vslam/ptam.h is layout-INcompatible with OpenCV's: a build against the Android OpenCV SDK defines
// VSLAM_HAVE_OPENCV and includes the real header.
#ifdef HAVE_JNI
#ifdef VSLAM_JNI_SELFTEST
// Compile/link check without a JDK (tests/test_capi_symbols.py): the five JNI types this file uses, nothing else.
typedef long long jlong; typedef int jint; typedef void* jobject; struct JNIEnv_; typedef JNIEnv_ JNIEnv;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#else
#include <jni.h>
#endif
#ifdef VSLAM_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#endif
#include "../../include/vslam/ptam.h"

extern "C" {

JNIEXPORT jlong JNICALL Java_vision_ar_monoslam_SystemPTAM_native_1createTest(JNIEnv*, jobject) {
  vslam_params p;
  if (vslam_default_params(&p, 800, 480, 1) != VSLAM_OK) return 0;      // jni/jni_part.cpp:41 constructs the tracker at 800 x 480
  vslam_system* sys = nullptr;
  return vslam_create(&p, &sys) == VSLAM_OK ? reinterpret_cast<jlong>(sys) : 0;
}

JNIEXPORT void JNICALL Java_vision_ar_monoslam_SystemPTAM_native_1disposeTest(JNIEnv*, jobject, jlong cptr) {
  vslam_destroy(reinterpret_cast<vslam_system*>(cptr));
}

JNIEXPORT void JNICALL Java_vision_ar_monoslam_SystemPTAM_native_1touchScreen(JNIEnv*, jobject, jlong cptr) {
  vslam_touch(reinterpret_cast<vslam_system*>(cptr));
}

JNIEXPORT jint JNICALL Java_vision_ar_monoslam_SystemPTAM_native_1update(JNIEnv*, jobject, jlong cptr, jlong addrGray, jlong /*addrRgba*/) {
  cv::Mat& gray = *reinterpret_cast<cv::Mat*>(addrGray);               // CV_8UC1, as MainActivity.onCameraFrame passes it
  vslam_update(reinterpret_cast<vslam_system*>(cptr), gray.data, gray.step, 0);   // Tracker::TrackFrame
  return 0;                                                            // the reference returns constant 0 (:144)
}

JNIEXPORT void JNICALL Java_vision_ar_monoslam_MainActivity_FindFeatures(JNIEnv*, jobject, jlong, jlong) {}

}  // extern "C"
#endif  // HAVE_JNI
