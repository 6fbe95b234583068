"""CPU test: libvslam_hip.so loads and exports every symbol include/vslam_c.h declares."""
import ctypes
import os
import re

from visualslam_android_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="vslam_c.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vslam_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert "vslam_create" in syms and "vslam_make_keyframe_lite" in syms
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libvslam_hip.so does not export %s" % s
    # and the Python mirror binds exactly that set
    assert sorted(capi.SYMBOLS) == syms
    from visualslam_android_amd import feeder
    fsyms = declared_symbols("vslam_feeder.h")
    for s in fsyms:
        assert hasattr(lib, s), "libvslam_hip.so does not export %s" % s
    assert sorted(feeder.FEEDER_SYMBOLS) == fsyms


def test_params_struct_defaults():
    p = capi.default_params(640, 480, 3)
    assert (p.width, p.height, p.n_streams) == (640, 480, 3)
    assert list(p.fast_threshold) == [10, 15, 15, 10]          # jni/KeyFrame.cc:32-39
    assert p.patch_size == 11 and p.max_patches_per_frame == 1000
    assert abs(p.cam[0] - 0.841906) < 1e-12 and abs(p.cam[4] + 0.0133843) < 1e-12


def test_jni_alias_translation_unit_exports_the_reference_symbols(tmp_path):
    """csrc/jni_alias.cpp (jni/jni_part.cpp:84-145 as aliases of the C ABI) is empty without -DHAVE_JNI (no JDK in the image); with
    it -- and the self-test typedefs standing in for jni.h -- it compiles and defines exactly the reference's five JNI symbols."""
    import subprocess
    src = os.path.join(ROOT, "visualslam_android_amd", "csrc", "jni_alias.cpp")
    obj = str(tmp_path / "jni_alias.o")
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-DHAVE_JNI", "-DVSLAM_JNI_SELFTEST", "-c", src, "-o", obj])
    out = subprocess.check_output(["nm", "-g", "--defined-only", obj], text=True)
    syms = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert syms == sorted(["Java_vision_ar_monoslam_SystemPTAM_native_1createTest", "Java_vision_ar_monoslam_SystemPTAM_native_1disposeTest",
                           "Java_vision_ar_monoslam_SystemPTAM_native_1touchScreen", "Java_vision_ar_monoslam_SystemPTAM_native_1update",
                           "Java_vision_ar_monoslam_MainActivity_FindFeatures"])
    empty = str(tmp_path / "jni_alias_empty.o")
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-c", src, "-o", empty])
    assert "Java_" not in subprocess.check_output(["nm", "-g", empty], text=True)
