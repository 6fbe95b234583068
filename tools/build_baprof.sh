#!/bin/bash
# Diagnostic build of libvslam_hip.so with clock64() phase stamps inside the persistent BA kernel (-DVSLAM_BA_PROF).
# Output: visualslam_android_amd/libvslam_hip_baprof.so (never the product library). Use with tests/tools: see DESIGN.md.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/visualslam_android_amd && cp -r $ROOT/visualslam_android_amd/csrc $T/visualslam_android_amd/ && cp -r $ROOT/include $T/
rm -rf $T/visualslam_android_amd/csrc/_build
make -s -j4 -C $T/visualslam_android_amd/csrc HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -DVSLAM_BA_PROF"
cp $T/visualslam_android_amd/libvslam_hip.so $ROOT/visualslam_android_amd/libvslam_hip_baprof.so
rm -rf $T
echo built $ROOT/visualslam_android_amd/libvslam_hip_baprof.so
