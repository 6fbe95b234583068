#!/bin/bash
# Diagnostic build of the library with extra compiler flags into tools/probes/_build/<name>.so (never the product library):
#   bash tools/build_variant.sh noquick -DVSLAM_FE_SKIP_QUICK
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
T=$(mktemp -d)
mkdir -p $T/visualslam_android_amd && cp -r $ROOT/visualslam_android_amd/csrc $T/visualslam_android_amd/ && cp -r $ROOT/include $T/
rm -rf $T/visualslam_android_amd/csrc/_build
make -s -j4 -C $T/visualslam_android_amd/csrc HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off $*"
mkdir -p $ROOT/tools/probes/_build
cp $T/visualslam_android_amd/libvslam_hip.so $ROOT/tools/probes/_build/$NAME.so
rm -rf $T
echo built tools/probes/_build/$NAME.so
