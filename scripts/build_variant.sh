#!/bin/bash
# A build of libtbcheck.so whose several-histories-per-wavefront kernel (wgl_narrow.hip) is compiled with other build-time forms:
#   scripts/build_variant.sh eb1 -DTBC_NARROW_EB=1      ->  jepsen-tigerbeetle_amd/csrc/variants/libtbcheck_eb1.so
#   SRC=pack_one scripts/build_variant.sh packprof -DTBC_PACK_PROF=1     (another source file than wgl_narrow.hip)
# Every other object is the default build's.  Loaded with TBC_LIB_PATH=... (jepsen-tigerbeetle_amd/_native.py) for A/B runs on the GPU box.
set -e
cd "$(dirname "$0")/../jepsen-tigerbeetle_amd/csrc"
tag=$1; shift
SRC=${SRC:-wgl_narrow}
make -s -j8 libtbcheck.so
mkdir -p variants build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter "$@" -c $SRC.hip -o variants/${SRC}_$tag.o
objs=$(ls build/*.o | grep -v "build/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libtbcheck_$tag.so $objs variants/${SRC}_$tag.o -ldl
echo variants/libtbcheck_$tag.so
