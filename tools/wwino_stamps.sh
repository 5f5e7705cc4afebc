#!/bin/bash
# Phase stamps of wgrad_wino_kernel (GPU box, from the repo root; needs build/obj/*.o of a normal build):
# builds wgrad_wino.hip with -DUNETPP_WWINO_STAMPS into build/stamps/ and runs tools/wino_stamps.py --wgrad on the
# given shapes ("cin cout hw" triples, default the level-0 and level-2 layers of BASELINE configs[1]).
set -e
R=$PWD
mkdir -p build/stamps
HIPCC=/opt/rocm/bin/hipcc
OBJS=$(ls build/obj/*.o | grep -v wgrad_wino.o)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DUNETPP_WWINO_STAMPS $EXTRA -I include -I unet_nested4tiny_objects_keypoints_amd/csrc -c unet_nested4tiny_objects_keypoints_amd/csrc/wgrad_wino.hip -o build/stamps/wgrad_wino.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o build/stamps/libunetpp_stamps.so $OBJS build/stamps/wgrad_wino.o
for shape in "${@:-32 32 256}"; do
  echo "== $shape"
  UNETPP_LIB=$R/build/stamps/libunetpp_stamps.so timeout -k 10 120 python tools/wino_stamps.py --wgrad $shape 2>&1 | grep -v amdgpu | tail -14
done
