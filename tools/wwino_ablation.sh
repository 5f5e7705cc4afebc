#!/bin/bash
# A/B builds of wgrad_wino.hip (GPU box, from the repo root; needs build/obj/*.o): EXTRA flags per variant, timed with
# tools/bench_kernels.py (fp32, BASELINE configs[1] shapes).  usage: tools/wwino_ablation.sh "<name>=<flags>" ...
set -e
R=$PWD
mkdir -p build/exp
HIPCC=/opt/rocm/bin/hipcc
OBJS=$(ls build/obj/*.o | grep -v wgrad_wino.o)
for spec in "BASE=" "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -I include -I unet_nested4tiny_objects_keypoints_amd/csrc -c unet_nested4tiny_objects_keypoints_amd/csrc/wgrad_wino.hip -o build/exp/wgrad_wino_$name.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_ww_$name.so $OBJS build/exp/wgrad_wino_$name.o
done
for spec in "BASE=" "$@" "BASE="; do
  name=${spec%%=*}
  echo "== $name"
  for L in ${LAYERS:-X01.conv2 X03.conv1 enc2.conv1 X21.conv1}; do
    UNETPP_LIB=$R/build/exp/libunetpp_ww_$name.so REPS=20 timeout -k 10 120 python tools/bench_kernels.py $L 2>&1 | grep "^$L" | cut -c1-50,95-112
  done
done
