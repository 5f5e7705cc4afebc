#!/bin/bash
# Ablation of wgrad_bf16_quad_kernel (GPU box, from the repo root; needs build/obj/*.o of a normal build): experiment
# builds of wgrad_bf16.hip with one phase of the tile loop removed (wrong results, timing only), linked into build/exp/,
# timed with tools/bench_kernels.py (wgrad + finish launch) on wide shapes of BASELINE configs[3] (512 x 512, batch 8).
set -e
R=$PWD
mkdir -p build/exp
HIPCC=/opt/rocm/bin/hipcc
OBJS=$(ls build/obj/*.o | grep -v wgrad_bf16.o)
V=${VARIANTS:-BASE INTERLEAVE NO_STORE_NO_LOAD NO_SLAB CLOCK}
for v in ${V/CLOCK/}; do
  D=""
  for part in COMPUTE STORE LOAD SLAB FENCE BARRIER; do
    [[ $v == *NO_$part* ]] && D="$D -DUNETPP_WQ_EXP_NO_$part"
  done
  [[ $v == *INTERLEAVE* ]] && D="$D -DUNETPP_WQ_EXP_INTERLEAVE"
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I unet_nested4tiny_objects_keypoints_amd/csrc $D -c unet_nested4tiny_objects_keypoints_amd/csrc/wgrad_bf16.hip -o build/exp/wgrad_bf16_$v.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_wq_$v.so $OBJS build/exp/wgrad_bf16_$v.o
done
export REPS=20 DTYPE=bf16 B=8 SIZE=512
if [[ " $V " == *" CLOCK "* ]]; then
  V=${V/CLOCK/}
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I unet_nested4tiny_objects_keypoints_amd/csrc -DUNETPP_WQ_EXP_CLOCK -c unet_nested4tiny_objects_keypoints_amd/csrc/wgrad_bf16.hip -o build/exp/wgrad_bf16_CLOCK.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_wq_CLOCK.so $OBJS build/exp/wgrad_bf16_CLOCK.o
  echo "== CLOCK"
  UNETPP_LIB=$R/build/exp/libunetpp_wq_CLOCK.so timeout -k 10 120 python tools/wgrad_quad_clock.py 2>&1 | grep operands
fi
for v in $V BASE; do
  echo "== $v"
  for L in ${LAYERS:-enc1.conv2 X12.conv1 enc3.conv2}; do
    UNETPP_LIB=$R/build/exp/libunetpp_wq_$v.so timeout -k 10 120 python tools/bench_kernels.py $L 2>&1 | grep "^$L" | cut -c1-50,95-112
  done
done
