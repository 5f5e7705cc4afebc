#!/bin/bash
# Ablation of wgrad_bf16_quad_kernel (GPU box, from the repo root; needs build/obj/*.o of a normal build): experiment
# builds of wgrad_bf16.hip with one phase of the tile loop removed (wrong results, timing only), linked into build/exp/,
# timed with tools/bench_kernels.py (wgrad + finish launch) on wide shapes of BASELINE configs[3] (512 x 512, batch 8).
set -e
R=$PWD
mkdir -p build/exp
HIPCC=/opt/rocm/bin/hipcc
OBJS=$(ls build/obj/*.o | grep -v wgrad_bf16.o)
V=${VARIANTS:-BASE NO_STORE_NO_LOAD NO_COMPUTE_NO_STORE NO_SLAB NO_STORE_NO_LOAD_NO_BARRIER}
for v in $V; do
  D=""
  for part in COMPUTE STORE LOAD SLAB FENCE BARRIER; do
    [[ $v == *NO_$part* ]] && D="$D -DUNETPP_WQ_EXP_NO_$part"
  done
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I unet_nested4tiny_objects_keypoints_amd/csrc $D -c unet_nested4tiny_objects_keypoints_amd/csrc/wgrad_bf16.hip -o build/exp/wgrad_bf16_$v.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_wq_$v.so $OBJS build/exp/wgrad_bf16_$v.o
done
export REPS=20 DTYPE=bf16 B=8 SIZE=512
for v in $V BASE; do
  echo "== $v"
  for L in ${LAYERS:-enc1.conv2 X12.conv1 enc3.conv2}; do
    UNETPP_LIB=$R/build/exp/libunetpp_wq_$v.so timeout -k 10 120 python tools/bench_kernels.py $L 2>&1 | grep "^$L" | cut -c1-50,95-112
  done
done
