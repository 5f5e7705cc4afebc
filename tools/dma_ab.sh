#!/bin/bash
# A/B of gemm_bf16_dma.hip builds on ONE box (boxes differ by up to 30 % on store-heavy launches), alternating:
#   cur = the tree's kernel, alt = the file named by ALT (default: the committed one, `git show HEAD:...`) with ALT_FLAGS
set -e
R=$PWD
mkdir -p build/exp
HIPCC=/opt/rocm/bin/hipcc
OBJS=$(ls build/obj/*.o | grep -v gemm_bf16_dma.o)
CS=unet_nested4tiny_objects_keypoints_amd/csrc
if [ -n "$ALT" ]; then cp "$ALT" build/exp/gemm_bf16_dma_alt.hip; else git show HEAD:$CS/gemm_bf16_dma.hip > build/exp/gemm_bf16_dma_alt.hip; fi
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $CS -c $CS/gemm_bf16_dma.hip -o build/exp/ab_cur.o
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $CS $ALT_FLAGS -c build/exp/gemm_bf16_dma_alt.hip -o build/exp/ab_alt.o
for v in cur alt; do $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_ab_$v.so $OBJS build/exp/ab_$v.o; done
export REPS=20 DTYPE=bf16
for round in 1 2; do
  for v in alt cur; do
    echo "== $v (round $round)"
    UNETPP_LIB=$R/build/exp/libunetpp_ab_$v.so BASE=64 FULL=1 DEPTH=4 B=4 SIZE=384 timeout -k 10 120 python tools/bench_kernels.py 2>&1 | grep "TOTAL fwd\|TOTAL dgrad\|X04.conv1\|X13.conv1"
    UNETPP_LIB=$R/build/exp/libunetpp_ab_$v.so B=8 SIZE=512 timeout -k 10 120 python tools/bench_kernels.py 2>&1 | grep "TOTAL fwd\|TOTAL dgrad\|X03.conv1"
  done
done
