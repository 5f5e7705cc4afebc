#!/bin/bash
# Ablation of gemm_bf16_dma_kernel (GPU box, from the repo root; needs build/obj/*.o of a normal build): experiment
# builds of gemm_bf16_dma.hip with one phase removed (wrong results, timing only), linked into build/exp/, timed with
# tools/bench_kernels.py on the level-0 shapes of BASELINE configs[3] (512 x 512, batch 8).
set -e
R=$PWD
mkdir -p build/exp
HIPCC=/opt/rocm/bin/hipcc
OBJS=$(ls build/obj/*.o | grep -v gemm_bf16_dma.o)
for v in ${VARIANTS:-BASE NO_MFMA NO_EPI NO_INDMA NO_MFMA_NO_EPI STORE_LINEAR}; do
  D=""
  case $v in
    NO_MFMA) D="-DUNETPP_DMA_EXP_NO_MFMA";;
    NO_EPI) D="-DUNETPP_DMA_EXP_NO_EPI";;
    NO_INDMA) D="-DUNETPP_DMA_EXP_NO_INDMA";;
    NO_MFMA_NO_EPI) D="-DUNETPP_DMA_EXP_NO_MFMA -DUNETPP_DMA_EXP_NO_EPI";;
    STORE_LINEAR) D="-DUNETPP_DMA_EXP_STORE_LINEAR";;
    STORE_LINEAR_NO_MFMA) D="-DUNETPP_DMA_EXP_STORE_LINEAR -DUNETPP_DMA_EXP_NO_MFMA";;
    NO_STORE) D="-DUNETPP_DMA_EXP_NO_STORE";;
    NO_WDMA) D="-DUNETPP_DMA_EXP_NO_WDMA";;
    HALF_WDMA) D="-DUNETPP_DMA_EXP_HALF_WDMA";;




  esac
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I unet_nested4tiny_objects_keypoints_amd/csrc $D -c unet_nested4tiny_objects_keypoints_amd/csrc/gemm_bf16_dma.hip -o build/exp/gemm_bf16_dma_$v.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_dma_$v.so $OBJS build/exp/gemm_bf16_dma_$v.o
done
export REPS=20 DTYPE=bf16 B=${B:-8} SIZE=${SIZE:-512}
# SHAPES: layer-name filter of tools/bench_kernels.py (default: the level-0 shapes of configs[3]); with BASE=64 FULL=1
# DEPTH=4 B=4 SIZE=384 in the environment the table is that of configs[4]
for v in ${VARIANTS:-BASE NO_MFMA NO_EPI NO_INDMA NO_MFMA_NO_EPI STORE_LINEAR} BASE; do
  echo "== $v"
  UNETPP_LIB=$R/build/exp/libunetpp_dma_$v.so timeout -k 10 120 python tools/bench_kernels.py ${SHAPES:-X03} 2>&1 | grep "${SHAPES:-X03}"
done
