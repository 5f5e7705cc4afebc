#!/bin/bash
# builds the stamped gemm_bf16_dma.hip into build/exp/ and runs tools/dma_stamps.py on a few shapes (GPU box, repo root)
set -e
R=$PWD
mkdir -p build/exp
HIPCC=/opt/rocm/bin/hipcc
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUNETPP_DMA_STAMPS $EXTRA -I include -I unet_nested4tiny_objects_keypoints_amd/csrc -c unet_nested4tiny_objects_keypoints_amd/csrc/gemm_bf16_dma.hip -o build/exp/dma_stamps.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_dstamps.so $(ls build/obj/*.o | grep -v gemm_bf16_dma.o) build/exp/dma_stamps.o
export UNETPP_LIB=$R/build/exp/libunetpp_dstamps.so
python tools/dma_stamps.py 64,64,64,64,64 64 384 4
python tools/dma_stamps.py 64,64,64,64,64 64 384 4 dgrad
python tools/dma_stamps.py 128,128,128 128 192 4
python tools/dma_stamps.py 32,32,32,32 32 512 8
python tools/dma_stamps.py 32,32,32,32 32 512 8 dgrad
