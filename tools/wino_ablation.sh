#!/bin/bash
# Ablation matrix for gemm_wino_kernel (GPU box): experiment builds of gemm_wino.hip (-DUNETPP_WINO_EXP plus one of
# -DUNETPP_WINO_EXP_NO_STAGING / _NO_STORE / _NO_LOADS / _NO_DMA / _NO_BARRIER / _NO_EPILOGUE, linked into
# build/exp/libunetpp_<name>.so) timed on the X_0,3 shapes; UNETPP_WINO_ONE_PER_CU=1 gives a wave its SIMD to itself.
export REPS=30
for lib in ${LIBS:-BASE BASE NO_STAGING NO_STORE NO_LOADS NO_DMA BASE}; do
  echo "== one_per_cu=${UNETPP_WINO_ONE_PER_CU:-0} lib=$lib"
  UNETPP_LIB=build/exp/libunetpp_$lib.so timeout -k 10 100 python tools/bench_kernels.py X03 2>&1 | grep "X03"
done
