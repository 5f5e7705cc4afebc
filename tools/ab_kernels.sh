#!/bin/bash
# Kernel-level A/B of prebuilt libraries on ONE box: tools/bench_kernels.py (filter $FILTER, default "X0": the level-0
# decoder shapes) for the tree's library and every library path in "$@", ROUNDS (default 2) times alternating.
# usage: FILTER=X03 tools/ab_kernels.sh build/exp/libunetpp_a.so ...
R=$PWD
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in tree "$@"; do
    L=$R/unet_nested4tiny_objects_keypoints_amd/libunetpp_hip.so
    [ "$lib" != tree ] && L=$R/$lib
    echo "== $(basename $lib)"
    UNETPP_LIB=$L REPS=${REPS:-20} python tools/bench_kernels.py ${FILTER:-X0} 2>/dev/null | grep -v "^layer\|^deconv"
  done
done
