#!/bin/bash
# Whole-step A/B of prebuilt libraries on ONE box: alternates the tree's library ("tree") and every library named in "$@"
# (paths, e.g. build/exp/libunetpp_r5.so) over short bench.py runs.  HEADLINE=1: the fp32 headline, else both bf16
# configurations; ROUNDS (default 2).   usage: HEADLINE=1 tools/ab_libs.sh build/exp/libunetpp_r5.so ...
R=$PWD
C3="--dtype bf16 --size 512 --batch 8"
C5="--dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4"
C2=""
CFGS="C3 C5"
[ -n "$HEADLINE" ] && CFGS="C2"
[ -n "$ALLCFG" ] && CFGS="C2 C3 C5"
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in tree "$@"; do
    L=$R/unet_nested4tiny_objects_keypoints_amd/libunetpp_hip.so
    [ "$lib" != tree ] && L=$R/$lib
    for cfg in $CFGS; do
      eval "ARGS=\$$cfg"
      line=$(UNETPP_LIB=$L python bench.py $ARGS --no-cpu-baseline --no-launch-timing --no-other-configs --no-live-pmc --steps 40 --warmup 10 --prewarm 10 2>/dev/null | tail -1)
      echo "$(basename $lib) $cfg $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
    done
  done
done
