#!/bin/bash
# Whole-step A/B of ONE recompiled source on one box: builds build/exp/libunetpp_alt.so = the tree's objects with
# csrc/$1 recompiled with the flags in $2, then alternates it with the tree's library over short bench.py runs of the
# two bf16 configurations (and the fp32 headline with HEADLINE=1).   usage: tools/ab_lib.sh gemm_bf16.hip "-DFOO=2"
set -e
R=$PWD
SRC=$1
FLAGS=$2
CS=unet_nested4tiny_objects_keypoints_amd/csrc
mkdir -p build/exp
OBJ=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $CS $FLAGS -c $CS/$SRC -o build/exp/alt_$OBJ.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_alt.so $(ls build/obj/*.o | grep -v "/$OBJ.o") build/exp/alt_$OBJ.o
set +e
C3="--dtype bf16 --size 512 --batch 8"
C5="--dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4"
C2=""
CFGS="C3 C5"
[ -n "$HEADLINE" ] && CFGS="C2"
for round in 1 2; do
  for lib in tree alt; do
    L=$R/unet_nested4tiny_objects_keypoints_amd/libunetpp_hip.so
    [ $lib = alt ] && L=$R/build/exp/libunetpp_alt.so
    for cfg in $CFGS; do
      eval "ARGS=\$$cfg"
      line=$(UNETPP_LIB=$L python bench.py $ARGS --no-cpu-baseline --no-launch-timing --no-other-configs --steps 40 --warmup 10 --prewarm 10 2>/dev/null | tail -1)
      echo "$lib $cfg $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
    done
  done
done
