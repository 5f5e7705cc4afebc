#!/bin/bash
# Whole-step A/B of SEVERAL builds of one source on one box: for every flag set in "$@" (after the source name) builds
# build/exp/libunetpp_alt<i>.so = the tree's objects with csrc/$1 recompiled with those flags, then alternates the tree's
# library and the alternatives over short bench.py runs.  HEADLINE=1: the fp32 headline, else the two bf16 configurations.
# usage: tools/ab_multi.sh gemm_wino.hip "-DFOO=1" "-DFOO=2" ...      (EXTRA="-fno-slp-vectorize" for the Winograd sources)
set -e
R=$PWD
SRC=$1
shift
CS=unet_nested4tiny_objects_keypoints_amd/csrc
mkdir -p build/exp
OBJ=$(basename $SRC .hip)
i=0
LIBS="tree"
for FLAGS in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $CS $EXTRA $FLAGS -c $CS/$SRC -o build/exp/alt${i}_$OBJ.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libunetpp_alt$i.so $(ls build/obj/*.o | grep -v "/$OBJ.o") build/exp/alt${i}_$OBJ.o
  LIBS="$LIBS alt$i"
done
set +e
C3="--dtype bf16 --size 512 --batch 8"
C5="--dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4"
C2=""
CFGS="C3 C5"
[ -n "$HEADLINE" ] && CFGS="C2"
for round in 1 2; do
  for lib in $LIBS; do
    L=$R/unet_nested4tiny_objects_keypoints_amd/libunetpp_hip.so
    [ $lib != tree ] && L=$R/build/exp/libunetpp_$lib.so
    for cfg in $CFGS; do
      eval "ARGS=\$$cfg"
      line=$(UNETPP_LIB=$L python bench.py $ARGS --no-cpu-baseline --no-launch-timing --no-other-configs --no-live-pmc --steps 40 --warmup 10 --prewarm 10 2>/dev/null | tail -1)
      echo "$lib $cfg $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
    done
  done
done
