#!/bin/bash
# As tools/ab_env.sh, the two bf16 configurations only.   usage: tools/ab_env_bf16.sh VAR=value ...
C3="--dtype bf16 --size 512 --batch 8"
C5="--dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4"
for round in $(seq 1 ${ROUNDS:-3}); do
  for side in default switched; do
    for cfg in C3 C5; do
      eval "ARGS=\$$cfg"
      if [ $side = switched ]; then
        line=$(env "$@" python bench.py $ARGS --no-cpu-baseline --no-launch-timing --no-other-configs --no-live-pmc --steps 40 --warmup 10 --prewarm 10 2>/dev/null | tail -1)
      else
        line=$(python bench.py $ARGS --no-cpu-baseline --no-launch-timing --no-other-configs --no-live-pmc --steps 40 --warmup 10 --prewarm 10 2>/dev/null | tail -1)
      fi
      echo "$side $cfg $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
    done
  done
done
