#!/bin/bash
# whole-step A/B on ONE box: alternates environment settings over short bench.py runs of the two bf16 configurations.
# usage: tools/ab_steps.sh "VAR=a" "VAR=b" ...   (each argument: one environment assignment list, quoted)
C3="--dtype bf16 --size 512 --batch 8"
C5="--dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4"
for round in 1 2; do
  for setting in "$@"; do
    for cfg in C3 C5; do
      eval "ARGS=\$$cfg"
      line=$(env $setting python bench.py $ARGS --no-cpu-baseline --no-launch-timing --no-other-configs --steps 40 --warmup 10 --prewarm 10 2>/dev/null | tail -1)
      echo "$setting $cfg $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
    done
  done
done
