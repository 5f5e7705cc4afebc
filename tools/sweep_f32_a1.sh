#!/bin/bash
# fp32 headline over UNETPP_F32_A1_MAX_MB (encoder blocks whose first-stage tensor is at most this large write
# a1 = relu(bn(y1)) instead of folding it into conv2's operand load): 0 = never ... 300 = every level.  Same box, alternating.
for round in 1 2; do
  for v in ${VALUES:-0 40 80 160 300}; do
    line=$(UNETPP_F32_A1_MAX_MB=$v python bench.py --no-cpu-baseline --no-launch-timing --no-other-configs --no-live-pmc --steps 40 --warmup 10 --prewarm 10 2>/dev/null | tail -1)
    echo "a1_max_mb=$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
