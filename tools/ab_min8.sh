C3="--dtype bf16 --size 512 --batch 8"
C5="--dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4"
for round in 1 2; do
 for v in 2 3 5 100; do
  for cfg in C3 C5; do
    eval "ARGS=\$$cfg"
    line=$(UNETPP_BF16_DMA_MIN8=$v python bench.py $ARGS --no-cpu-baseline --no-launch-timing --no-other-configs --no-live-pmc --steps 40 --warmup 10 --prewarm 10 2>/dev/null | tail -1)
    echo "min8=$v $cfg $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
 done
done
