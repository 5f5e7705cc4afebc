#!/bin/bash
# Round-2 measurement batch (GPU box, from the repo root): bench lines + rocprofv3 kernel stats for the fp32 headline
# configuration and the two bf16 geometries.  Outputs under gpurun_out/r2_final/ (copied into profiles/r2/ by hand).
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2_final
mkdir -p $O
python bench.py > $O/bench_f32_c2.log 2>&1
python bench.py --dtype bf16 --size 512 --batch 8 --no-cpu-baseline > $O/bench_bf16_c4.log 2>&1
python bench.py --dtype f32 --size 512 --batch 8 --no-cpu-baseline > $O/bench_f32_c4.log 2>&1
python bench.py --dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4 --no-cpu-baseline > $O/bench_bf16_c5.log 2>&1
python bench.py --dtype f32 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4 --no-cpu-baseline > $O/bench_f32_c5.log 2>&1
UNETPP_BENCH_SINGLE_DEVICE=1 UNETPP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --prewarm 3 --batch 8 > $O/bench_2rank_rehearsal.log 2>&1
python tools/x00_probe.py > $O/x00_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32_c2 -o p -- python3 $R/bench.py --steps 5 --warmup 2 --prewarm 3 --no-cpu-baseline --no-launch-timing > $O/prof_f32_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16_c4 -o p -- python3 $R/bench.py --dtype bf16 --size 512 --batch 8 --steps 5 --warmup 2 --prewarm 3 --no-cpu-baseline --no-launch-timing > $O/prof_bf16_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16_c5 -o p -- python3 $R/bench.py --dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4 --steps 5 --warmup 2 --prewarm 3 --no-cpu-baseline --no-launch-timing > $O/prof_bf16_c5.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete   # per-dispatch traces are large; the stats files are what is kept
ls -R $O | head -40
# HBM traffic of the fp32 headline configuration: two separate --pmc passes (MI355X_MICROARCH.md), summarised with the
# build's source hash so that bench.py accepts the file for roofline.traffic
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-launch-timing > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-launch-timing > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic_r2.json > $O/pmc_hbm_traffic_r2.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
tail -12 $O/pmc_hbm_traffic_r2.txt
