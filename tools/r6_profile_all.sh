#!/bin/bash
# Round-6 measurement batch (GPU box, from the repo root).  usage: tools/r6_profile_all.sh [tag] [what...]
#   what: bench (bench lines of the five configurations), stats (rocprofv3 --kernel-trace --stats), pmc (FETCH / WRITE /
#   SQ passes -> profiles-style json per configuration), rehearsal (4 ranks on the one GPU over gloo).  Default: all.
# Outputs under gpurun_out/r6_<tag>/ (copied into profiles/r6/ by hand).  Every rocprofv3 line has python3 itself after `--`.
set -o pipefail
R=$PWD
TAG=${1:-final}
shift
WHAT=${*:-bench stats pmc rehearsal}
O=$R/gpurun_out/r6_$TAG
mkdir -p $O
C2=""
C4="--dtype bf16 --size 512 --batch 8"
C5="--dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4"
K2=f32_d4_fs1_s256_b32_i1_c4
K4=bf16_d4_fs1_s512_b8_i1_c4
K5=bf16_d5_fs0.5_s384_b4_i3_c5
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  SQ="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY"
  for cfg in 2 4 5; do
    eval "ARGS=\$C$cfg; KEY=\$K$cfg"
    B="--steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-launch-timing --no-other-configs"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$cfg -o p -- python3 $R/bench.py $ARGS $B > $O/pmc_fetch_$cfg.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$cfg -o p -- python3 $R/bench.py $ARGS $B > $O/pmc_write_$cfg.log 2>&1 || exit 1
    rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq_$cfg -o p -- python3 $R/bench.py $ARGS $B > $O/pmc_sq_$cfg.log 2>&1 || exit 1
    (cd $R && python tools/pmc_traffic.py $O/pmc_fetch_$cfg $O/pmc_write_$cfg $O/pmc_hbm_traffic.json $KEY $O/pmc_sq_$cfg > $O/pmc_summary_$KEY.txt 2>&1) || exit 1
    (cd $R && python tools/pmc_sq.py $O/pmc_sq_$cfg 20 > $O/pmc_sq_$KEY.txt 2>&1)
    rm -rf $O/pmc_fetch_$cfg $O/pmc_write_$cfg $O/pmc_sq_$cfg
    tail -14 $O/pmc_summary_$KEY.txt
  done
  cd $R
  # the bench lines below read this file for roofline.traffic / mfma_busy_pmc (same build: the hash matches)
  cp $O/pmc_hbm_traffic.json $R/profiles/pmc_hbm_traffic_latest.json
fi
if has bench; then
  python bench.py > $O/bench_default.log 2>&1 || exit 1     # the driver's command: headline + other_configs
  python bench.py --no-other-configs > $O/bench_f32_c2.log 2>&1 || exit 1
  python bench.py $C4 --no-cpu-baseline --no-other-configs > $O/bench_bf16_c4.log 2>&1 || exit 1
  python bench.py $C5 --no-cpu-baseline --no-other-configs > $O/bench_bf16_c5.log 2>&1 || exit 1
  python bench.py --dtype f32 --size 512 --batch 8 --no-cpu-baseline > $O/bench_f32_c4.log 2>&1 || exit 1
  python bench.py --dtype f32 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4 --no-cpu-baseline > $O/bench_f32_c5.log 2>&1 || exit 1
  python tools/x00_probe.py > $O/x00_probe.txt 2>&1
  grep -h '"metric"' $O/bench_*.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['dtype'], d['config']['workload'][:60], d['value'], 'img/s', d['ms_per_step'], 'ms', r['kernel'], r['frac'], r.get('traffic_over_algorithmic'), r.get('mfma_busy_pmc'), (d.get('roofline_x00') or {}).get('frac'))
"
fi
if has rehearsal; then
  UNETPP_BENCH_SINGLE_DEVICE=1 UNETPP_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 10 --warmup 3 --prewarm 3 --batch 8 > $O/bench_4rank_rehearsal.log 2>&1 || exit 1
  tail -1 $O/bench_4rank_rehearsal.log | cut -c1-700
fi
if has stats; then
  cd /tmp && export TMPDIR=/tmp
  P="--steps 5 --warmup 2 --prewarm 3 --no-cpu-baseline --no-launch-timing --no-other-configs"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32_c2 -o p -- python3 $R/bench.py $P > $O/prof_f32_c2.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16_c4 -o p -- python3 $R/bench.py $C4 $P > $O/prof_bf16_c4.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16_c5 -o p -- python3 $R/bench.py $C5 $P > $O/prof_bf16_c5.log 2>&1 || exit 1
  cd $R
  find $O -name "*kernel_trace.csv" -delete   # per-dispatch traces are large; the stats files are what is kept
fi
ls $O | head -40
