#!/bin/bash
# Copies the evidence of one tools/r6_profile_all.sh batch from gpurun_out/r6_<tag>/ into profiles/r6/ (and the PMC
# file bench.py reads into profiles/).  usage: tools/r6_collect.sh <tag> [prefix]   (prefix e.g. "base_")
set -e
TAG=${1:?tag}
PRE=${2:-}
O=gpurun_out/r6_$TAG
D=profiles/r6
for f in $O/bench_*.log; do
  n=$(basename $f .log)
  grep -h '"metric"' $f | tail -1 > $D/${PRE}$n.json
done
for c in f32_c2 bf16_c4 bf16_c5; do
  s=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && cp $s $D/${PRE}kernel_stats_$c.csv
done
cp $O/pmc_summary_*.txt $O/pmc_sq_*.txt $D/ 2>/dev/null || true
[ -f $O/x00_probe.txt ] && cp $O/x00_probe.txt $D/${PRE}x00_probe.txt
if [ -f $O/pmc_hbm_traffic.json ]; then
  cp $O/pmc_hbm_traffic.json $D/pmc_hbm_traffic_${PRE:-r6}.json
  [ -z "$PRE" ] && cp $O/pmc_hbm_traffic.json profiles/pmc_hbm_traffic_latest.json
fi
python - <<PY
import json
d = json.load(open("$O/pmc_hbm_traffic.json"))
print("pmc build_hash", {k: v.get("build_hash") for k, v in d["configs"].items()})
PY
