#!/bin/bash
# Per-kernel times of single weight-gradient shapes (GPU box): rocprofv3 kernel stats of tools/bench_kernels.py <layer>.
# usage: tools/wgrad_probe.sh <outdir> <layer> [<layer> ...]     env: B SIZE DTYPE as tools/bench_kernels.py
R=$PWD
O=$R/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$L -o p -- python3 $R/tools/bench_kernels.py $L > $O/$L.log 2>&1 || exit 1
  echo "== $L"; grep -v amdgpu $O/$L.log | grep "$L" | cut -c1-50,95-130
  s=$(find $O/p_$L -name "*kernel_stats.csv" | head -1)
  python3 - "$s" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "wgrad" in n:
        print("   %-60s calls %4s avg %9.1f us" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  find $O/p_$L -name "*kernel_trace.csv" -delete
done
