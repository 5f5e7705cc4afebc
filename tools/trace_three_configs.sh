cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6h
B="--no-cpu-baseline --no-launch-timing --no-other-configs --no-live-pmc --steps 5 --warmup 5 --prewarm 5"
rocprofv3 --kernel-trace --output-format csv -d $O/trace2_c2 -o t -- python3 $R/bench.py $B > $O/trace2_c2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace2_c3 -o t -- python3 $R/bench.py --dtype bf16 --size 512 --batch 8 $B > $O/trace2_c3.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace2_c5 -o t -- python3 $R/bench.py --dtype bf16 --depth 5 --feature-scale 0.5 --in-channels 3 --n-classes 5 --size 384 --batch 4 $B > $O/trace2_c5.log 2>&1
cd $R
for c in c2 c3 c5; do echo "# $c"; python tools/step_census.py $O/trace2_$c/t_kernel_trace.csv; done
