"""Launch census of ONE training step from a `rocprofv3 --kernel-trace --output-format csv` of bench.py (the last of its
steps, found by the optimizer's multi_tensor_apply launches): dispatches, span, idle time between kernels, and the
kernels shorter than 12 us grouped by name -- how the launch diet of round 6 was sized.
usage: python tools/step_census.py <..._kernel_trace.csv> [steps in the trace (default 15)]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
per = len(mt) // steps
ends = [mt[(s + 1) * per - 1] for s in range(steps)]
sel = rows[ends[steps - 2] + 1:ends[steps - 1] + 1]


def short(n):
    n = re.sub(r"unetpp::\(anonymous namespace\)::|\(anonymous namespace\)::|^void |at::native::", "", n)
    return n.split("(")[0][:70]


t0 = int(sel[0]["Start_Timestamp"])
prev, idle = t0, 0
count, time = collections.Counter(), collections.Counter()
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > prev:
        idle += s - prev
    prev = max(prev, e)
    if e - s < 12000:
        count[short(r["Kernel_Name"])] += 1
        time[short(r["Kernel_Name"])] += (e - s) / 1e3
print("%d dispatches, span %.1f us, idle between kernels %.1f us" % (len(sel), (prev - t0) / 1e3, idle / 1e3))
print("kernels under 12 us: %d launches, %.1f us" % (sum(count.values()), sum(time.values())))
for k, v in count.most_common():
    print("%4d %7.1f us  %s" % (v, time[k], k))
