"""Device-side step anatomy from a rocprofv3 kernel trace of bench.py (rocprofv3 --kernel-trace -d <dir> -- python3 bench.py ...).

A training step ends with the fused Adam kernel; for the last steps of every run of steps with the same launch count
this prints the step span (Adam end to Adam end), the sum of kernel durations inside it and the idle remainder.
usage: python tools/step_gaps.py <dir-or-db>"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))
for f in dbs:
    con = sqlite3.connect(f)
    rows = list(con.execute("select name, start, end from kernels order by start"))
    steps, cur = [], []
    for k, (name, s, e) in enumerate(rows):
        cur.append((name, s, e))
        last_adam = "FusedAdam" in name and (k + 1 == len(rows) or "FusedAdam" not in rows[k + 1][0])
        if last_adam:   # (the fused optimizer takes a few launches per step)
            steps.append(cur)
            cur = []
    print(f, "kernels", len(rows), "steps", len(steps))
    # group consecutive steps by launch count (one group per configuration / phase)
    groups = []
    for i in range(1, len(steps)):
        n = len(steps[i])
        if groups and groups[-1][0] == n:
            groups[-1][1].append(i)
        else:
            groups.append((n, [i]))
    for n, idx in groups:
        if len(idx) < 8:
            continue
        take = idx[-8:]
        span = sum(steps[i][-1][2] - steps[i - 1][-1][2] for i in take) / len(take) / 1e6
        busy = sum(sum(e - s for _, s, e in steps[i]) for i in take) / len(take) / 1e6
        gaps = []
        for i in take:
            prev_end = steps[i - 1][-1][2]
            for _, s, e in steps[i]:
                gaps.append(max(0, s - prev_end))
                prev_end = max(prev_end, e)
        gaps.sort()
        last = steps[take[-1]]
        prev_end, big = steps[take[-1] - 1][-1][2], []
        for k, (name, s0, e0) in enumerate(last):
            big.append((s0 - prev_end, k, last[k - 1][0][:60] if k else "(previous step)", name[:60]))
            prev_end = max(prev_end, e0)
        for g, k, before, after in sorted(big, reverse=True)[:6]:
            print("      gap %7.1f us before launch %3d: %s -> %s" % (g / 1e3, k, before, after))
        print("  %4d launches/step x %3d steps: span %.3f ms, kernels %.3f ms, idle %.3f ms; gap median %.1f us, p90 %.1f us, max %.1f us"
              % (n, len(idx), span, busy, span - busy, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))
