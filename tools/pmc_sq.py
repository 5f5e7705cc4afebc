"""Per-kernel averages of the counters of one rocprofv3 --pmc pass (csv output).
usage: python tools/pmc_sq.py <dir containing p_counter_collection.csv> [min_avg_us]
Prints, per kernel, the average duration and every counter's per-dispatch average; with GRBM_GUI_ACTIVE present also
the effective shader clock (GUI_ACTIVE / 8 XCDs / wall time, MI355X_MICROARCH.md 'DVFS give-back') and, with
SQ_VALU_MFMA_BUSY_CYCLES, the MFMA pipe occupancy (busy cycles / (1024 SIMDs x kernel cycles))."""
import collections
import csv
import re
import sys


def key(name):
    name = name.replace("void ", "").replace("unetpp::(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def main(d, min_us=50.0):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open("%s/p_counter_collection.csv" % d)):
        k = key(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(acc, key=lambda k: -sum(dur[k])):
        us = sum(dur[k]) / len(dur[k])
        if us < min_us:
            continue
        print("%s: %d dispatches, avg %.1f us" % (k, len(dur[k]), us))
        av = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
        for c in sorted(av):
            print("    %-32s %16.0f" % (c, av[c]))
        if "GRBM_GUI_ACTIVE" in av:
            cyc = av["GRBM_GUI_ACTIVE"] / 8
            print("    effective clock %.3f GHz" % (cyc / us / 1e3))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in av:
                print("    MFMA pipe occupancy %.3f" % (av["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 50.0)
