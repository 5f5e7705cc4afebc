"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-kernel HBM bytes per launch.

On the GPU box (gpurun), from /tmp with TMPDIR=/tmp:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-launch-timing
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-launch-timing
then here:  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_hbm_traffic_latest.json

Counter units are KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: gfx950 reports exactly half the bytes of wide
coalesced reads; torch's elementwise kernels in the same trace come out at 6 TB/s with the correction, which
calibrates it).  The counters are collected in separate passes as that guide prescribes (TCC slots).
"""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def key(name):
    name = name.replace("void ", "").replace("unetpp::(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def _build_hash():
    from unet_nested4tiny_objects_keypoints_amd import _lib
    return _lib.source_hash()


def main(fetch_dir, write_dir, out_path):
    res = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "ns": 0})
    for d, ctr in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        for r in csv.DictReader(open("%s/p_counter_collection.csv" % d)):
            if r["Counter_Name"] != ctr:
                continue
            k = key(r["Kernel_Name"])
            res[k][ctr] += float(r["Counter_Value"])
            if ctr == "FETCH_SIZE":
                res[k]["n"] += 1
                res[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = []
    for k, v in res.items():
        if v["n"]:
            rows.append((v["ns"], k, v["n"], 2 * v["FETCH_SIZE"] * 1024 / v["n"], v["WRITE_SIZE"] * 1024 / v["n"]))
    rows.sort(reverse=True)
    out = []
    for ns, k, n, f, w in rows[:20]:
        print("%-34s launches %4d  avg %8.1f us  fetch(x2) %8.2f MB  write %8.2f MB  => %5.2f TB/s" % (
            k[:34], n, ns / n / 1e3, f / 1e6, w / 1e6, (f + w) / (ns / n) / 1e3))
        out.append({"kernel": k, "launches": n, "avg_us": round(ns / n / 1e3, 2),
                    "fetch_bytes_x2_per_launch": round(f), "write_bytes_per_launch": round(w)})
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 2 --warmup 1 "
                       "--no-cpu-baseline --no-launch-timing` (3 train steps + 7 eval forwards); KB x1024; FETCH_SIZE "
                       "doubled per MI355X_MICROARCH.md", "build_hash": _build_hash(), "kernels": out}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
