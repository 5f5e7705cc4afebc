"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-kernel HBM bytes per launch.

On the GPU box (gpurun), from /tmp with TMPDIR=/tmp:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-launch-timing
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-launch-timing
then here:  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_hbm_traffic_latest.json [config key [sq dir]]

The output file holds one entry per benchmark configuration (`config key` = bench.config_key(args): dtype, depth,
widths, size, batch, channels; default = the fp32 headline), each stamped with the hash of the sources the library was
built from; entries of other configurations already in the file are kept.  `sq dir` (optional): a third pass with
`--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
SQ_WAIT_INST_ANY SQ_WAIT_ANY` -- per kernel the matrix-pipe occupancy (MFMA busy cycles / (1024 SIMDs x kernel cycles)),
the effective clock and the instruction mix per dispatch go into the same entry.

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


DEFAULT_KEY = "f32_d4_fs1_s256_b32_i1_c4"


def sq_summary(sq_dir):
    """per kernel: averages per dispatch of one SQ/GRBM pass (tools/pmc_sq.py prints the same numbers)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open("%s/p_counter_collection.csv" % sq_dir)):
        k = key(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = {}
    for k in acc:
        us = sum(dur[k].values()) / len(dur[k])
        av = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
        row = {"dispatches": len(dur[k]), "avg_us": round(us, 2)}
        if "GRBM_GUI_ACTIVE" in av:
            cyc = av["GRBM_GUI_ACTIVE"] / 8
            row["effective_clock_ghz"] = round(cyc / us / 1e3, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in av:
                row["mfma_busy"] = round(av["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY",
                  "SQ_WAIT_ANY"):
            if c in av:
                row[c.lower()] = round(av[c])
        out[k] = row
    return out


def main(fetch_dir, write_dir, out_path, cfg_key=DEFAULT_KEY, sq_dir=None):
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
    sq = sq_summary(sq_dir) if sq_dir else {}
    rows = []
    for k, v in res.items():
        if v["n"]:
            rows.append((v["ns"], k, v["n"], 2 * v["FETCH_SIZE"] * 1024 / v["n"], v["WRITE_SIZE"] * 1024 / v["n"]))
    rows.sort(reverse=True)
    out = []
    print("configuration %s" % cfg_key)
    for ns, k, n, f, w in rows[:24]:
        extra = ""
        if k in sq and "mfma_busy" in sq[k]:
            extra = "  MFMA busy %.3f @ %.2f GHz" % (sq[k]["mfma_busy"], sq[k]["effective_clock_ghz"])
        print("%-34s launches %4d  avg %8.1f us  fetch(x2) %8.2f MB  write %8.2f MB  => %5.2f TB/s%s" % (
            k[:34], n, ns / n / 1e3, f / 1e6, w / 1e6, (f + w) / (ns / n) / 1e3, extra))
        row = {"kernel": k, "launches": n, "avg_us": round(ns / n / 1e3, 2),
               "fetch_bytes_x2_per_launch": round(f), "write_bytes_per_launch": round(w),
               "hbm_gb_per_s": round((f + w) / (ns / n), 1)}
        if k in sq:
            row["sq"] = sq[k]
        out.append(row)
    doc = {"configs": {}}
    try:
        old = json.load(open(out_path))
        if "configs" in old:
            doc = old
    except (OSError, ValueError):
        pass
    doc["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (/ SQ + GRBM), separate passes over `bench.py --steps 2 --warmup 1 "
                   "--prewarm 0 --no-cpu-baseline --no-launch-timing` of the named configuration (3 train steps + 7 eval "
                   "forwards); KB x1024; FETCH_SIZE doubled per MI355X_MICROARCH.md; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                   "(1024 SIMDs x GRBM_GUI_ACTIVE / 8)")
    doc["configs"][cfg_key] = {"build_hash": _build_hash(), "kernels": out}
    json.dump(doc, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:6])
