"""Registers, spills, scratch and LDS of every gfx950 kernel of the library, from hipcc's own resource report
(-Rpass-analysis=kernel-resource-usage; runs on the CPU, no GPU needed).

  python tools/kernel_resources.py [file.hip ...] [--spills-only] [-D MACRO ...]

ScratchSize is checked as well as the spill counts: hipcc reports a lambda that is not inlined as scratch, not as
spills.  tests/test_isa_hazards.py asserts the no-spill list of this tool's output.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unet_nested4tiny_objects_keypoints_amd", "csrc")
SLP_OFF = {"gemm_wino.hip", "wgrad_wino.hip"}   # built with -fno-slp-vectorize (_lib.build_library)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"unetpp::\(anonymous namespace\)::|unetpp::", "", n) for n in out]


def report(path, defines=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
           "-I", CSRC, "-c", path, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + ["-D" + d for d in defines]
    if os.path.basename(path) in SLP_OFF:
        cmd.insert(5, "-fno-slp-vectorize")
    run = subprocess.run(cmd, capture_output=True, text=True)
    if run.returncode != 0:   # a source that does not compile reports no kernels at all -- which must not read as "no spills"
        raise RuntimeError("hipcc failed on %s:\n%s" % (path, "\n".join(l for l in run.stderr.splitlines() if "error" in l)[:2000]))
    err = run.stderr
    rows, cur = [], None
    for ln in err.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("sspill", r"SGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, ln)
            if m:
                cur[key] = int(m.group(1))
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = n
    return rows


def all_reports(files=None, defines=(), workers=6):
    """[(file, rows)] for the given sources (default: every .hip of the library), compiled in parallel"""
    from concurrent.futures import ThreadPoolExecutor
    files = files or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    paths = [f if os.path.exists(f) else os.path.join(CSRC, f) for f in files]
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(zip(files, ex.map(lambda p: report(p, defines), paths)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    spills_only = "--spills-only" in sys.argv
    defines = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "-D"]
    args = [a for a in args if a not in defines]
    files = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for f, rows in all_reports(files, defines):
        for r in rows:
            spilled = r.get("vspill", 0) or r.get("sspill", 0) or r.get("scratch", 0)
            bad += int(bool(r.get("vspill", 0) or r.get("scratch", 0)))
            if spills_only and not spilled:
                continue
            print("%-28s %-96s vgpr %3d agpr %3d sgpr %3d  spill v %3d s %3d  scratch %4d  occ %d  lds %6d" % (
                os.path.basename(f), r["name"][:96], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("vspill", 0),
                r.get("sspill", 0), r.get("scratch", 0), r.get("occ", -1), r.get("lds", 0)))
    print("kernels with vector spills or scratch:", bad)


if __name__ == "__main__":
    main()
