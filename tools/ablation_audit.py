"""Audit of the experiment (ablation) builds: an ablation that removes a CONSUMER of the accumulators can make the MFMAs
dead code -- hipcc then deletes them and the variant measures more than its name says (that happened to the no-epilogue
variant of gemm_bf16_dma.hip in rounds 3-4).  Compiles every experiment macro of the kernel sources to gfx950 ISA (CPU
only) and prints the v_mfma / global_store / LDS-DMA / ds_read counts beside those of the normal build.

    python tools/ablation_audit.py
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unet_nested4tiny_objects_keypoints_amd", "csrc")
SLP_OFF = {"gemm_wino.hip", "wgrad_wino.hip"}
NEEDS = {"UNETPP_WINO_EXP_": ["UNETPP_WINO_EXP"]}   # variants that only exist under an umbrella macro
WANT_FEWER_MFMA = ("NO_MFMA", "NO_COMPUTE")
SEAMS = {"wino_experiments.h": "gemm_wino.hip", "dma_experiments.h": "gemm_bf16_dma.hip"}  # switches kept in a header of their own


def counts(path, defines):
    with tempfile.NamedTemporaryFile(suffix=".s", dir="/tmp", delete=False) as f:
        out = f.name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
           "-I", CSRC, "-S", "--cuda-device-only", "-o", out, path] + ["-D" + d for d in defines]
    if os.path.basename(path) in SLP_OFF:
        cmd.insert(5, "-fno-slp-vectorize")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        os.unlink(out)
        return None
    text = open(out).read()
    os.unlink(out)
    return {"mfma": len(re.findall(r"\bv_mfma", text)), "store": len(re.findall(r"\bglobal_store|\bbuffer_store", text)),
            "dma": len(re.findall(r"\bbuffer_load_dword\S* .* lds|\bglobal_load_lds", text)),
            "ds_read": len(re.findall(r"\bds_read", text))}


def audit(workers=8):
    """-> (report lines, variants that silently lose MFMAs, variants that do not compile)"""
    per_source = {}
    for f in sorted(os.listdir(CSRC)):
        if not (f.endswith(".hip") or f in SEAMS):
            continue
        src = open(os.path.join(CSRC, f)).read()
        macros = set(re.findall(r"#\s*(?:el)?if(?:n?def| defined\()?\s*\(?(UNETPP_[A-Z0-9_]*EXP_[A-Z0-9_]+)", src))
        if macros:
            per_source.setdefault(SEAMS.get(f, f), set()).update(macros)
    jobs = []
    for f, macros in sorted(per_source.items()):
        jobs.append((f, []))
        for m in sorted(macros):
            extra = [u for k, us in NEEDS.items() if m.startswith(k) for u in us]
            jobs.append((f, extra + [m]))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        res = list(ex.map(lambda j: counts(os.path.join(CSRC, j[0]), j[1]), jobs))
    base, bad, broken, lines = {}, [], [], []
    for (f, d), c in zip(jobs, res):
        if not d:
            base[f] = c
        name = d[-1] if d else "(normal build)"
        if c is None:
            lines.append("%-22s %-34s does not compile" % (f, name))
            broken.append((f, name))
            continue
        note = ""
        if d and c["mfma"] < base[f]["mfma"] and not any(w in name for w in WANT_FEWER_MFMA):
            note = "   <-- MFMAs removed although the variant does not say so"
            bad.append((f, name))
        lines.append("%-22s %-34s v_mfma %5d  stores %4d  lds-dma %4d  ds_read %5d%s" % (f, name, c["mfma"], c["store"], c["dma"], c["ds_read"], note))
    return lines, bad, broken


def main():
    lines, bad, broken = audit()
    print("\n".join(lines))
    print("variants that silently lose MFMAs:", len(bad), " variants that do not compile:", len(broken))
    return 1 if (bad or broken) else 0


if __name__ == "__main__":
    sys.exit(main())
