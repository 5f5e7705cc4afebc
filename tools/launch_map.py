"""Per-launch map of one training step: every MFMA-kernel launch of a configuration with its HIP-event time, its
algorithmic FLOPs / bytes and the time it loses against a PRACTICAL floor (max of FLOPs / practical matrix rate and
bytes / practical HBM rate), sorted by lost time and grouped by (kernel, FLOPs, bytes) signature.

    python tools/launch_map.py [c2|c3|c5] [steps]

Practical rates (what the best launches of this repository reach, not the data-sheet peaks): fp32 MFMA 157 TFLOP/s x 0.75
(x2.25 for the Winograd kernels), bf16 MFMA 1.3 PFLOP/s, HBM 4.8 TB/s.  Output: gpurun_out/launch_map_<cfg>.txt"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, ops, train_step  # noqa: E402

CFG = {"c2": dict(dtype="f32", size=256, batch=32, fs=1, depth=4, cin=1, ncls=4),
       "c3": dict(dtype="bf16", size=512, batch=8, fs=1, depth=4, cin=1, ncls=4),
       "c5": dict(dtype="bf16", size=384, batch=4, fs=0.5, depth=5, cin=3, ncls=5)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c5"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    c = CFG[name]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = UNet_Nested(in_channels=c["cin"], n_classes=c["ncls"], feature_scale=c["fs"], depth=c["depth"]).to(dev).train()
    if c["dtype"] == "bf16":
        model.set_activation_dtype(torch.bfloat16)
    x = torch.randn(c["batch"], c["cin"], c["size"], c["size"], device=dev)
    t = torch.rand(c["batch"], c["ncls"], c["size"], c["size"], device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    for _ in range(12):
        train_step(model, opt, crit, x, t)
    timer = ops.LaunchTimer()
    timer.want_regions = False
    ops.set_timer(timer)
    for _ in range(steps):
        train_step(model, opt, crit, x, t)
    torch.cuda.synchronize()
    ops.set_timer(None)
    per_step = len(timer.launches) // steps
    sig = collections.OrderedDict()
    for i, (kind, flops, nbytes, s, e) in enumerate(timer.launches):
        k = (i % per_step, kind, flops, nbytes)   # position in the step: the same launch of every step
        sig.setdefault(k, []).append(s.elapsed_time(e))
    bf = c["dtype"] == "bf16"
    rows = []
    for (pos, kind, flops, nbytes), ms in sig.items():
        t_ms = sorted(ms)[len(ms) // 2]
        rate = 1.3e15 if bf else 157.3e12 * 0.75 * (2.25 if "wino" in kind else 1.0)
        floor = max(flops / rate, nbytes / 4.8e12) * 1e3
        rows.append((t_ms - floor, pos, kind, flops, nbytes, t_ms, floor))
    total = sum(r[5] for r in rows)
    out = ["configuration %s: %d timed launches per step, %.3f ms in them, %.3f ms above the practical floors" % (
        name, per_step, total, sum(r[0] for r in rows))]
    out.append("%4s %-44s %9s %9s %8s %8s %7s %7s" % ("pos", "kernel", "GFLOP", "MB", "ms", "floor", "lost", "TF/s"))
    for lost, pos, kind, flops, nbytes, t_ms, floor in sorted(rows, reverse=True):
        out.append("%4d %-44s %9.2f %9.1f %8.4f %8.4f %7.4f %7.1f" % (pos, kind[:44], flops / 1e9, nbytes / 1e6, t_ms, floor, lost,
                                                                     flops / (t_ms * 1e-3) / 1e12))
    by_kind = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for lost, pos, kind, flops, nbytes, t_ms, floor in rows:
        b = by_kind[kind]
        b[0] += 1
        b[1] += t_ms
        b[2] += lost
    out.append("")
    for kind, (n, t_ms, lost) in sorted(by_kind.items(), key=lambda kv: -kv[1][2]):
        out.append("%-44s launches %3d  ms %7.3f  above floor %7.3f" % (kind[:44], n, t_ms, lost))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "launch_map_%s.txt" % name)
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(out[:40]))


if __name__ == "__main__":
    main()
