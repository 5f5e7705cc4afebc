"""Pointwise GEMMs of the transposed convolutions of BASELINE configs[1] (GPU box): forward, input gradient and weight
gradient per level, each launch timed ALONE with HIP events around the library call only (ops.LaunchTimer: the weight-image
pack launch of an un-planned ops.gemm_fwd call is outside the pair), over SETS rotating operand sets (default 4: about
1.6 GB at level 0, so no launch finds its operands in the 256 MB Infinity Cache -- as in a training step).
  python tools/bench_pw.py            UNETPP_PW_DIRECT=0 / UNETPP_PW_NT=0|1 / UNETPP_LIB=... for A/B runs; env B, SIZE, REPS, SETS"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import engine, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

B = int(os.environ.get("B", "32"))
SIZE = int(os.environ.get("SIZE", "256"))
REPS = int(os.environ.get("REPS", "12"))
SETS = int(os.environ.get("SETS", "4"))
BASE = int(os.environ.get("BASE", "32"))
f = [BASE << i for i in range(4)]


def timed(calls):
    """calls: list of thunks (one per operand set); returns the mean launch time in us and the kernel name"""
    for c in calls:
        c()
    torch.cuda.synchronize()
    t = ops.LaunchTimer()
    t.want_regions = False
    ops.set_timer(t)
    for r in range(REPS):
        calls[r % len(calls)]()
    ops.set_timer(None)
    torch.cuda.synchronize()
    launches, _ = t.summary()
    (name, d), = launches.items()
    return 1e3 * d["ms"] / d["launches"], name


print("%-8s %5s %5s %5s | %-22s %9s %7s | %-22s %9s %7s | %-22s %9s" % ("deconv", "hw_lo", "cin", "cout", "fwd kernel", "us", "TB/s",
                                                                       "dgrad kernel", "us", "TB/s", "wgrad kernel", "us"))
tot = [0.0, 0.0, 0.0]
for i in range(3):
    hw, ci, co = (SIZE // 2) >> i, f[i + 1], f[i]
    sets = []
    for s in range(SETS):
        x = torch.randn(B, hw, hw, ci, device="cuda")
        up = torch.randn(B, 2 * hw, 2 * hw, co, device="cuda")
        sets.append((x, up, torch.empty_like(x), torch.randn(B, hw, hw, ci, device="cuda")))
    w = torch.randn(ci, co, 2, 2, device="cuda") * 0.05
    bias = torch.randn(co, device="cuda")
    wf, wd, b4 = engine.pack_deconv_fwd(w), engine.pack_deconv_dgrad(w), engine.tile_bias4(bias)
    byts = 4.0 * B * hw * hw * (ci + 4 * co)
    t_f, n_f = timed([lambda x=x, up=up: ops.gemm_fwd(B, hw, hw, 1, [V(x)], engine._phase_views(up), wf, b4) for x, up, _, _ in sets])
    # input gradient as the engine launches it: ReLU gate of the source on the accumulated sum
    t_d, n_d = timed([lambda up=up, dx=dx, gt=gt: ops.gemm_fwd(B, hw, hw, 1, engine._phase_views(up),
                                                                [V(dx, accumulate=True, gate=gt, gate_sum=True)], wd)
                      for _, up, dx, gt in sets])
    dw, db = torch.empty_like(w), torch.empty_like(bias)
    # the weight gradient: main launch (ops.LaunchTimer) and the whole call with its finish launch (events around it)
    tw = ops.LaunchTimer()
    tw.want_regions = False
    for x, up, _, _ in sets:
        ops.wgrad(B, hw, hw, 1, [V(x)], engine._phase_views(up), dw, (0, 4 * co, 4, 1), db, n_inner=co)
    torch.cuda.synchronize()
    ops.set_timer(tw)
    for r in range(REPS):
        x, up, _, _ = sets[r % SETS]
        ops.wgrad(B, hw, hw, 1, [V(x)], engine._phase_views(up), dw, (0, 4 * co, 4, 1), db, n_inner=co)
    ops.set_timer(None)
    torch.cuda.synchronize()
    lw, _ = tw.summary()
    t_w = 1e3 * sum(d["ms"] for d in lw.values()) / REPS
    n_w = "+".join(lw.keys())
    pairs = []
    for r in range(REPS):
        x, up, _, _ = sets[r % SETS]
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        ops.wgrad(B, hw, hw, 1, [V(x)], engine._phase_views(up), dw, (0, 4 * co, 4, 1), db, n_inner=co)
        s1.record()
        pairs.append((s0, s1))
    torch.cuda.synchronize()
    t_wf = 1e3 * sum(a.elapsed_time(b) for a, b in pairs) / REPS
    n_w += " (+finish %.1f)" % t_wf
    print("%-8s %5d %5d %5d | %-22s %9.1f %7.2f | %-22s %9.1f %7.2f | %-22s %9.1f" % (
        "level %d" % i, hw, ci, co, n_f, t_f, byts / t_f * 1e-6, n_d, t_d, (byts + 8.0 * B * hw * hw * ci) / t_d * 1e-6, n_w, t_w))
    for k, v in enumerate((t_f, t_d, t_wf)):
        tot[k] += v * (3 - i)  # launches per step: three transposed convolutions at level 0, two at level 1, one at level 2
    del sets
    torch.cuda.empty_cache()
print("per training step (3 + 2 + 1 layers): forward %.3f ms, input gradient %.3f ms, weight gradient %.3f ms, sum %.3f ms" % (
    tot[0] * 1e-3, tot[1] * 1e-3, tot[2] * 1e-3, sum(tot) * 1e-3))
