"""Kernel microbenchmark (GPU box): the MFMA GEMM / wgrad launches of BASELINE configs[1] (base 32, 256x256, batch 32),
each timed alone with HIP events.  Prints TFLOP/s per shape and the FLOP-weighted total.  Used to iterate on kernels.
env: B (batch), SIZE (256), DTYPE=bf16 (the bf16-storage kernels; also prints the fraction of max(HBM, MFMA) floor)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import engine, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

B = int(os.environ.get("B", "32"))
REPS = int(os.environ.get("REPS", "5"))
only = sys.argv[1] if len(sys.argv) > 1 else ""
SIZE = int(os.environ.get("SIZE", "256"))
BF = os.environ.get("DTYPE", "f32") == "bf16"
ES = 2.0 if BF else 4.0
PEAK_TF = 2500.0 if BF else 157.3


def floor_ms(flops, elems):
    return max(flops / (PEAK_TF * 1e12), elems * ES / 8e12) * 1e3


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS


def rnd(*shape, act=True):
    t = torch.randn(*shape, device="cuda")
    return t.to(torch.bfloat16) if (BF and act) else t


# (name, level spatial, [cin per view], cout)
CONVS = []
BASE = int(os.environ.get("BASE", "32"))
if os.environ.get("FULL", "0") == "1":   # every 3x3 layer of a depth-DEPTH network (bottom level and its decoders included)
    L = int(os.environ.get("DEPTH", "4"))
    f = [BASE << i for i in range(L + 1)]
    n_lev, n_dec = L + 1, L
else:                                     # the four upper levels of the depth-4 network (tables of rounds 1-3)
    f = [BASE << i for i in range(4)]
    n_lev, n_dec = 4, 3
for i in range(n_lev):
    hw = SIZE >> i
    CONVS.append(("enc%d.conv2" % i, hw, [f[i]], f[i]))
    if i > 0:
        CONVS.append(("enc%d.conv1" % i, hw, [f[i - 1]], f[i]))
for j in range(1, n_dec + 1):
    for i in range(n_lev - j):
        hw = SIZE >> i
        CONVS.append(("X%d%d.conv1" % (i, j), hw, [f[i]] * (j + 1), f[i]))
        CONVS.append(("X%d%d.conv2" % (i, j), hw, [f[i]], f[i]))

tot_ms = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
tot_fl = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print("%-14s %5s %-22s %5s | %9s %7s | %9s %7s | %9s %7s" % ("layer", "hw", "cin", "cout", "fwd ms", "TF/s", "dgrad ms",
                                                               "TF/s", "wgrad ms", "TF/s"))
for name, hw, cins, co in CONVS:
    if only and only not in name:
        continue
    ci = sum(cins)
    xs = [rnd(B, hw, hw, c) for c in cins]
    y = rnd(B, hw, hw, co)
    w = rnd(co, ci, 3, 3, act=False) * 0.05
    bias = rnd(co, act=False)
    wp, wd = engine.pack_conv_fwd(w), engine.pack_conv_dgrad(w)
    flops = 2.0 * B * hw * hw * 9 * ci * co
    t_f = timeit(lambda: ops.gemm_fwd(B, hw, hw, 9, [V(t) for t in xs], [V(y, relu=True)], wp, bias))
    dxs = [torch.empty_like(t) for t in xs]
    t_d = timeit(lambda: ops.gemm_fwd(B, hw, hw, 9, [V(y)], [V(t) for t in dxs], wd))
    dw, db = torch.empty_like(w), torch.empty_like(bias)
    t_w = timeit(lambda: ops.wgrad(B, hw, hw, 9, [V(t) for t in xs], [V(y)], dw, (1, 9, ci * 9, 0), db,
                                   target_blocks=int(os.environ.get("TB", "256"))))
    for k, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
        tot_ms[k] += t
        tot_fl[k] += flops
    fl = floor_ms(flops, float(B) * hw * hw * (ci + co))
    print("%-14s %5d %-22s %5d | %9.3f %7.1f | %9.3f %7.1f | %9.3f %7.1f | floor %.3f ms: %.2f %.2f %.2f" % (
        name, hw, str(cins), co, t_f, flops / t_f / 1e9, t_d, flops / t_d / 1e9, t_w, flops / t_w / 1e9, fl, fl / t_f,
        fl / t_d, fl / t_w))
    tot_ms.setdefault("floor", 0.0)
    tot_ms["floor"] += fl
for k in ("fwd", "dgrad", "wgrad"):
    if tot_ms[k] > 0:
        print("TOTAL %-6s %8.3f ms  %7.1f TF/s   (sum of floors %.3f ms: %.2f)" % (
            k, tot_ms[k], tot_fl[k] / tot_ms[k] / 1e9, tot_ms.get("floor", 0.0), tot_ms.get("floor", 0.0) / tot_ms[k]))

# ---- 2x2 stride-2 transposed convolutions (pointwise GEMM + pixel phases) ----
print("%-14s %5s %5s %5s | %9s %7s %7s | %9s %7s | %9s %7s" % ("deconv", "hw_lo", "cin", "cout", "fwd ms", "TF/s", "GB/s",
                                                               "dgrad ms", "TF/s", "wgrad ms", "TF/s"))
for i in range(len(f) - 1):
    if only and "deconv" not in only:
        break
    hw = (SIZE // 2) >> i
    ci, co = f[i + 1], f[i]
    x = rnd(B, hw, hw, ci)
    up = rnd(B, 2 * hw, 2 * hw, co)
    w = rnd(ci, co, 2, 2, act=False) * 0.05
    bias = rnd(co, act=False)
    wf, wd, b4 = engine.pack_deconv_fwd(w), engine.pack_deconv_dgrad(w), engine.tile_bias4(bias)
    flops = 2.0 * B * hw * hw * ci * 4 * co
    byts = ES * B * hw * hw * (ci + 4 * co)
    t_f = timeit(lambda: ops.gemm_fwd(B, hw, hw, 1, [V(x)], engine._phase_views(up), wf, b4))
    dx = torch.empty_like(x)
    t_d = timeit(lambda: ops.gemm_fwd(B, hw, hw, 1, engine._phase_views(up), [V(dx)], wd))
    dw, db = torch.empty_like(w), torch.empty_like(bias)
    t_w = timeit(lambda: ops.wgrad(B, hw, hw, 1, [V(x)], engine._phase_views(up), dw, (0, 4 * co, 4, 1), db, n_inner=co))
    print("%-14s %5d %5d %5d | %9.3f %7.1f %7.0f | %9.3f %7.1f | %9.3f %7.1f" % (
        "deconv%d" % i, hw, ci, co, t_f, flops / t_f / 1e9, byts / t_f / 1e6, t_d, flops / t_d / 1e9, t_w, flops / t_w / 1e9))
