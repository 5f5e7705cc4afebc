"""Do a weight-gradient launch and the input-gradient launch of the same layer overlap when they are put on two streams?
Both only read dY, so the engine could run every weight gradient beside the input-gradient chain.  Prints, per layer,
the time of the two launches back to back on one stream and side by side on two (events around the pair, 20 repeats).
env as tools/bench_kernels.py: B, SIZE, BASE, DTYPE."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import engine, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

B, SIZE, BASE = int(os.environ.get("B", "32")), int(os.environ.get("SIZE", "256")), int(os.environ.get("BASE", "32"))
BF = os.environ.get("DTYPE", "f32") == "bf16"
REPS = 20


def rnd(*shape, act=True):
    t = torch.randn(*shape, device="cuda")
    return t.to(torch.bfloat16) if (BF and act) else t


side = torch.cuda.Stream()
f = [BASE << i for i in range(4)]
layers = []
for i in range(4):
    layers.append(("X%d0.conv2" % i, SIZE >> i, [f[i]], f[i]))
for j in range(1, 4):
    for i in range(4 - j):
        layers.append(("X%d%d.conv1" % (i, j), SIZE >> i, [f[i]] * (j + 1), f[i]))
tot_seq = tot_par = 0.0
for name, hw, cins, co in layers:
    xs = [rnd(B, hw, hw, c) for c in cins]
    dy = rnd(B, hw, hw, co)
    w = rnd(co, sum(cins), 3, 3, act=False) * 0.05
    wd = engine.pack_conv_dgrad(w)
    dxs = [torch.empty_like(t) for t in xs]
    dw, db = torch.empty_like(w), torch.empty(co, device="cuda")
    dgrad = lambda: ops.gemm_fwd(B, hw, hw, 9, [V(dy)], [V(t) for t in dxs], wd)                                    # noqa: E731
    wgrad = lambda: ops.wgrad(B, hw, hw, 9, [V(t) for t in xs], [V(dy)], dw, (1, 9, sum(cins) * 9, 0), db)            # noqa: E731

    def seq():
        wgrad()
        dgrad()

    def par():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        torch.cuda.current_stream().wait_stream(side)

    res = []
    for fn in (seq, par, seq, par):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(REPS):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / REPS)
    t_seq, t_par = min(res[0], res[2]), min(res[1], res[3])
    tot_seq += t_seq
    tot_par += t_par
    print("%-12s %4d %-18s -> %3d   one stream %.3f ms   two streams %.3f ms   (%+.1f %%)"
          % (name, hw, cins, co, t_seq, t_par, 100 * (t_par / t_seq - 1)), flush=True)
print("TOTAL one stream %.3f ms, two streams %.3f ms (%+.1f %%)" % (tot_seq, tot_par, 100 * (tot_par / tot_seq - 1)))
