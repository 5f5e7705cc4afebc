"""Would the backward pass gain from running a layer's input gradient and weight gradient on two streams?  (They depend on
the same dy and on nothing of each other.)  Times the pair back to back on one stream and concurrently on two, for
level-0 decoder shapes of the three benchmarked configurations.  python tools/probes/two_stream_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from unet_nested4tiny_objects_keypoints_amd import engine, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

REPS = 30


def case(name, dtype, b, h, w, cins, cout):
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    mk = lambda *s: torch.randn(*s, generator=g).to(dtype).to(dev)   # noqa: E731
    xs = [mk(b, h, w, c) for c in cins]
    dy = mk(b, h, w, cout)
    wt = (torch.randn(cout, sum(cins), 3, 3, generator=g) * 0.05).to(dev)
    wd = engine.pack_conv_dgrad(wt)
    dxs = [torch.zeros_like(x) for x in xs]
    dw, db = torch.empty_like(wt), torch.empty(cout, device=dev)
    ci = sum(cins)

    def dgrad():
        ops.gemm_fwd(b, h, w, 9, [V(dy)], [V(o, accumulate=True, gate=x, gate_sum=True) for o, x in zip(dxs, xs)], wd)

    def wgrad():
        ops.wgrad(b, h, w, 9, [V(x) for x in xs], [V(dy)], dw, (1, 9, ci * 9, 0), db)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / REPS

    def serial():
        dgrad()
        wgrad()

    def forked():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            wgrad()
        dgrad()
        cur.wait_stream(s2)

    td, tw, ts, tf = timed(dgrad), timed(wgrad), timed(serial), timed(forked)
    print("%-40s dgrad %.3f  wgrad %.3f  serial %.3f  two streams %.3f ms  (%+.1f %%)" % (name, td, tw, ts, tf, 100 * (tf / ts - 1)))


def main():
    bf = torch.bfloat16
    case("c2 f32 [32x3]->32 256x256 b32", torch.float32, 32, 256, 256, (32, 32, 32), 32)
    case("c2 f32 [64x2]->64 128x128 b32", torch.float32, 32, 128, 128, (64, 64), 64)
    case("c2 f32 [128x2]->128 64x64 b32", torch.float32, 32, 64, 64, (128, 128), 128)
    case("c3 bf16 [32x3]->32 512x512 b8", bf, 8, 512, 512, (32, 32, 32), 32)
    case("c3 bf16 [64x2]->64 256x256 b8", bf, 8, 256, 256, (64, 64), 64)
    case("c5 bf16 [64x5]->64 384x384 b4", bf, 4, 384, 384, (64,) * 5, 64)
    case("c5 bf16 [128x3]->128 192x192 b4", bf, 4, 192, 192, (128,) * 3, 128)
    case("c5 bf16 [256x2]->256 96x96 b4", bf, 4, 96, 96, (256, 256), 256)
    case("c5 bf16 [512x2]->512 48x48 b4", bf, 4, 48, 48, (512, 512), 512)
    case("c5 bf16 [1024x2]->1024 24x24 b4", bf, 4, 24, 24, (1024, 1024), 1024)


if __name__ == "__main__":
    main()
