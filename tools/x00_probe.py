"""X_0,0 block forward (BASELINE configs[1]: 1 -> 32 -> 32 channels at 256x256, batch 32, train-mode BatchNorm), every
launch of the block timed alone with HIP events -- where the block's time goes against SURVEY 8(d)'s floors."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import engine, ops  # noqa: E402
from unet_nested4tiny_objects_keypoints_amd.ops import V  # noqa: E402

B = int(os.environ.get("B", "32"))
HW = int(os.environ.get("HW", "256"))
REPS = int(os.environ.get("REPS", "20"))
C = 32


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / REPS


dev = "cuda"
x = torch.randn(B, HW, HW, 1, device=dev)
w1, b1 = torch.randn(C, 1, 3, 3, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1
w2, b2 = torch.randn(C, C, 3, 3, device=dev) * 0.05, torch.randn(C, device=dev) * 0.1
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
y1, y2, out = (torch.empty(B, HW, HW, C, device=dev) for _ in range(3))
pooled = torch.empty(B, HW // 2, HW // 2, C, device=dev)
idx = torch.empty(B, HW // 2, HW // 2, C, dtype=torch.uint8, device=dev)
blocks = ops.gemm_pixel_blocks(B, HW, HW)
part1, part2 = torch.empty(blocks * C * 2, device=dev), torch.empty(blocks * C * 2, device=dev)
wp1, wp2 = engine.pack_conv_fwd(w1), engine.pack_conv_fwd(w2)


def conv1():
    ops.gemm_fwd(B, HW, HW, 9, [V(x)], [V(y1)], wp1, b1, part1)


def fin(part):
    return ops.bn_finalize(part, blocks, C, B * HW * HW, gamma, beta, 1e-5, 0.1, rm, rv)


conv1()
st1 = fin(part1)


def conv2():
    ops.gemm_fwd(B, HW, HW, 9, [V(y1, scale=st1[2], shift=st1[3], relu=True)], [V(y2)], wp2, b2, part2)


def conv2_plain():
    ops.gemm_fwd(B, HW, HW, 9, [V(y1)], [V(y2)], wp2, b2, None)


def conv2_fold_only():
    ops.gemm_fwd(B, HW, HW, 9, [V(y1, scale=st1[2], shift=st1[3], relu=True)], [V(y2)], wp2, b2, None)


def conv2_stats_only():
    ops.gemm_fwd(B, HW, HW, 9, [V(y1)], [V(y2)], wp2, b2, part2)


conv2()
st2 = fin(part2)


def apply_pool():
    ops.affine_relu_pool(y2, st2[2], st2[3], True, out, pooled, idx)


def block_two_launch_finalize():   # rounds 1-2: a bn_finalize launch behind each convolution
    conv1()
    s1 = fin(part1)
    ops.gemm_fwd(B, HW, HW, 9, [V(y1, scale=s1[2], shift=s1[3], relu=True)], [V(y2)], wp2, b2, part2)
    s2 = fin(part2)
    ops.affine_relu_pool(y2, s2[2], s2[3], True, out, pooled, idx)


rows_f = ops.gemm_stats_rows(B, HW, HW)
partf1, partf2 = torch.empty(rows_f * C * 2, device=dev), torch.empty(rows_f * C * 2, device=dev)


def finish():
    return ops.BatchNormFinish(gamma, beta, rm, rv, 1e-5, 0.1, B * HW * HW)


def conv1_fused():
    f = finish()
    ops.gemm_fwd(B, HW, HW, 9, [V(x)], [V(y1)], wp1, b1, partf1, bn=f)
    return f


def conv2_fused(f1=None):
    f1 = f1 or conv2_fused.f1
    f = finish()
    ops.gemm_fwd(B, HW, HW, 9, [V(y1, scale=f1.scale, shift=f1.shift, relu=True)], [V(y2)], wp2, b2, partf2, bn=f)
    return f


conv2_fused.f1 = conv1_fused()


def block():   # what engine._pair_fwd launches now: per-workgroup rows, finalize enqueued by the convolution call
    f1 = conv1_fused()
    f2 = conv2_fused(f1)
    ops.affine_relu_pool(y2, f2.scale, f2.shift, True, out, pooled, idx)


d_pool, d_act = torch.randn_like(pooled), torch.randn_like(out)


def pool_bwd():
    ops.maxpool_bwd(d_pool, idx, d_act)


rows = [("conv1 (1->32) + stats", conv1), ("bn_finalize", lambda: fin(part1)),
        ("conv2 (32->32) BN-fold + stats", conv2), ("conv2 plain, no stats", conv2_plain),
        ("conv2 BN-fold, no stats", conv2_fold_only), ("conv2 plain + stats", conv2_stats_only),
        ("conv1 + stats + attached finalize", conv1_fused), ("conv2 BN-fold+stats+attached finalize", conv2_fused),
        ("BN-apply + ReLU + pool", apply_pool), ("whole block (attached finalize)", block),
        ("whole block (two finalize launches)", block_two_launch_finalize),
        ("max-pool backward (not in block)", pool_bwd)]
for rep in range(int(os.environ.get("PASSES", "1"))):  # PASSES=2: a second pass shows warm-up / clock drift
    for name, fn in rows:
        us = timeit(fn)
        print("%-34s %8.1f us  %6.2f us/img" % (name, us, us / B))
