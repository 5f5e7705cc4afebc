"""The X_0,0 block (models/unet.py:220,257-259: conv3x3-BN-ReLU x2 + MaxPool2d(2)) at the BENCHMARKED size of
BASELINE configs[1] -- batch 32, 256 x 256, 1 -> 32 -> 32 channels -- forward AND backward through the engine's own
pair schedule, against a float64 PyTorch statement on the CPU.

Why at this size: a BatchNorm channel sums 2.1 M values here (262 k at the batch-4 whole-network case), the weight
gradient runs its full <= 4096-slab split, and the fp32 CPU library itself loses up to 7e-4 on such sums
(profiles/r2/grad_error_fp32_vs_fp64_c2.txt) -- the comparison has to be against float64.  GPU only.
"""
import copy

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import GatedReLU, RoutedMaxPool, check_flips, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


@pytest.mark.parametrize("b,h,w", [(32, 256, 256)])
def test_x00_block_forward_backward_at_benchmark_batch_vs_float64(dev, b, h, w):
    from unet_nested4tiny_objects_keypoints_amd import engine
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    from unet_nested4tiny_objects_keypoints_amd.unet import unetConv2
    torch.manual_seed(41)
    blk = unetConv2(1, 32, True)
    conv1, bn1, conv2, bn2 = (getattr(blk.conv1, "0"), getattr(blk.conv1, "1"), getattr(blk.conv2, "0"),
                              getattr(blk.conv2, "1"))
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.copy_(1 + 0.1 * torch.randn(32))
            bn.bias.copy_(0.1 * torch.randn(32))
    g = torch.Generator().manual_seed(42)
    x = torch.randn(b, 1, h, w, generator=g)
    d_out = torch.randn(b, 32, h, w, generator=g)            # gradient reaching X_0,0 through the dense skips
    d_pool = torch.randn(b, 32, h // 2, w // 2, generator=g)  # gradient reaching it through the max-pool

    # ---- the HIP path: the engine's own schedule of the pair (what the network's forward / backward run)
    blk_d = copy.deepcopy(blk).to(dev)
    x_nhwc = x.to(dev).view(b, h, w, 1)
    r = engine._pair_fwd(blk_d, [V(x_nhwc)], b, h, w, True, pool=True)
    engine.flush_batch_counters()

    # ---- float64 statement on the CPU (batch statistics, biased variance in the normalisation: SURVEY appendix B).
    # Its BACKWARD uses the ReLU gates and pool winners of the HIP forward (tests/helpers.py: among 1.3e8 gated values a
    # handful sit within fp32 rounding of zero, and with a random upstream gradient every such gate moves a
    # 2-million-term random-walk sum by 1e-3 of its size although both sides are right); the differing gates are
    # counted and must be few and only where the float64 pre-activation itself is rounding noise around zero.
    a1_hip = (r.y1.double() * r.bn1[2].double() + r.bn1[3].double()).float()   # only the sign is used
    relu1 = GatedReLU((a1_hip.permute(0, 3, 1, 2) > 0).cpu())
    del a1_hip
    relu2 = GatedReLU((r.out.permute(0, 3, 1, 2) > 0).cpu())
    pool = RoutedMaxPool([r.pool_idx.cpu()])
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in blk.named_parameters()}
    rm = [torch.zeros(32, dtype=torch.float64) for _ in range(2)]
    rv = [torch.ones(32, dtype=torch.float64) for _ in range(2)]
    y1 = F.conv2d(x.double(), p64["conv1.0.weight"], p64["conv1.0.bias"], padding=1)
    a1 = relu1(F.batch_norm(y1, rm[0], rv[0], p64["conv1.1.weight"], p64["conv1.1.bias"], True, 0.1, 1e-5))
    y2 = F.conv2d(a1, p64["conv2.0.weight"], p64["conv2.0.bias"], padding=1)
    out = relu2(F.batch_norm(y2, rm[1], rv[1], p64["conv2.1.weight"], p64["conv2.1.bias"], True, 0.1, 1e-5))
    pooled = pool(out)
    ((out * d_out.double()).sum() + (pooled * d_pool.double()).sum()).backward()
    want = {k: v.grad for k, v in p64.items()}
    out, pooled, y2 = out.detach(), pooled.detach(), y2.detach()
    del y1, a1
    check_flips([relu1, relu2, pool], "x00-block b%d %dx%d" % (b, h, w))

    assert rel_err(_nchw(r.y2), y2) < TOL
    assert rel_err(_nchw(r.out), out) < TOL
    assert rel_err(_nchw(r.pooled), pooled) < TOL
    blk = blk_d
    for bn, m, v in ((getattr(blk.conv1, "1"), rm[0], rv[0]), (getattr(blk.conv2, "1"), rm[1], rv[1])):
        assert rel_err(bn.running_mean.cpu(), m) < TOL and rel_err(bn.running_var.cpu(), v) < TOL
        assert int(bn.num_batches_tracked) == 1
    grads = {}
    d = d_out.permute(0, 2, 3, 1).contiguous().to(dev)
    dp = d_pool.permute(0, 2, 3, 1).contiguous().to(dev)
    engine._pair_bwd(blk, r, d, None, b, grads, pool_grad=(dp, r.pool_idx))
    torch.cuda.synchronize()
    got = {k: grads[p].cpu() for k, p in blk.named_parameters()}
    report = {}
    for k, wg in want.items():
        if k.endswith(".0.bias"):   # a bias in front of BatchNorm: the exact gradient is zero, both sides hold noise
            scale = float(want[k.replace(".bias", ".weight")].abs().max())
            assert float(got[k].abs().max()) <= 1e-3 * scale, k
            continue
        report[k] = rel_err(got[k], wg)
    print("X_0,0 block at %dx%dx%d, HIP vs float64:" % (b, h, w), {k: "%.2e" % v for k, v in report.items()})
    assert max(report.values()) < TOL, report
