"""CPU experiment (not a test; not collected): fp32 operands split into two bf16 halves, three bf16 products per fp32 product.

    python tests/bf16x3_experiment.py

The fp32 matrix rate of the MI355X is 1/16 of its bf16 rate.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi) keeps 16
bits of x; a product x * w ~ hi_x hi_w + hi_x lo_w + lo_x hi_w (the lo * lo term, ~2^-16 relative, is dropped) runs as
three v_mfma_f32_*_bf16 with fp32 accumulation at a 16 / 3 = 5.3x higher matrix ceiling than exact fp32.  Whether the
1e-4 bar of BASELINE's north_star survives it is a question about the network, not the kernel: this script prices it
on the CPU before anything is built (the round-5 review's item 7).  Every convolution of the pinned oracle (3x3, 1x1,
transposed: forward, input gradient and weight gradient) is evaluated as three fp32 convolutions of bf16-representable
operands (products of two bf16 values are exact in fp32, the sums are fp32 as in the MFMA accumulators); printed: the
worst relative error (max |a - b| / max |b|) of outputs and parameter gradients against the float64 oracle on a base-32
network, the backward taken through the float64 forward's ReLU gates and pool winners (tests/helpers.py), beside exact fp32.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle.unet_nested_oracle import UNetNestedOracle
from tests.helpers import GatedReLU
from tests.wino_f43_experiment import capture_routing, install_routing, report, run


def hi(t):
    return t.to(torch.bfloat16).to(torch.float32)


def lo(t):
    return (t - hi(t)).to(torch.bfloat16).to(torch.float32)


def three(op, a, b):
    """op(a, b) with both operands split: hi hi + hi lo + lo hi"""
    return op(hi(a), hi(b)) + op(hi(a), lo(b)) + op(lo(a), hi(b))


class SplitConv(torch.autograd.Function):
    """Conv2d (stride 1) or ConvTranspose2d (2x2, stride 2) whose forward, input gradient and weight gradient are ALL taken
    with split operands, as kernels built on bf16 MFMAs would take them."""

    @staticmethod
    def forward(ctx, x, w, b, pad, transposed):
        ctx.save_for_backward(x, w)
        ctx.pad, ctx.transposed = pad, transposed
        if transposed:
            y = three(lambda a, c: F.conv_transpose2d(a, c, None, stride=2), x, w)
        else:
            y = three(lambda a, c: F.conv2d(a, c, None, padding=pad), x, w)
        return y + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if ctx.transposed:
            dx = three(lambda a, c: F.conv2d(a, c, None, stride=2), dy, w)
            dw = three(lambda a, c: torch.nn.grad.conv2d_weight(a, w.shape, c, stride=2), dy, x)   # [ci, co, 2, 2]: x as the "output gradient"
        else:
            dx = three(lambda a, c: torch.nn.grad.conv2d_input(x.shape, c, a, padding=ctx.pad), dy, w)
            dw = three(lambda a, c: torch.nn.grad.conv2d_weight(a, w.shape, c, padding=ctx.pad), x, dy)
        return dx, dw, dy.sum((0, 2, 3)), None, None


def patch(model, split):
    for mod in model.modules():
        if isinstance(mod, torch.nn.Conv2d) and mod.in_channels >= 8:
            pad = mod.padding[0]
            mod.forward = (lambda x, mod=mod, pad=pad: SplitConv.apply(x, mod.weight, mod.bias, pad, False)) if split else \
                (lambda x, mod=mod, pad=pad: F.conv2d(x, mod.weight, mod.bias, padding=pad))
        if isinstance(mod, torch.nn.ConvTranspose2d):
            mod.forward = (lambda x, mod=mod: SplitConv.apply(x, mod.weight, mod.bias, 0, True)) if split else \
                (lambda x, mod=mod: F.conv_transpose2d(x, mod.weight, mod.bias, stride=2))


def main():
    torch.set_num_threads(8)
    print("base-32 network, 64 x 64, batch 2, against the float64 oracle (backward through the float64 forward's routing)")
    torch.manual_seed(3)
    ctor = dict(in_channels=1, n_classes=4, feature_scale=1)
    m64 = UNetNestedOracle(**ctor).double().train()
    m64.drop_out.eval()
    x, target = torch.randn(2, 1, 64, 64), torch.rand(2, 4, 64, 64)
    gates, pools, hooks = capture_routing(m64)
    o64, g64 = run(m64, x.double(), target.double())
    for h in hooks:
        h.remove()
    m32 = UNetNestedOracle(**ctor).train()
    m32.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in m64.state_dict().items()})
    m32.drop_out.eval()
    state = {k: v.clone() for k, v in m32.state_dict().items()}
    for split, label in ((False, "exact fp32 products"), (True, "three bf16 products per fp32 product")):
        m32.load_state_dict(state)
        patch(m32, split)
        install_routing(m32, gates, pools)
        outs, grads = run(m32, x, target)
        flips = sum(g.flips for g in m32.modules() if isinstance(g, GatedReLU))
        report(label + " [%d gate flips]" % flips, outs, grads, o64, g64, ctor)


if __name__ == "__main__":
    main()
