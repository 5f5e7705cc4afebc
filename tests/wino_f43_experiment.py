"""CPU experiment (not a test; not collected): fp32 Winograd F(4x4,3x3) priced through the pinned oracle.

    python tests/wino_f43_experiment.py

Every 3x3 convolution with at least 8 input channels of oracle/unet_nested_oracle.py is replaced by an fp32
transform-domain evaluation (F(2x2,3x3) = what csrc/gemm_wino.hip runs, or F(4x4,3x3) with Lavin's points 0, +-1, +-2,
inf); backward is autograd through the same transforms, i.e. input and weight gradients are taken in the transform
domain as a Winograd dgrad / wgrad kernel would.  Printed: worst relative error (max |a-b| / max |b|, the 1e-4 bar's
measure) of outputs and parameter gradients against (a) the reference's own golden fixture and (b) the float64 oracle
on a base-32 network.  Results: DESIGN.md section 8 (round 4).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle.step_oracle import focal_bce_2d_oracle
from oracle.unet_nested_oracle import UNetNestedOracle
from tests.helpers import GatedReLU, RoutedMaxPool, is_pre_bn_bias, load_golden, rel_err, sub

MATS = {
    2: (torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]]),
        torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]]),
        torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])),
    4: (torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                      [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1.]]),
        torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1.]]),
        torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1.]])),
}


def wino_conv(x, w, b, m):
    """3x3 / pad 1 convolution as F(m x m, 3x3) in x's dtype (fp32): tiles of (m+2)^2 at stride m."""
    BT, G, AT = (t.to(x.dtype) for t in MATS[m])
    n, c, h, wd = x.shape
    k = w.shape[0]
    t = m + 2
    hp, wp = -(-h // m) * m, -(-wd // m) * m
    xp = F.pad(x, (1, 1 + wp - wd, 1, 1 + hp - h))
    tiles = xp.unfold(2, t, m).unfold(3, t, m)                       # [n, c, th, tw, t, t]
    V = torch.einsum("ij,nchwjk,lk->nchwil", BT, tiles, BT)
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)                    # [k, c, t, t]
    M = torch.einsum("kcim,nchwim->nkhwim", U, V)
    Y = torch.einsum("ij,nkhwjl,ml->nkhwim", AT, M, AT)              # [n, k, th, tw, m, m]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, hp, wp)[:, :, :h, :wd]
    return y + b.view(1, -1, 1, 1)


def patch(model, m):
    for mod in model.modules():
        if isinstance(mod, torch.nn.Conv2d) and mod.kernel_size == (3, 3) and mod.in_channels >= 8:
            mod.forward = (lambda x, mod=mod: wino_conv(x, mod.weight, mod.bias, m)) if m else \
                (lambda x, mod=mod: F.conv2d(x, mod.weight, mod.bias, padding=1))


def capture_routing(model):
    """hooks that record every ReLU's gate and every max-pool's winners (2*iy + ix, [b, h/2, w/2, c]) of a forward"""
    gates, pools = [], []

    def relu_hook(mod, inp, out):
        gates.append((out.detach() > 0))

    def pool_hook(mod, inp, out):
        v = inp[0].detach()
        b, c, h, w = v.shape
        win = v.view(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, c, h // 2, w // 2, 4)
        pools.append(win.argmax(-1).permute(0, 2, 3, 1).to(torch.uint8).contiguous())

    hs = [m.register_forward_hook(relu_hook) for m in model.modules() if isinstance(m, torch.nn.ReLU)]
    hs.append(model.maxpool.register_forward_hook(pool_hook))
    return gates, pools, hs


def install_routing(model, gates, pools):
    """backward of `model` through the given gates / winners (forward values stay its own): tests/helpers.py's stand-ins"""
    it = iter(gates)
    for seq in [m for m in model.modules() if isinstance(m, torch.nn.Sequential)]:
        for i, child in enumerate(seq):
            if isinstance(child, (torch.nn.ReLU, GatedReLU)):
                seq[i] = GatedReLU(next(it))
    model.maxpool = RoutedMaxPool(pools)


def run(model, x, target):
    model.zero_grad()
    outs = model(x)
    loss = sum(focal_bce_2d_oracle(o, target.to(o.dtype)) for o in outs) / len(outs)
    loss.backward()
    return [o.detach() for o in outs], {k: p.grad.detach().clone() for k, p in model.named_parameters()}


def report(label, outs, grads, ref_outs, ref_grads, ctor):
    eo = max(rel_err(o, r) for o, r in zip(outs, ref_outs))
    eg = {k: rel_err(grads[k], ref_grads[k]) for k in ref_grads if not is_pre_bn_bias(k, ctor)}
    worst = max(eg.items(), key=lambda kv: kv[1])
    print("  %-42s outputs %.2e   gradients worst %.2e (%s)  median %.2e"
          % (label, eo, worst[1], worst[0], sorted(eg.values())[len(eg) // 2]), flush=True)


def main():
    torch.set_num_threads(8)
    print("golden fixture c1_fs4_64x64_b4_seed0 (base 8; reference outputs and gradients)")
    z, ctor = load_golden("c1_fs4_64x64_b4_seed0")
    model = UNetNestedOracle(**ctor).train()
    model.load_state_dict(sub(z, "state0"))
    model.drop_out.eval()
    x, target = torch.from_numpy(z["x"]), torch.from_numpy(z["target"])
    ref_outs = [torch.from_numpy(z["train_out/%d" % i]) for i in range(3)]
    ref_grads = sub(z, "grad")
    for m, label in ((0, "direct fp32 (oracle)"), (2, "F(2x2,3x3) fp32"), (4, "F(4x4,3x3) fp32")):
        patch(model, m)
        model.train()
        model.drop_out.eval()
        bufs = {k: v.clone() for k, v in model.named_buffers()}
        outs, grads = run(model, x, target)
        model.load_state_dict({**model.state_dict(), **bufs})
        report(label, outs, grads, ref_outs, ref_grads, ctor)

    print("base-32 network, 64 x 64, batch 2, against the float64 oracle (widest K: 128 -> 32, 256 -> 128)")
    torch.manual_seed(3)
    ctor = dict(in_channels=1, n_classes=4, feature_scale=1)
    m64 = UNetNestedOracle(**ctor).double().train()
    m64.drop_out.eval()
    x, target = torch.randn(2, 1, 64, 64), torch.rand(2, 4, 64, 64)
    gates, pools, hooks = capture_routing(m64)
    o64, g64 = run(m64, x.double(), target.double())
    for h in hooks:
        h.remove()
    print("  (backward of the fp32 variants through the float64 forward's ReLU gates and pool winners)")
    m32 = UNetNestedOracle(**ctor).train()
    m32.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in m64.state_dict().items()})
    m32.drop_out.eval()
    state = {k: v.clone() for k, v in m32.state_dict().items()}
    for m, label in ((0, "direct fp32"), (2, "F(2x2,3x3) fp32"), (4, "F(4x4,3x3) fp32")):
        m32.load_state_dict(state)
        patch(m32, m)
        install_routing(m32, gates, pools)
        outs, grads = run(m32, x, target)
        flips = sum(g.flips for g in m32.modules() if isinstance(g, GatedReLU))
        report(label + " [%d gate flips]" % flips, outs, grads, o64, g64, ctor)


if __name__ == "__main__":
    main()
