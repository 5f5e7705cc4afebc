"""CPU oracle for the bf16-storage twins: the nested U-Net oracle evaluated with the ROUNDINGS of the bf16 path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The arithmetic is the oracle's (torch.nn.functional, float64), applied
to the parameters of a ``UNetNestedOracle``; after every tensor the HIP path stores in bf16 the value is rounded to
bf16 (straight-through in backward), and the convolution weights are rounded as the kernels' weight images are:

  * every conv / transposed-conv output (+ bias, + ReLU for the BatchNorm-less decoder blocks) is stored in bf16;
  * BatchNorm statistics are those of the STORED tensor; its apply + ReLU is one fp32 fma, stored (or consumed) in bf16;
  * max-pool, concatenation and the heads' dropout / 1x1 / sigmoid see the stored values; the probabilities are fp32;
  * is_deconv=False: the bilinear interpolation of the stored tensor is stored in bf16, its 1x1 convolution as above;
  * the network's first convolution (1..4 input channels) and the heads use the fp32 weights (VALU kernels).

Backward is plain autograd through this forward: the activation GRADIENTS are not rounded here (the HIP path stores
them in bf16 as well, which moves parameter gradients by ~3e-3 relative: tests/test_gpu_bf16.py states the bar).

``routing`` (optional): the ReLU gate patterns and max-pool winners of another forward (the HIP one).  With 8 mantissa
bits ~0.5 % of the ReLU gates of a layer sit within rounding of zero, and two correct bf16 implementations that differ
in fp32 summation order open different ones; over 20 layers that moves parameter gradients by 10-30 % although every
kernel is right to one rounding.  Given the routing, backward uses THOSE gates and winners (forward values stay this
oracle's own), exactly as tests/helpers.py does for the fp32 parity tests, and the gradients become comparable again.
Follows models/unet.py:121-156,182-202,255-300 exactly as oracle/unet_nested_oracle.py does.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class _GatedReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate):
        ctx.save_for_backward(gate)
        return x.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (gate,) = ctx.saved_tensors
        return g * gate.to(g.dtype), None


class _RoutedPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat_idx):
        ctx.save_for_backward(flat_idx)
        ctx.shape = x.shape
        return F.max_pool2d(x, 2)

    @staticmethod
    def backward(ctx, g):
        (flat_idx,) = ctx.saved_tensors
        b, c, h, w = ctx.shape
        out = torch.zeros(b, c, h * w, dtype=g.dtype)
        out.scatter_(2, flat_idx.view(b, c, -1), g.reshape(b, c, -1))
        return out.view(b, c, h, w), None


def _relu(x, gate, stats):
    if gate is None:
        return torch.relu(x)
    y = _GatedReLU.apply(x, gate)
    stats["gates"] += gate.numel()
    stats["gate_flips"] += int(((y > 0) != gate).sum())
    return y


def _pool(x, idx, stats):
    """idx: uint8 [b, h/2, w/2, c], 2*iy + ix of the winner inside its window (the HIP layout), or None"""
    if idx is None:
        return F.max_pool2d(x, 2)
    b, c, h, w = x.shape
    i = idx.permute(0, 3, 1, 2).long()
    flat = ((2 * torch.arange(h // 2).view(1, 1, -1, 1) + i // 2) * w + 2 * torch.arange(w // 2).view(1, 1, 1, -1) + i % 2)
    y = _RoutedPool.apply(x, flat.contiguous())
    at = x.detach().reshape(b, c, -1).gather(2, flat.view(b, c, -1)).view_as(y)
    stats["windows"] += y.numel()
    stats["pool_flips"] += int((at != y.detach()).sum())
    return y


def rb(t):
    """round to bf16, straight-through gradient"""
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def _bn_relu(y, bn, training, gate, stats):
    """y is the stored (rounded) conv output.  Returns relu(y*scale+shift) rounded, scale/shift in fp32 as the kernels'."""
    if training:
        mean = y.mean((0, 2, 3))
        var = y.var((0, 2, 3), unbiased=False)
    else:
        mean, var = bn.running_mean.to(y.dtype), bn.running_var.to(y.dtype)
    invstd = 1.0 / torch.sqrt(var + bn.eps)
    scale = bn.weight.to(y.dtype) * invstd
    shift = bn.bias.to(y.dtype) - mean * scale
    scale32 = scale + (scale.detach().float().to(y.dtype) - scale.detach())   # the coefficients live in fp32
    shift32 = shift + (shift.detach().float().to(y.dtype) - shift.detach())
    return rb(_relu(y * scale32.view(1, -1, 1, 1) + shift32.view(1, -1, 1, 1), gate, stats))


def _pair(blk, x, with_bn, training, gates, stats):
    for name, gate in zip(("conv1", "conv2"), gates):
        seq = getattr(blk, name)
        conv = seq[0]
        w = conv.weight.to(x.dtype)
        if conv.in_channels > 4:   # the 1..4-channel first convolution runs on the VALU from the fp32 weights and input
            w = rb(w)
        y = F.conv2d(x, w, conv.bias.to(x.dtype), padding=1)
        if with_bn:
            x = _bn_relu(rb(y), seq[1], training, gate, stats)
        else:
            x = rb(_relu(y, gate, stats))
    return x


def routing_of(hip_saved):
    """ReLU gates (NCHW bool, per node: after conv1, after conv2) and pool winners of a HIP forward (engine._Saved)."""
    gates, pools = {}, {}
    for key, rec in hip_saved.pairs.items():
        a1 = rec.a1
        if a1 is None:  # BatchNorm pairs fold BN1-apply + ReLU into the consumer's load: rebuild the sign from y1
            a1 = rec.y1.float() * rec.bn1[2] + rec.bn1[3]
        gates[key] = ((a1.permute(0, 3, 1, 2) > 0).cpu(), (rec.out.permute(0, 3, 1, 2) > 0).cpu())
        if rec.pool_idx is not None:
            pools[key[0]] = rec.pool_idx.cpu()
    return gates, pools


def forward_bf16_sim(model, inputs, training=True, dtype=torch.float64, routing=None, stats=None):
    """model: a UNetNestedOracle.  Dropout must be off (model.drop_out.eval() / p = 0).
    routing = routing_of(hip_saved) or None; stats (dict) receives the counts of differing gates / winners."""
    d = model.depth
    x = inputs.to(dtype)
    gates, pools = routing if routing is not None else ({}, {})
    if stats is None:
        stats = {}
    stats.update(gates=0, gate_flips=0, windows=0, pool_flips=0)
    none2 = (None, None)
    X = [[None] * d for _ in range(d)]
    X[0][0] = _pair(model.conv00, x, model.is_batchnorm, training, gates.get((0, 0), none2), stats)
    for i in range(1, d):
        X[i][0] = _pair(getattr(model, "conv%d0" % i), _pool(X[i - 1][0], pools.get(i - 1), stats), model.is_batchnorm,
                        training, gates.get((i, 0), none2), stats)
    for j in range(1, d):
        for i in range(d - j):
            up = getattr(model, "up_concat%d%d" % (i, j))
            if model.is_deconv:
                u = rb(F.conv_transpose2d(X[i + 1][j - 1], rb(up.up.weight.to(dtype)), up.up.bias.to(dtype), stride=2))
            else:  # bilinear x2 (align_corners) stored in bf16, then the 1x1 convolution on the MFMA (bf16 weights)
                conv = up.up[1]
                interp = rb(F.interpolate(X[i + 1][j - 1], scale_factor=2, mode="bilinear", align_corners=True))
                u = rb(F.conv2d(interp, rb(conv.weight.to(dtype)), conv.bias.to(dtype)))
            X[i][j] = _pair(up.conv, torch.cat([u] + X[i][:j], 1), False, training, gates.get((i, j), none2), stats)
    outs = []
    for j in range(1, d):
        head = getattr(model, "final_%d" % j)
        outs.append(torch.sigmoid(F.conv2d(X[0][j], head.weight.to(dtype), head.bias.to(dtype))))
    return tuple(outs)
