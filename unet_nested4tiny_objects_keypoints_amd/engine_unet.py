"""Forward / backward schedule of the reference's classic ``UNet`` (models/unet.py:94-117) over
the same HIP C ABI as the nested network: the conv-BN-ReLU pairs (BatchNorm partial sums in the GEMM epilogue,
BN1-apply + ReLU folded into the second convolution's operand load), the fused apply + ReLU + 2x2 max-pool, the
bilinear x2 (align_corners=True) upsampling, the virtual concatenation ``cat([x2, up(x1)], 1)`` (models/unet.py:78)
read as two K-slices by the consumer GEMM, and the 1x1 + sigmoid head are the kernels of ``engine.py``.

One ``torch.autograd.Function`` covers the network.  Gradient fan-in: every encoder output x1..x4 has two consumers
(the skip concatenation and the max-pool into the next level); the skip contribution is stored by the decoder
block's input-gradient GEMM, the pooled contribution is routed to the window argmax inside the node's BatchNorm
backward (no scatter pass, no memset).

Any H, W >= 16: for sizes that are not multiples of 16 the pooling floors (nn.MaxPool2d, models/unet.py:41) and the
upsampled tensor is zero-padded to its skip partner as the reference does (models/unet.py:63-70) -- pinned by the
reference-generated fixture unet_w8_rgb5_40x56_b2.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List

import torch

from . import engine, ops
from .engine import _PairRec, _Saved, _pair_bwd, _pair_fwd, _plan_of
from .ops import V

_ENC = ("inc", "down1", "down2", "down3", "down4")
_DEC = ("up1", "up2", "up3", "up4")


def _numbered_ns(**kw):
    ns = SimpleNamespace()
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def pair_of(double_conv):
    """A models/unet.py:8-25 double_conv (Sequential indices 0,1 | 3,4) in the shape engine._pair_fwd/_pair_bwd
    expect of a unetConv2: .conv1 / .conv2 holders whose children '0' (conv) and '1' (BatchNorm)."""
    seq = double_conv.conv
    return SimpleNamespace(is_batchnorm=True,
                           conv1=_numbered_ns(**{"0": getattr(seq, "0"), "1": getattr(seq, "1")}),
                           conv2=_numbered_ns(**{"0": getattr(seq, "3"), "1": getattr(seq, "4")}))


def _enc_pair(model, name):
    m = getattr(model, name)
    return pair_of(m.conv if name == "inc" else getattr(m.mpconv, "1"))


def _check_input(model, x):
    if x.dim() != 4:
        raise ValueError("expected NCHW input, got %d dims" % x.dim())
    if not x.is_cuda:
        raise RuntimeError("UNet (HIP) needs its input on the GPU: there is no CPU fallback for this path")
    if x.dtype != torch.float32:
        raise TypeError("expected float32 input, got %s" % x.dtype)
    if x.shape[1] != model.n_channels:
        raise ValueError("expected %d input channels, got %d" % (model.n_channels, x.shape[1]))
    if x.shape[2] < 16 or x.shape[3] < 16:
        raise ValueError("H and W must be at least 16 (four 2x2 poolings), got %dx%d" % (x.shape[2], x.shape[3]))
    if next(model.parameters()).device != x.device:
        raise RuntimeError("model and input are on different devices")


def _forward(model, x, training: bool, save: bool):
    b, _, h0, w0 = x.shape
    x_nhwc = ops.nchw_to_nhwc(x.detach().contiguous())
    enc: List[_PairRec] = []
    inp, h, w = x_nhwc, h0, w0
    for i, name in enumerate(_ENC):  # models/unet.py:106-110
        r = _pair_fwd(_enc_pair(model, name), [V(inp)], b, h, w, training, pool=(i < 4))
        enc.append(r)
        if i < 4:
            inp, h, w = r.pooled, h // 2, w // 2
    dec: List[_PairRec] = []
    interps = []
    low = enc[4].out
    for k, name in enumerate(_DEC):  # :111-114: up(x1 = low, x2 = skip) -> cat([x2, up(x1)]) -> double_conv
        skip = enc[3 - k]
        hs, ws = skip.h, skip.w
        hl, wl = low.shape[1], low.shape[2]
        if (2 * hl, 2 * wl) == (hs, ws):
            interp = torch.empty((b, hs, ws, low.shape[3]), dtype=torch.float32, device=x.device)
            ops.bilinear2x_fwd(low, interp)
        else:
            # sizes that are not multiples of 16: the pooling floored, so the upsampled tensor is a row / column short
            # of its skip partner and the reference zero-pads it (models/unet.py:63-70: diff // 2 before, the rest after).
            # Pure data movement, by PyTorch: a zeroed tensor with the interpolation copied into its window.
            tmp = torch.empty((b, 2 * hl, 2 * wl, low.shape[3]), dtype=torch.float32, device=x.device)
            ops.bilinear2x_fwd(low, tmp)
            interp = torch.zeros((b, hs, ws, low.shape[3]), dtype=torch.float32, device=x.device)
            oy, ox = (hs - 2 * hl) // 2, (ws - 2 * wl) // 2
            interp[:, oy:oy + 2 * hl, ox:ox + 2 * wl].copy_(tmp)
        r = _pair_fwd(pair_of(getattr(model, name).conv), [V(skip.out), V(interp)], b, hs, ws, training, pool=False)
        dec.append(r)
        interps.append(interp)
        low = r.out
    outc = model.outc.conv
    out = torch.empty((b, outc.out_channels, h0, w0), dtype=torch.float32, device=x.device)
    ops.head_fwd(low, outc.weight.detach().view(outc.out_channels, -1), outc.bias.detach(), 0.0, 0, None, out)  # :115-116
    if not save:
        return out, None
    s = _Saved()
    s.x_nhwc, s.enc, s.dec, s.out, s.shape = x_nhwc, enc, dec, out, (b, h0, w0)
    s.low_shapes = [t.shape for t in interps]
    return out, s


def _backward(model, s, d_out, want_input_grad: bool):
    b, h0, w0 = s.shape
    grads: Dict[torch.nn.Parameter, torch.Tensor] = {}
    outc = model.outc.conv
    top = s.dec[3].out
    d_top = torch.empty_like(top)
    dw, db = ops.head_bwd(d_out.contiguous(), s.out, top, outc.weight.detach().view(outc.out_channels, -1), 0.0, 0, None,
                          d_top, False)
    grads[outc.weight], grads[outc.bias] = dw, db
    d_skip = [None] * 4  # gradient of enc[i].out from the skip concatenation, i = 0..3
    d_cur = d_top
    for k in range(3, -1, -1):  # up4 .. up1
        r = s.dec[k]
        skip_i = 3 - k
        d_skip[skip_i] = torch.empty_like(s.enc[skip_i].out)
        d_interp = torch.empty(s.low_shapes[k], dtype=torch.float32, device=d_cur.device)
        _pair_bwd(pair_of(getattr(model, _DEC[k]).conv), r, d_cur, [V(d_skip[skip_i]), V(d_interp)], b, grads)
        low = s.dec[k - 1].out if k > 0 else s.enc[4].out
        d_low = torch.empty_like(low)
        hl, wl, (hs, ws) = low.shape[1], low.shape[2], s.low_shapes[k][1:3]
        if (2 * hl, 2 * wl) != (hs, ws):  # the padding's backward: the gradient of the window the interpolation was copied into
            oy, ox = (hs - 2 * hl) // 2, (ws - 2 * wl) // 2
            d_interp = d_interp[:, oy:oy + 2 * hl, ox:ox + 2 * wl].contiguous()
        ops.bilinear2x_bwd(d_interp, d_low, False)
        d_cur = d_low
    # encoder, deepest first: d_cur is the gradient of x5; every level hands the gradient of its pooled input upwards,
    # where it is routed to the argmax inside that level's BatchNorm backward
    dx_in = None
    pool_grad = None
    for i in range(4, -1, -1):
        r = s.enc[i]
        d_node = d_cur if i == 4 else d_skip[i]
        mine, pool_grad = pool_grad, None
        blk = _enc_pair(model, _ENC[i])
        if i > 0:
            d_pooled = torch.empty_like(s.enc[i - 1].pooled)
            _pair_bwd(blk, r, d_node, [V(d_pooled)], b, grads, pool_grad=mine)
            pool_grad = (d_pooled, s.enc[i - 1].pool_idx)
        elif want_input_grad:
            dx_in = torch.empty_like(s.x_nhwc)
            _pair_bwd(blk, r, d_node, [V(dx_in)], b, grads, pool_grad=mine)
        else:
            _pair_bwd(blk, r, d_node, None, b, grads, pool_grad=mine)
    return grads, dx_in


def _with_plan(model, phase, fn):
    try:
        if not engine.USE_PACK_PLAN:
            return fn()
        plan = _plan_of(model)
        plan.begin(phase)
        ops.set_pack_plan(plan)
        try:
            return fn()
        finally:
            ops.set_pack_plan(None)
    finally:
        engine.flush_batch_counters()  # the BatchNorm batch counters of a training forward, one multi-tensor add


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        out, saved = _with_plan(model, "fwd", lambda: _forward(model, x, model.training, True))
        if getattr(model, "_debug_keep_saved", False):
            model._debug_saved = saved
        ctx.model, ctx.saved, ctx.params = model, saved, params
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        model, saved = ctx.model, ctx.saved
        if saved is None:
            raise RuntimeError("UNet (HIP): backward called twice on the same forward")
        ctx.saved = None
        grads, dx = _with_plan(model, "bwd", lambda: _backward(model, saved, d_out, ctx.needs_input_grad[1]))
        dx_nchw = None if dx is None else ops.nhwc_to_nchw(dx)
        return (None, dx_nchw) + tuple(grads.get(p) if need else None
                                      for p, need in zip(ctx.params, ctx.needs_input_grad[2:]))


def run(model, x):
    """The body of UNet.forward."""
    _check_input(model, x)
    params = tuple(model.parameters())
    track = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if track:
        return _UNetFn.apply(model, x, *params)
    out, _ = _with_plan(model, "fwd", lambda: _forward(model, x, model.training, False))
    return out
