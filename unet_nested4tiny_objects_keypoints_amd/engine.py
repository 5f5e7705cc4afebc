"""Forward / backward schedule of UNet_Nested over the HIP C ABI.

One ``torch.autograd.Function`` covers the whole network (models/unet.py:255-300): ``forward`` walks the
nested graph launching HIP kernels and keeps the activations it needs; ``backward`` walks it in reverse
(gradient-ready order of SURVEY.md section 3c: heads, X_03, X_12, X_02, X_21, X_11, X_01, X_30 .. X_00),
summing the dense-skip fan-in directly in the dgrad epilogues (first contribution stores, later ones
accumulate -- no memsets, no concat/split copies).  PyTorch supplies device memory, the stream and the
autograd hand-off to the loss; every FLOP of the path is in ``libunetpp_hip.so``.

Layout: activations NHWC fp32; the dense-skip concatenation ``cat([up, X_i0, .., X_i,j-1], 1)``
(models/unet.py:198-202) is never materialised -- the consumer GEMM reads one view per source.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .ops import V


USE_PACK_PLAN = os.environ.get("UNETPP_NO_PACK_PLAN") is None  # one batched weight-image launch per pass (ops.PackPlan)
# Input gradients of the dense skips grouped by PRODUCER (one GEMM per skip tensor over the concatenated dY of all its
# consumers, written once) instead of by consumer (every consumer read-modify-writes a slice per input view).
USE_GROUPED_DGRAD = os.environ.get("UNETPP_NO_GROUPED_DGRAD") is None
# bf16 storage: encoder blocks whose first-stage tensor is at most this large write a1 = relu(bn(y1)) instead of folding it
# into conv2's load (see _pair_fwd); 0 = never.  Same box, alternating: configs[4] 11.16 -> 10.98 ms per step at 80 MB (11.02
# at 20-40 MB), configs[3] 7.37 -> 7.34-7.35 ms at 20-200 MB
# A/B switch (measurements only): the bias rows / grouped weights by one launch each instead of the plan's copy launch
USE_PLAN_COPIES = os.environ.get("UNETPP_NO_PLAN_COPIES") is None
_A1_MATERIALIZE_BYTES = int(float(os.environ.get("UNETPP_BF16_A1_MAX_MB", "80")) * (1 << 20))
# fp32 storage, the same choice (round 6): the fold costs the Winograd GEMM ~8 % of its matrix-pipe occupancy and the Winograd
# weight gradient ~20 % (profiles/r6/pmc_summary_f32_*: 0.56 / 0.52 against 0.61 / 0.66 on plain views), a pass over a small
# y1 costs less: headline 18.81 -> 18.74 ms at 40-80 MB (levels 2-4), 18.77 at 160, 18.84 at 300 (every level),
# tools/sweep_f32_a1.sh on one box, alternating.  0 = never.
_A1_MATERIALIZE_BYTES_F32 = int(float(os.environ.get("UNETPP_F32_A1_MAX_MB", "80")) * (1 << 20))


# ----------------------------------------------------------------------------- weight re-layouts
def _empty(n, like):
    return torch.empty(n, dtype=torch.float32, device=like.device)


def _packed(w, t, k, n, dstr, sstr, flip=False):
    dst = _empty(t * k * n, w)
    ops.pack_weight(dst, w, t, k, n, dstr, sstr, flip=flip)
    return dst


def pack_conv_fwd(w):
    """[co, ci, k, k] as the GEMM weight [taps][ci][co] (lazily: the fast kernels read the parameter directly)"""
    co, ci, kh, kw = w.shape
    t = kh * kw
    return ops.WSrc(w, t, ci, co, s_t=1, s_k=t, s_n=ci * t,
                    pack=lambda: _packed(w, t, ci, co, (ci * co, co, 1), (1, t, ci * t)))


def pack_conv_dgrad(w):
    """[co, ci, k, k] as [taps (rotated 180)][co][ci]: dx = conv(dy, this)"""
    co, ci, kh, kw = w.shape
    t = kh * kw
    return ops.WSrc(w, t, co, ci, s_t=1, s_k=ci * t, s_n=t, flip=True,
                    pack=lambda: _packed(w, t, co, ci, (co * ci, ci, 1), (1, ci * t, t), flip=True))


def pack_conv_dgrad_slice(w, c_off, c_len):
    """pack_conv_dgrad of the input channels [c_off, c_off + c_len) of w -- read in place (strides of the whole weight)"""
    co, ci, kh, kw = w.shape
    t = kh * kw
    flat = w.reshape(-1)[c_off * t:]   # contiguous 1-D view that starts at channel c_off of output channel 0
    return ops.WSrc(flat, t, co, c_len, s_t=1, s_k=ci * t, s_n=t, flip=True,
                    pack=lambda: _packed(w[:, c_off:c_off + c_len].contiguous(), t, co, c_len, (co * c_len, c_len, 1),
                                         (1, c_len * t, t), flip=True))


def pack_deconv_fwd(w):
    """[ci, co, 2, 2] as [1][ci][4*co], column = (a*2+b)*co + c"""
    ci, co = w.shape[0], w.shape[1]
    return ops.WSrc(w, 1, ci, 4 * co, s_t=0, s_k=4 * co, s_n=4, s_no=1, n_inner=co,
                    pack=lambda: _packed(w, 4, ci, co, (co, 4 * co, 1), (1, 4 * co, 4)))


def pack_deconv_dgrad(w):
    """[ci, co, 2, 2] as [1][4*co][ci], row = (a*2+b)*co + c"""
    ci, co = w.shape[0], w.shape[1]
    return ops.WSrc(w, 1, 4 * co, ci, s_t=0, s_k=4, s_ko=1, k_inner=co, s_n=4 * co,
                    pack=lambda: _packed(w, 4, co, ci, (co * ci, ci, 1), (1, 4, 4 * co)))


def tile_bias4(b):
    """the bias of a 2x2 transposed convolution once per pixel phase (GEMM column = phase * co + c).  Under a pack plan the
    buffer is persistent and all of a pass's rows are written by the plan's one copy launch (ops.PackPlan.copy_for)."""
    co = b.numel()
    plan = ops.current_plan()
    if plan is not None and USE_PLAN_COPIES and b.is_contiguous():
        dst, ready = plan.copy_for(("bias4", b.data_ptr()), plan.phase, (b.data_ptr(), co),
                                   lambda: (_empty(4 * co, b), [b], [(b, 0, 0, 4, co, 0, co)]))
        if ready:
            return dst
    else:
        dst = _empty(4 * co, b)
    ops.pack_weight(dst, b, 4, 1, co, (co, 0, 1), (0, 0, 1))
    return dst


def _phase_views(t, **kw):
    """The four pixel phases (a, b) of a tensor at twice the logical resolution, in t = a*2+b order."""
    return [V(t, sy=2, sx=2, oy=a, ox=b, **kw) for a in (0, 1) for b in (0, 1)]


# ----------------------------------------------------------------------------- records kept for backward
class _PairRec:
    __slots__ = ("ins", "y1", "a1", "y2", "out", "pooled", "pool_idx", "bn1", "bn2", "h", "w")

    def __init__(self):
        for s in self.__slots__:
            setattr(self, s, None)


class _UpRec:
    __slots__ = ("src", "up", "interp", "h", "w")

    def __init__(self):
        for s in self.__slots__:
            setattr(self, s, None)


class _Saved:
    pass


# nn.BatchNorm2d counts its training batches; the counters of a pass are bumped together by one multi-tensor add
# at the end of the forward (eight separate one-element adds cost 5 us of device time each)
_PENDING_COUNTERS: list = []


def flush_batch_counters():
    if _PENDING_COUNTERS:
        torch._foreach_add_(_PENDING_COUNTERS, 1)
        _PENDING_COUNTERS.clear()


def _conv_bn_fwd(ins, conv, bn, y, b, h, w, training):
    """conv3x3 + bias -> y, with the BatchNorm partial sums taken in the GEMM epilogue (training)."""
    co = conv.out_channels
    wp = pack_conv_fwd(conv.weight.detach())
    if training:
        # the statistics are finished by the convolution call itself (ops.BatchNormFinish: the persistent kernels write one
        # row of sums per workgroup -- 512 or 1024 instead of 8192 for a level-0 layer -- and the library enqueues the finalize)
        partial = _empty(ops.gemm_stats_rows(b, h, w) * co * 2, y)
        fin = ops.BatchNormFinish(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps,
                                  bn.momentum, b * h * w)
        ops.gemm_fwd(b, h, w, 9, ins, [V(y)], wp, conv.bias.detach(), partial, bn=fin)
        _PENDING_COUNTERS.append(bn.num_batches_tracked)
        return (fin.mean, fin.invstd, fin.scale, fin.shift)
    ops.gemm_fwd(b, h, w, 9, ins, [V(y)], wp, conv.bias.detach(), None)
    scale, shift = ops.bn_eval_coeffs(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
    return (None, None, scale, shift)


def _pair_fwd(blk, ins: List[V], b, h, w, training, pool, adt=torch.float32) -> _PairRec:
    """models/unet.py:150-156: two [conv3x3 (+BN) + ReLU] stages; optional fused 2x2 max-pool of the result.
    adt = storage type of the activations written (fp32, or bf16 for the UNETPP_GEMM_BF16 kernels)."""
    conv1, conv2 = getattr(blk.conv1, "0"), getattr(blk.conv2, "0")
    co = conv1.out_channels
    like = ins[0].t
    new = lambda: torch.empty((b, h, w, co), dtype=adt, device=like.device)  # noqa: E731
    r = _PairRec()
    r.ins, r.h, r.w = ins, h, w
    if pool:
        r.pooled = torch.empty((b, h // 2, w // 2, co), dtype=adt, device=like.device)
        r.pool_idx = torch.empty((b, h // 2, w // 2, co), dtype=torch.uint8, device=like.device)
    if blk.is_batchnorm:
        bn1, bn2 = getattr(blk.conv1, "1"), getattr(blk.conv2, "1")
        r.y1 = new()
        r.bn1 = _conv_bn_fwd(ins, conv1, bn1, r.y1, b, h, w, training)
        # BatchNorm-apply + ReLU of the first stage is folded into the second convolution's operand load: the
        # normalised activation a1 = relu(y1*scale + shift) is never written (r.a1 stays None).  bf16 storage, small
        # tensors (round 4): a1 IS written -- the same fma and the same rounding the load transform applies, so the
        # operands are bit-identical -- because a plain bf16 view lets conv2 and its weight gradient take the LDS-DMA /
        # quad kernels, and below a few tens of MB the extra pass costs less than the register-staged kernels lose
        r.y2 = new()
        if ((adt == torch.bfloat16 and r.y1.numel() * 2 <= _A1_MATERIALIZE_BYTES) or
                (adt == torch.float32 and r.y1.numel() * 4 <= _A1_MATERIALIZE_BYTES_F32)):
            r.a1 = new()
            ops.affine_relu_pool(r.y1, r.bn1[2], r.bn1[3], True, r.a1, None, None)
            r.bn2 = _conv_bn_fwd([V(r.a1)], conv2, bn2, r.y2, b, h, w, training)
        else:
            r.bn2 = _conv_bn_fwd([V(r.y1, scale=r.bn1[2], shift=r.bn1[3], relu=True)], conv2, bn2, r.y2, b, h, w, training)
        r.out = new()
        ops.affine_relu_pool(r.y2, r.bn2[2], r.bn2[3], True, r.out, r.pooled, r.pool_idx)
    else:
        r.a1 = new()
        ops.gemm_fwd(b, h, w, 9, ins, [V(r.a1, relu=True)], pack_conv_fwd(conv1.weight.detach()), conv1.bias.detach())
        r.out = new()
        ops.gemm_fwd(b, h, w, 9, [V(r.a1)], [V(r.out, relu=True)], pack_conv_fwd(conv2.weight.detach()),
                     conv2.bias.detach())
        if pool:
            ops.affine_relu_pool(r.out, None, None, False, None, r.pooled, r.pool_idx)
    return r


def _up_fwd(upmod, is_deconv, src, b, hs, ws, adt=torch.float32) -> _UpRec:
    """models/unet.py:186-191,199: ConvTranspose2d(2,2) or bilinear x2 (align_corners) + conv1x1; src is [b,hs,ws,ci]."""
    u = _UpRec()
    u.src, u.h, u.w = src, hs, ws
    if is_deconv:
        co = upmod.out_channels
        u.up = torch.empty((b, 2 * hs, 2 * ws, co), dtype=adt, device=src.device)
        ops.gemm_fwd(b, hs, ws, 1, [V(src)], _phase_views(u.up), pack_deconv_fwd(upmod.weight.detach()),
                     tile_bias4(upmod.bias.detach()))
    else:
        conv = getattr(upmod, "1")
        ci, co = conv.in_channels, conv.out_channels
        u.interp = torch.empty((b, 2 * hs, 2 * ws, ci), dtype=adt, device=src.device)
        ops.bilinear2x_fwd(src, u.interp)
        u.up = torch.empty((b, 2 * hs, 2 * ws, co), dtype=adt, device=src.device)
        ops.gemm_fwd(b, 2 * hs, 2 * ws, 1, [V(u.interp)], [V(u.up)], pack_conv_fwd(conv.weight.detach()),
                     conv.bias.detach())
    return u


def _check_input(model, x):
    if x.dim() != 4:
        raise ValueError("expected NCHW input, got %d dims" % x.dim())
    if not x.is_cuda:
        raise RuntimeError("UNet_Nested (HIP) needs its input on the GPU: there is no CPU fallback for this path")
    if x.dtype != torch.float32:
        raise TypeError("expected float32 input, got %s" % x.dtype)
    if x.shape[1] != model.in_channels:
        raise ValueError("expected %d input channels, got %d" % (model.in_channels, x.shape[1]))
    m = 1 << (model.depth - 1)
    if x.shape[2] % m or x.shape[3] % m:
        # the reference fails inside torch.cat for such sizes; fail up front instead
        raise ValueError("H and W must be divisible by %d, got %dx%d" % (m, x.shape[2], x.shape[3]))
    if next(model.parameters()).device != x.device:
        raise RuntimeError("model and input are on different devices")


def _activation_dtype(model):
    """fp32, or bf16 storage (UNet_Nested.set_activation_dtype): BASELINE configs[3]/[4].  The bf16 kernels cover the
    reference's structures (transposed-convolution or bilinear up path, encoder with or without BatchNorm) at channel
    counts 8 * 2^k."""
    adt = getattr(model, "activation_dtype", torch.float32)
    if adt == torch.bfloat16:
        if any(f % 8 or (f // 8) & (f // 8 - 1) for f in model.filters):
            raise NotImplementedError("bf16 storage needs channel counts 8 * 2^k, got %s" % (model.filters,))
    return adt


def _dropout_config(model, training):
    p = float(model.drop_out.p) if (training and model.drop_out.training) else 0.0
    heads = model.depth - 1
    masks = model.dropout_masks if p > 0.0 else None
    if masks is not None and len(masks) != heads:
        raise ValueError("dropout_masks needs one mask per head")
    # A captured training step (graph.GraphedTrainStep) bakes the launch arguments in: the varying part of the seed then
    # lives in a device word that the step's owner rewrites before every replay, and the by-value part stays fixed
    seed_dev = getattr(model, "_dropout_seed_dev", None) if (p > 0.0 and masks is None) else None
    base = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p > 0.0 and masks is None and seed_dev is None) else 0
    seeds = [(base + 0x632BE59BD9B4E019 * (j + 1)) & 0xFFFFFFFFFFFFFFFF for j in range(heads)]
    return p, seeds, masks, seed_dev


def forward_impl(model, x, training: bool, save: bool):
    """Runs the forward DAG of models/unet.py:255-300.  Returns (outputs, saved-for-backward or None)."""
    _check_input(model, x)
    _PENDING_COUNTERS.clear()
    try:
        if not USE_PACK_PLAN:
            return _forward_impl(model, x, training, save)
        plan = _plan_of(model)
        plan.begin("fwd")  # every weight image of the pass in one launch (after the first pass recorded the jobs)
        ops.set_pack_plan(plan)
        try:
            return _forward_impl(model, x, training, save)
        finally:
            ops.set_pack_plan(None)
    finally:
        flush_batch_counters()


def _plan_of(model) -> "ops.PackPlan":
    plan = model.__dict__.get("_pack_plan")
    if plan is None:
        plan = ops.PackPlan()
        model.__dict__["_pack_plan"] = plan
    return plan


def _forward_impl(model, x, training: bool, save: bool):
    b, _, h0, w0 = x.shape
    d = model.depth
    adt = _activation_dtype(model)
    x_nhwc = ops.nchw_to_nhwc(x.detach().contiguous())
    X: Dict[Tuple[int, int], torch.Tensor] = {}
    pairs: Dict[Tuple[int, int], _PairRec] = {}
    ups: Dict[Tuple[int, int], _UpRec] = {}
    inp, h, w = x_nhwc, h0, w0
    for i in range(d):  # encoder column (:257-265)
        with ops.region("X%d0.fwd" % i):
            r = _pair_fwd(getattr(model, "conv%d0" % i), [V(inp)], b, h, w, training, pool=(i < d - 1), adt=adt)
        pairs[(i, 0)], X[(i, 0)] = r, r.out
        if i < d - 1:
            inp, h, w = r.pooled, h // 2, w // 2
    for j in range(1, d):  # decoder columns (:268-280)
        for i in range(d - j):
            mod = getattr(model, "up_concat%d%d" % (i, j))
            hi, wi = h0 >> i, w0 >> i
            u = _up_fwd(mod.up, model.is_deconv, X[(i + 1, j - 1)], b, hi // 2, wi // 2, adt)
            ins = [V(u.up)] + [V(X[(i, jj)]) for jj in range(j)]  # up first, then X_i0.. (:198-202)
            r = _pair_fwd(mod.conv, ins, b, hi, wi, training, pool=False, adt=adt)
            ups[(i, j)], pairs[(i, j)], X[(i, j)] = u, r, r.out
    p_drop, seeds, masks, seed_dev = _dropout_config(model, training)
    outs = []
    for j in range(1, d):  # heads (:283-286)
        head = getattr(model, "final_%d" % j)
        o = torch.empty((b, model.n_classes, h0, w0), dtype=torch.float32, device=x.device)
        ops.head_fwd(X[(0, j)], head.weight.detach().view(model.n_classes, -1), head.bias.detach(), p_drop,
                     seeds[j - 1], None if masks is None else masks[j - 1], o, seed_dev=seed_dev)
        outs.append(o)
    if not save:
        return outs, None
    s = _Saved()
    s.x_nhwc, s.X, s.pairs, s.ups, s.outs = x_nhwc, X, pairs, ups, outs
    s.p_drop, s.seeds, s.masks, s.seed_dev = p_drop, seeds, masks, seed_dev
    s.shape = (b, h0, w0)
    return outs, s


# ----------------------------------------------------------------------------- backward pieces
class _GradBook:
    """Gradient buffers of the node outputs; tracks whether a buffer already holds a contribution."""

    def __init__(self, X, depth, gate_keys, grouped=False):
        self.X = X
        self.buf: Dict[Tuple[int, int], torch.Tensor] = {}
        self.got: Dict[Tuple[int, int], int] = {}
        self.gate_keys = gate_keys   # nodes whose ReLU mask may be applied by their last contributor
        self.gated = set()
        # how many gradient contributions every node output receives (fan-in table of SURVEY.md 3c)
        self.expected = {}
        for (i, j) in X:
            n = (1 if (i == 0 and j >= 1) else 0)            # its deep-supervision head
            consumers = max(0, (depth - 1 - i) - j)           # concat input of X[i][j'] for j' > j
            n += min(consumers, 1) if grouped else consumers  # (grouped: ONE launch for all of them)
            n += 1 if i >= 1 else 0                           # upsampled into X[i-1][j+1]
            n += 1 if (j == 0 and i < depth - 1) else 0       # max-pooled into X[i+1][0]
            self.expected[(i, j)] = n

    def target(self, key, can_gate=False):
        """(tensor, accumulate?, gate) for the next contribution to d X[key].  `gate` is X[key] when this is
        the node's LAST contribution, the node is gate-eligible and the contributor can apply it
        (gate_sum semantics: mask the accumulated sum); the consumer then reads an already-masked gradient."""
        self.got[key] = self.got.get(key, 0) + 1
        gate = None
        if can_gate and key in self.gate_keys and self.got[key] == self.expected[key]:
            gate = self.X[key]
            self.gated.add(key)
        if key in self.buf:
            return self.buf[key], True, gate
        t = torch.empty_like(self.X[key])
        self.buf[key] = t
        return t, False, gate

    def take(self, key):
        assert self.got.get(key, 0) == self.expected[key], "gradient fan-in mismatch at node %s" % (key,)
        return self.buf.pop(key), key in self.gated


_ALLOC = [None]  # set for the duration of one backward: param -> gradient buffer (data-parallel flat buffer)


def _new_grad(p):
    return torch.empty_like(p) if _ALLOC[0] is None else _ALLOC[0](p)


def _conv_wgrad(conv, xs, dys, b, h, w, grads):
    co, ci, kh, kw = conv.weight.shape
    t = kh * kw
    dw = _new_grad(conv.weight)
    db = _new_grad(conv.bias)
    ops.wgrad(b, h, w, t, xs, dys, dw, (1, t, ci * t, 0), db)
    grads[conv.weight] = dw
    grads[conv.bias] = db


def _pair_bwd(blk, r: _PairRec, d_out, in_targets: Optional[List[V]], b, grads, pre_gated=False, pool_grad=None,
              flush=None, in_slice: Optional[Tuple[int, int]] = None):
    """Backward of models/unet.py:150-156.  d_out (gradient of r.out) is consumed.  in_targets: output views
    for the gradient of every entry of r.ins (None = the inputs need no gradient).  pre_gated: the last
    contributor already multiplied d_out by (r.out > 0).  pool_grad = (d_pooled, pool_idx): gradient of the max-pooled
    copy of r.out that has NOT been added to d_out yet (BatchNorm nodes only: routed inside BatchNorm backward).
    flush(): reports the parameter gradients finished so far to the data-parallel averager -- called after each
    convolution's weight gradient, so the deepest node's 3.5 MB (configs[1]) are two buckets rather than one.
    in_slice = (first channel, channels): in_targets cover only that slice of conv1's input channels (grouped input
    gradients: the caller computes the rest from the returned dy1).  Returns dy1, the gradient of conv1's output."""
    conv1, conv2 = getattr(blk.conv1, "0"), getattr(blk.conv2, "0")
    h, w = r.h, r.w
    if blk.is_batchnorm:
        bn1, bn2 = getattr(blk.conv1, "1"), getattr(blk.conv2, "1")
        mean, invstd, scale, shift = r.bn2
        dg, dbt = ops.bn_backward(d_out, r.y2, scale, shift, mean, invstd, bn2.weight.detach(), d_out,
                                  _new_grad(bn2.weight), _new_grad(bn2.bias), pool=pool_grad)
        grads[bn2.weight], grads[bn2.bias] = dg, dbt
        dy2 = V(d_out)
        mean, invstd, scale, shift = r.bn1
        if r.a1 is not None:   # bf16 storage, small tensors: a1 was materialised in the forward pass
            _conv_wgrad(conv2, [V(r.a1)], [dy2], b, h, w, grads)
        else:
            _conv_wgrad(conv2, [V(r.y1, scale=scale, shift=shift, relu=True)], [dy2], b, h, w, grads)  # a1 on the fly
        if flush is not None:
            flush()
        d_a1 = torch.empty_like(r.y1)
        ops.gemm_fwd(b, h, w, 9, [dy2], [V(d_a1)], pack_conv_dgrad(conv2.weight.detach()))
        dg, dbt = ops.bn_backward(d_a1, r.y1, scale, shift, mean, invstd, bn1.weight.detach(), d_a1,
                                  _new_grad(bn1.weight), _new_grad(bn1.bias))
        grads[bn1.weight], grads[bn1.bias] = dg, dbt
        dy1 = V(d_a1)
    else:
        # ReLU backward: already applied by the last contributor's epilogue, else folded into the operand load
        dy2 = V(d_out) if pre_gated else V(d_out, gate=r.out)
        _conv_wgrad(conv2, [V(r.a1)], [dy2], b, h, w, grads)
        if flush is not None:
            flush()
        d_a1 = torch.empty_like(r.a1)
        ops.gemm_fwd(b, h, w, 9, [dy2], [V(d_a1, gate=r.a1)], pack_conv_dgrad(conv2.weight.detach()))
        dy1 = V(d_a1)
    _conv_wgrad(conv1, r.ins, [dy1], b, h, w, grads)
    if flush is not None:
        flush()
    if in_targets is not None:
        if dy1.t.dtype == torch.bfloat16 and in_targets[0].t.dtype == torch.float32:
            # bf16 storage, gradient of the fp32 network input (1..4 channels): the first layer's own VALU kernel
            ops.first_layer_dgrad_bf16(dy1.t, conv1.weight.detach(), in_targets[0].t)
        elif in_slice is not None:
            ops.gemm_fwd(b, h, w, 9, [dy1], in_targets, pack_conv_dgrad_slice(conv1.weight.detach(), in_slice[0], in_slice[1]))
        else:
            ops.gemm_fwd(b, h, w, 9, [dy1], in_targets, pack_conv_dgrad(conv1.weight.detach()))
    return dy1


def _up_bwd(upmod, is_deconv, u: _UpRec, d_up, d_src, accumulate, gate, b, grads):
    hs, ws = u.h, u.w
    if is_deconv:
        ci, co = upmod.weight.shape[0], upmod.weight.shape[1]
        dw = _new_grad(upmod.weight)
        db = _new_grad(upmod.bias)
        ops.wgrad(b, hs, ws, 1, [V(u.src)], _phase_views(d_up), dw, (0, 4 * co, 4, 1), db, n_inner=co)
        grads[upmod.weight], grads[upmod.bias] = dw, db
        ops.gemm_fwd(b, hs, ws, 1, _phase_views(d_up), [V(d_src, accumulate=accumulate, gate=gate, gate_sum=True)],
                     pack_deconv_dgrad(upmod.weight.detach()))
    else:
        conv = getattr(upmod, "1")
        _conv_wgrad(conv, [V(u.interp)], [V(d_up)], b, 2 * hs, 2 * ws, grads)
        d_interp = torch.empty_like(u.interp)
        ops.gemm_fwd(b, 2 * hs, 2 * ws, 1, [V(d_up)], [V(d_interp)], pack_conv_dgrad(conv.weight.detach()))
        ops.bilinear2x_bwd(d_interp, d_src, accumulate, gate)  # (gate: bf16 storage only, see backward_impl)


def _grouped_weights(model, refresh=False, plan=None):
    """{(i, jj): [sum of consumer widths, f_i, 3, 3]}: for every skip tensor X[i][jj] the input-channel slices that read it
    in conv1 of its consumers (i, jj+1), (i, jj+2), ..., concatenated along the OUTPUT-channel axis -- as a convolution
    weight its input gradient is the sum of the consumers' contributions.  Buffers are kept (stable addresses: the
    pack plan prepacks their images with everything else) and refilled from the current parameters when `refresh`:
    by one torch.cat each the first time, from then on by the plan's one copy launch at ``plan.begin("bwd")`` (the slices
    are strided copies: ops.PackPlan.copy_for; 6 launches per step at depth 4, 10 at depth 5 otherwise)."""
    cache = getattr(model, "_grouped_dgrad_w", None)
    d = model.depth
    if cache is None or refresh:
        fresh = {}
        for i in range(d - 1):
            for jj in range(d - 1 - i):
                ws = [getattr(getattr(model, "up_concat%d%d" % (i, jc)).conv.conv1, "0").weight.detach()
                      for jc in range(jj + 1, d - i)]
                f = ws[0].shape[0]
                parts = [w[:, (1 + jj) * f:(2 + jj) * f] for w in ws]
                old = None if cache is None else cache.get((i, jj))
                if old is not None and not (old.device == parts[0].device and old.dtype == parts[0].dtype):
                    old = None
                if plan is not None and all(w.is_contiguous() and w.dtype == torch.float32 for w in ws):
                    key, sig = ("grouped", i, jj), tuple(w.data_ptr() for w in ws)
                    if old is not None and plan.knows_copy(key, "bwd", sig + (old.data_ptr(),)):
                        plan.copy_for(key, "bwd", sig + (old.data_ptr(),), None)   # (marks the entry used)
                        fresh[(i, jj)] = old
                        continue
                    buf = old if old is not None else torch.empty((len(ws) * f, f) + tuple(ws[0].shape[2:]),
                                                                   dtype=ws[0].dtype, device=ws[0].device)
                    t = ws[0].shape[2] * ws[0].shape[3]
                    items = [(w, (1 + jj) * f * t, q * f * f * t, f, f * t, w.shape[1] * t, f * t) for q, w in enumerate(ws)]
                    plan.copy_for(key, "bwd", sig + (buf.data_ptr(),), lambda: (buf, list(ws), items))
                    old = buf
                if old is not None:
                    torch.cat(parts, 0, out=old)
                    fresh[(i, jj)] = old
                else:
                    fresh[(i, jj)] = torch.cat(parts, 0).contiguous()
        cache = fresh
        model._grouped_dgrad_w = cache
    return cache


def backward_impl(model, s: _Saved, d_outs, want_input_grad: bool, grad_sink=None):
    """Returns (dict param -> grad, dx NHWC or None).  grad_sink(list of (param, grad)) is called each time a
    node's parameter gradients are final (used by the data-parallel bucketed all-reduce)."""
    if USE_GROUPED_DGRAD:   # before the pack plan packs this pass's weight images
        _grouped_weights(model, refresh=True, plan=_plan_of(model) if (USE_PACK_PLAN and USE_PLAN_COPIES) else None)
    if not USE_PACK_PLAN:
        return _backward_impl(model, s, d_outs, want_input_grad, grad_sink)
    plan = _plan_of(model)
    plan.begin("bwd")
    ops.set_pack_plan(plan)
    try:
        return _backward_impl(model, s, d_outs, want_input_grad, grad_sink)
    finally:
        ops.set_pack_plan(None)


def _backward_impl(model, s: _Saved, d_outs, want_input_grad: bool, grad_sink=None):
    b, h0, w0 = s.shape
    d = model.depth
    grads: Dict[torch.nn.Parameter, torch.Tensor] = {}
    # Nodes without BatchNorm end in a plain ReLU whose mask the last gradient contributor can apply for free
    # (decoder nodes always; encoder nodes only when is_batchnorm=False -- with BN the mask lives in BN backward).
    gate_keys = {k for k in s.X if k[1] >= 1 or not model.is_batchnorm}
    grouped = USE_GROUPED_DGRAD
    book = _GradBook(s.X, d, gate_keys, grouped)
    bf16 = s.X[(0, 0)].dtype == torch.bfloat16
    dy1s: Dict[Tuple[int, int], V] = {}   # grouped input gradients: dY of conv1 of the decoder nodes done so far
    seen = set()

    def flush():
        if grad_sink is not None and len(grads) > len(seen):
            fresh = [(p, g) for p, g in grads.items() if id(p) not in seen]
            seen.update(id(p) for p, _ in fresh)
            if fresh:
                grad_sink(fresh)

    for j in range(d - 1, 0, -1):  # heads: the first contribution to d X[0][j]
        head = getattr(model, "final_%d" % j)
        go = d_outs[j - 1]
        if go is None:
            go = torch.zeros_like(s.outs[j - 1])
        go = go.contiguous()
        dx, acc, gate = book.target((0, j), can_gate=True)
        dw, db = ops.head_bwd(go, s.outs[j - 1], s.X[(0, j)], head.weight.detach().view(model.n_classes, -1), s.p_drop,
                              s.seeds[j - 1], None if s.masks is None else s.masks[j - 1], dx, acc,
                              gate_x=gate is not None, seed_dev=s.seed_dev)
        if _ALLOC[0] is not None:
            gw, gb = _new_grad(head.weight), _new_grad(head.bias)
            gw.copy_(dw)
            gb.copy_(db)
            dw, db = gw, gb
        grads[head.weight], grads[head.bias] = dw, db
    flush()
    for j in range(d - 1, 0, -1):  # decoder columns, last column first
        for i in range(d - 1 - j, -1, -1):
            mod = getattr(model, "up_concat%d%d" % (i, j))
            r, u = s.pairs[(i, j)], s.ups[(i, j)]
            d_out, pre_gated = book.take((i, j))
            d_up = torch.empty_like(u.up)
            targets = [V(d_up)]
            if grouped:
                # this node's own launch only makes the gradient of its upsampled input; node (i, j) is the LAST consumer
                # of X[i][j-1] in backward order, so that tensor's whole skip gradient is due now: one GEMM over the dY of
                # its consumers (i, j), (i, j+1), ... with the matching input-channel slices of their weights (K-concatenated
                # in the order of _grouped_weights), written -- or added to the head / upsampling contribution -- once
                c_up = u.up.shape[3]
                dy1s[(i, j)] = _pair_bwd(mod.conv, r, d_out, targets, b, grads, pre_gated, flush=flush, in_slice=(0, c_up))
                consumers = list(range(j, d - i))
                t, acc, gate = book.target((i, j - 1), can_gate=True)
                ops.gemm_fwd(b, r.h, r.w, 9, [dy1s[(i, jc)] for jc in consumers],
                             [V(t, accumulate=acc, gate=gate, gate_sum=True)],
                             pack_conv_dgrad(_grouped_weights(model)[(i, j - 1)]))
                if j == 1:
                    for jc in consumers:   # level i is done with its dY tensors
                        del dy1s[(i, jc)]
            else:
                for jj in range(j):
                    t, acc, gate = book.target((i, jj), can_gate=True)
                    targets.append(V(t, accumulate=acc, gate=gate, gate_sum=True))
                _pair_bwd(mod.conv, r, d_out, targets, b, grads, pre_gated, flush=flush)
            # the ReLU mask of a BatchNorm-less node is applied by its LAST gradient contribution: the transposed
            # convolution's input-gradient epilogue, or (bf16 storage, whose kernels take no gate on load) the bilinear
            # backward kernel
            t, acc, gate = book.target((i + 1, j - 1), can_gate=model.is_deconv or bf16)
            _up_bwd(mod.up, model.is_deconv, u, d_up, t, acc, gate, b, grads)
            flush()
    dx_in = None
    pool_grad = None  # (d_pooled, pool_idx) of the node below, still to be added to this node's gradient
    for i in range(d - 1, -1, -1):  # encoder column, deepest first
        blk = getattr(model, "conv%d0" % i)
        r = s.pairs[(i, 0)]
        d_out, pre_gated = book.take((i, 0))
        if pool_grad is not None and not model.is_batchnorm:
            # fp32: the ReLU mask follows as a gate on the consumers' loads (is_batchnorm=False never gates encoder nodes
            # early); bf16 storage: the pool gradient is the node's last contribution and its kernel applies the mask
            ops.maxpool_bwd(pool_grad[0], pool_grad[1], d_out, gate=r.out if (bf16 and not pre_gated) else None)
            pre_gated = pre_gated or bf16
            pool_grad = None
        mine, pool_grad = pool_grad, None
        if i > 0:
            d_pooled = torch.empty_like(s.pairs[(i - 1, 0)].pooled)
            _pair_bwd(blk, r, d_out, [V(d_pooled)], b, grads, pre_gated, pool_grad=mine, flush=flush)
            # the pool gradient is the last contribution to the node above: with BatchNorm it is routed to the argmax
            # inside that node's BatchNorm backward, which reads the gradient anyway
            t, acc, _ = book.target((i - 1, 0))  # counts as the node's last contribution (no gate here)
            if not acc:
                t.zero_()
            pool_grad = (d_pooled, s.pairs[(i - 1, 0)].pool_idx)
        elif want_input_grad:
            if bf16 and s.x_nhwc.shape[3] > 4:
                raise NotImplementedError("bf16 storage: the input gradient needs at most 4 input channels")
            dx_in = torch.empty_like(s.x_nhwc)
            _pair_bwd(blk, r, d_out, [V(dx_in)], b, grads, pre_gated, pool_grad=mine, flush=flush)
        else:
            _pair_bwd(blk, r, d_out, None, b, grads, pre_gated, pool_grad=mine, flush=flush)
        flush()
    return grads, dx_in


class _UNetNestedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        outs, saved = forward_impl(model, x, model.training, save=True)
        if getattr(model, "_debug_keep_saved", False):  # test hook: expose the activations kept for backward
            model._debug_saved = saved
        ctx.model, ctx.saved = model, saved
        ctx.params = params
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *d_outs):
        model, saved = ctx.model, ctx.saved
        if saved is None:
            raise RuntimeError("UNet_Nested (HIP): backward called twice on the same forward")
        ctx.saved = None
        sink = getattr(model, "_grad_sink", None)
        _ALLOC[0] = getattr(model, "_grad_alloc", None)
        try:
            grads, dx = backward_impl(model, saved, d_outs, ctx.needs_input_grad[1], sink)
        finally:
            _ALLOC[0] = None
        dx_nchw = None
        if dx is not None:
            dx_nchw = ops.nhwc_to_nchw(dx)
        done = getattr(model, "_grad_done", None)
        if done is not None and done():
            # data parallel: the averager has put (or accumulated) the averaged gradients into p.grad itself; handing
            # the flat-buffer views to autograd as well would let AccumulateGrad alias p.grad with the work buffer
            return (None, dx_nchw) + (None,) * len(ctx.params)
        return (None, dx_nchw) + tuple(grads.get(p) if need else None
                                      for p, need in zip(ctx.params, ctx.needs_input_grad[2:]))


def run(model, x):
    """The body of UNet_Nested.forward."""
    params = tuple(model.parameters())
    track = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if track:
        return _UNetNestedFn.apply(model, x, *params)
    outs, _ = forward_impl(model, x, model.training, save=False)
    return tuple(outs)
