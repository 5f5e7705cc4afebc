"""Data-parallel training: one process per GPU, gradients averaged with bucketed all-reduces (RCCL over xGMI)
that overlap the remaining backward kernels.

Replaces the reference's single-process ``torch.nn.DataParallel`` (trainer/trainer.py:336-340), which
re-broadcasts every parameter each forward and reduce-adds every gradient onto GPU 0.  Here parameters stay
replicated (one broadcast at start-up), BatchNorm statistics stay per replica exactly as under DataParallel,
and the only exchange per step is the gradient average:

  * all parameter gradients live in ONE flat fp32 buffer laid out in gradient-ready order (heads, X_03,
    X_12, X_02, X_21, X_11, X_01, X_30 .. X_00 -- SURVEY.md section 3c), the wgrad kernels write straight into
    it (engine._new_grad), so a bucket is a contiguous slice and needs no packing copy;
  * whenever the engine reports a node's gradients final and the open bucket has reached ``bucket_bytes``,
    the slice is all-reduced on a side stream behind an event; backward keeps launching kernels meanwhile;
  * the compute stream joins the side stream once, at the end of backward;
  * the averager -- not autograd -- then delivers the gradients: ``p.grad`` becomes the parameter's slice of the flat
    buffer when it was None (the buffers alternate between two homes, so the next backward never writes into a
    live ``p.grad``), and is ADDED to when it already exists (gradient accumulation, ``zero_grad(set_to_none=False)``),
    exactly as autograd's AccumulateGrad would.  The engine returns no parameter gradients to autograd in this mode.

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is bound by one link and a small one by
latency: the default bucket size follows the gradient volume (``auto_bucket_bytes``: about a dozen buckets per step --
0.7 MiB at the 8.8 MB of BASELINE configs[1]/[2], 12 MiB at the 145 MB of configs[4]), few enough that per-collective
latency (tens of microseconds) stays hidden behind the backward kernels, many enough that all but the last bucket
overlap with them.

Contract of the delivered gradients: parameters with ``requires_grad=False`` get none (``p.grad`` stays None, exactly as
autograd would leave them); a handed-out ``p.grad`` is a view into one of two alternating homes and keeps its contents
until the backward AFTER the next one (clone it to keep it longer); tensor hooks / post-accumulate hooks on parameters
never fire on this path (delivery does not go through AccumulateGrad), so registering them is refused, and
``torch.autograd.grad(loss, params)`` is not supported with an attached averager (use ``loss.backward()``).
"""
from __future__ import annotations

import os

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def ready_groups(model) -> List[List[torch.nn.Parameter]]:
    """Parameters in the order engine.backward_impl finishes their gradients, grouped as it REPORTS them (one group per
    engine flush: all heads, then per convolution of a node -- conv2 (+BN2) before conv1 (+BN1) -- and per up-path)."""
    d = model.depth
    out: List[List[torch.nn.Parameter]] = []

    def pair(blk):
        # backward runs conv2 (+BN2) before conv1 (+BN1)
        for name in ("conv2", "conv1"):
            seq = getattr(blk, name)
            grp = []
            if blk.is_batchnorm:
                bn = getattr(seq, "1")
                grp.extend([bn.weight, bn.bias])
            conv = getattr(seq, "0")
            grp.extend([conv.weight, conv.bias])
            out.append(grp)

    heads = []
    for j in range(d - 1, 0, -1):
        head = getattr(model, "final_%d" % j)
        heads.extend([head.weight, head.bias])
    out.append(heads)
    for j in range(d - 1, 0, -1):
        for i in range(d - 1 - j, -1, -1):
            mod = getattr(model, "up_concat%d%d" % (i, j))
            pair(mod.conv)
            up = mod.up if model.is_deconv else getattr(mod.up, "1")
            out.append([up.weight, up.bias])
    for i in range(d - 1, -1, -1):
        pair(getattr(model, "conv%d0" % i))
    return out


def ready_order(model) -> List[torch.nn.Parameter]:
    """Parameters in the order engine.backward_impl finishes their gradients."""
    out = [p for grp in ready_groups(model) for p in grp]
    assert len(out) == len(list(model.parameters())) and len({id(p) for p in out}) == len(out)
    return out


TARGET_BUCKETS = 12


def auto_bucket_bytes(total_bytes: int) -> int:
    """Bucket size for a gradient vector of `total_bytes`: ~TARGET_BUCKETS all-reduces per step (the node-granular
    ready order makes it 8..16 in practice), never below 256 KiB (latency-bound collectives) nor above 64 MiB."""
    return int(min(64 << 20, max(256 << 10, -(-total_bytes // TARGET_BUCKETS))))


class GradientAverager:
    """Flat gradient buffer + bucketed, overlapped all-reduce.  Works on any backend (nccl = RCCL on ROCm;
    gloo on CPU for tests, where the 'side stream' degenerates to in-order execution)."""

    def __init__(self, params: Sequence[torch.nn.Parameter], process_group=None, bucket_bytes: Optional[int] = None,
                 always_reduce: bool = False):
        self.params = list(params)
        self.always_reduce = always_reduce  # issue the collectives even in a world of one (backend smoke tests)
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        dev = self.params[0].device
        self.cuda = dev.type == "cuda"
        total = sum(p.numel() for p in self.params)
        # two homes for the flat gradient vector: backward writes into `flat`; when its slices are handed out as
        # p.grad the other home becomes the work buffer, so a live p.grad is never overwritten by the next backward
        self._homes = [torch.zeros(total, dtype=torch.float32, device=dev), None]
        self._cur = 0
        self.flat = self._homes[0]
        self._avg_op = dist.ReduceOp.SUM
        if dist.get_backend(process_group) == "nccl":
            self._avg_op = dist.ReduceOp.AVG  # RCCL averages in the collective: no divide kernel per bucket
        self.offset = {}
        off = 0
        for p in self.params:
            self.offset[id(p)] = off
            off += p.numel()
        if bucket_bytes is None:
            bucket_bytes = auto_bucket_bytes(4 * total)
        self.bucket_bytes = int(bucket_bytes)
        self.bucket_elems = max(1, self.bucket_bytes // 4)
        self.stream = torch.cuda.Stream(device=dev) if self.cuda else None
        self._ready_upto = 0     # elements of `flat` whose gradients are final
        self._sent_upto = 0      # elements already handed to an all-reduce
        self._pending = set()
        self._works = []
        self.buckets_last_step: List[Tuple[int, int]] = []

    # ---- hooks installed on the model -------------------------------------------------------
    def alloc(self, p):
        o = self.offset[id(p)]
        return self.flat[o:o + p.numel()].view(p.shape)

    def sink(self, fresh):
        """fresh: [(param, grad)] whose gradients just became final (any order within the call)."""
        for p, _ in fresh:
            self._pending.add(id(p))
        # advance the contiguous "ready" frontier along the flat layout
        idx = self._frontier_index
        while idx < len(self.params) and id(self.params[idx]) in self._pending:
            self._ready_upto += self.params[idx].numel()
            idx += 1
        self._frontier_index = idx
        if self._ready_upto - self._sent_upto >= self.bucket_elems:
            self._launch(self._ready_upto)

    def done(self):
        """End of backward: flush the tail bucket, make the compute stream wait for the averages and deliver them
        into ``p.grad``.  Returns True: the engine must not hand the same tensors to autograd as well."""
        if self._frontier_index != len(self.params):
            raise RuntimeError("backward finished but %d parameter gradients were never reported"
                               % (len(self.params) - self._frontier_index))
        if self._sent_upto < self._ready_upto:
            self._launch(self._ready_upto)
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for w in self._works:
                w.wait()
        self._works = []
        self._begin()
        self._deliver()
        return True

    def _check_hooks(self):
        for p in self.params:
            if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
                raise RuntimeError("data-parallel gradient delivery does not run tensor hooks: a parameter of the model has "
                                   "a (post-accumulate-)grad hook registered; remove it or do the work after backward()")

    def _deliver(self):
        self._check_hooks()
        live = [p for p in self.params if p.requires_grad]   # frozen parameters get no gradient (autograd's contract)
        held = [p.grad for p in live]
        if all(g is None for g in held):
            for p in live:
                p.grad = self.alloc(p)
            self._cur ^= 1  # the slices are live now: the next backward works in the other home
            if self._homes[self._cur] is None:
                self._homes[self._cur] = torch.zeros_like(self.flat)
            self.flat = self._homes[self._cur]
            return
        other = self._homes[self._cur ^ 1]
        if (other is not None and len(live) == len(self.params)
                and all(g is not None and g.data_ptr() == other.data_ptr() + 4 * self.offset[id(p)]
                        for p, g in zip(live, held))):
            other.add_(self.flat)  # every p.grad is its slice of the other home: one add for all of them
            return
        for p, g in zip(live, held):
            if g is None:
                p.grad = self.alloc(p).clone()
            else:
                g.add_(self.alloc(p))

    # ---- internals --------------------------------------------------------------------------------
    def _begin(self):
        self._ready_upto = 0
        self._sent_upto = 0
        self._frontier_index = 0
        self._pending = set()

    def _launch(self, upto):
        chunk = self.flat[self._sent_upto:upto]
        self.buckets_last_step.append((self._sent_upto, upto))
        self._sent_upto = upto
        if self.world == 1 and not self.always_reduce:
            return
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())  # the slice's wgrad kernels are enqueued before this
            with torch.cuda.stream(self.stream):
                dist.all_reduce(chunk, op=self._avg_op, group=self.group)
                if self._avg_op == dist.ReduceOp.SUM:
                    chunk.div_(self.world)
        else:
            w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w.wait()
            chunk.div_(self.world)

    def attach(self, model):
        self._begin()
        self.buckets_last_step = []
        model._grad_alloc = self.alloc
        model._grad_sink = self._sink_and_log
        model._grad_done = self.done
        return self

    def _sink_and_log(self, fresh):
        if self._frontier_index == 0 and self._ready_upto == 0:
            self.buckets_last_step = []
        self.sink(fresh)


def broadcast_parameters(model, src: int = 0, process_group=None) -> None:
    """Identical start on every rank (DataParallel's per-forward replicate, done once)."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)


def make_data_parallel(model, process_group=None, bucket_bytes: Optional[int] = None,
                       always_reduce: bool = False, reserved_cus: Optional[int] = None) -> GradientAverager:
    """Broadcast rank 0's parameters/buffers, then hook the gradient averager into the model's backward.

    reserved_cus: CUs the persistent compute grids leave free for the RCCL kernels of the side stream
    (include/unetpp_hip.h, unetpp_set_reserved_cus; without the argument the library's own UNETPP_RESERVED_CUS
    environment variable -- the one variable for this knob -- decides).  Default: none -- the knob exists because the
    overlap of a bucket's all-reduce with the remaining backward is per kernel boundary otherwise;
    UNMEASURED: no multi-GPU node has run it yet, so it is not switched on blindly."""
    if reserved_cus is not None and next(model.parameters()).is_cuda:
        from . import _lib
        _lib.lib().unetpp_set_reserved_cus(int(reserved_cus))
    broadcast_parameters(model, 0, process_group)
    return GradientAverager(ready_order(model), process_group, bucket_bytes, always_reduce).attach(model)
