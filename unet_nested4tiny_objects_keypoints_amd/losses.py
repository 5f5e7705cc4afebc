"""Criterion of the training step.

Mirror of ``FocalLoss_BCE_2d`` (/root/reference/tools/losses/focal_loss.py:255-301) as used by the trainer
(trainer/trainer.py:426-427: gamma=3, size_average=False).  The reference moves the head outputs to the CPU for the
loss every step (trainer/trainer.py:122-135, with a "put on GPU" TODO); here GPU tensors take one fused HIP kernel
that produces the loss value and d loss / d pred together (csrc/caller.hip), so backward is a single
scaling.  CPU tensors (host-side tests of the multi-process logic) take the same arithmetic as plain torch ops.
"""
import torch
from torch import nn


class _FocalBCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, rows, gamma):
        from . import ops
        loss, grad = ops.focal_bce(pred.contiguous(), target.contiguous(), rows, gamma, want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad is not None else None), None, None, None


class FocalLoss_BCE_2d(nn.Module):
    def __init__(self, gamma=3, alpha=0.25, size_average=False):
        super().__init__()
        self.gamma = gamma
        self.alpha = alpha  # kept for signature parity; the reference never uses it in forward
        self.size_average = size_average

    def forward(self, input, target):
        if input.dim() > 2:
            input = input.reshape(-1, input.size(2), input.size(3))
        target = target.reshape(-1, target.size(2), target.size(3))
        samples_num = target.shape[0]
        if (input.is_cuda and target.is_cuda and input.dtype == torch.float32 and target.dtype == torch.float32
                and not target.requires_grad):
            rows = input.numel() if self.size_average else samples_num  # mean over elements / sum over N*C rows
            return _FocalBCEFn.apply(input, target, rows, float(self.gamma))
        error = 1 - torch.abs(input - target) + 1e-20
        loss = -1 * (1 - error) ** self.gamma * torch.log(error)
        if self.size_average:
            return loss.mean()
        return loss.sum() / samples_num

    def mean_over_heads(self, outputs, target):
        """The trainer's loop over the deep-supervision heads (trainer/trainer.py:122-135) -- ``avg = 0; avg = avg +
        criterion(o, target) for o in outputs; avg = 1.0 * avg / len(outputs)`` -- in ONE launch, value and gradients
        together: -> (avg: 0-dim tensor without a graph, [d avg / d output]) or None when the heads do not qualify (CPU
        tensors, other dtypes or layouts, more than 8 heads, a target that needs a gradient): the caller then runs the loop.
        Same bits as the loop (tests/test_gpu_caller.py); about 15 launches of ~5 us fewer per step at three or four heads."""
        from . import _lib, ops
        if not (isinstance(outputs, (tuple, list)) and 2 <= len(outputs) <= _lib.MAX_HEADS):
            return None
        if not (target.is_cuda and target.dtype == torch.float32 and not target.requires_grad and target.dim() == 4):
            return None
        for o in outputs:
            if not (o.is_cuda and o.dtype == torch.float32 and o.shape == target.shape and o.is_contiguous()
                    and o.device == target.device):
                return None
        t = target.contiguous()
        rows = t.numel() if self.size_average else t.shape[0] * t.shape[1]
        loss, grads = ops.focal_bce_heads([o.detach() for o in outputs], t, rows, float(self.gamma))
        return loss[0], grads
