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
