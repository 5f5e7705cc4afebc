"""Host-side criterion of the training step (stays PyTorch-ROCm code by the north star).

Mirror of ``FocalLoss_BCE_2d`` (/root/reference/tools/losses/focal_loss.py:255-301) as used by the
trainer (trainer/trainer.py:426-427: gamma=3, size_average=False).  Runs on whatever device its
inputs are on, so the loss stays on the GPU instead of the reference's per-step D2H/H2D crossing
(trainer/trainer.py:122-135); the arithmetic is the same.
"""
import torch
from torch import nn


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
        error = 1 - torch.abs(input - target) + 1e-20
        loss = -1 * (1 - error) ** self.gamma * torch.log(error)
        if self.size_average:
            return loss.mean()
        return loss.sum() / samples_num
