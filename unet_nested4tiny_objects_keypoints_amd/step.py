"""One training step, as the reference's loop body runs it (trainer/trainer.py:114-136).

zero_grad -> forward -> criterion on every deep-supervision head -> mean over heads -> backward ->
optimizer.step().  The loss and the optimizer are PyTorch-ROCm host code; forward and backward of the
network are the HIP path.  Unlike the reference the head outputs are not moved to the CPU for the loss.
"""
from __future__ import annotations

import os

import torch

# A/B switch (measurements only): the written-out loop over the heads instead of the fused loss launch
_FUSED_HEAD_LOSS = os.environ.get("UNETPP_NO_FUSED_HEAD_LOSS") is None


def loss_and_backward(criterion, outputs, target):
    """criterion on every head, mean over heads, backward (the middle of the loop body) -> the mean loss.

    With this package's FocalLoss_BCE_2d on GPU heads the loop, its tensor arithmetic and the autograd walk back to the
    heads are one fused launch (``FocalLoss_BCE_2d.mean_over_heads``: same values bit for bit) and the network's backward is
    started directly from the head gradients; any other criterion runs the loop as written."""
    if isinstance(outputs, tuple):
        fused = getattr(criterion, "mean_over_heads", None)
        if _FUSED_HEAD_LOSS and fused is not None and torch.is_grad_enabled() and all(o.requires_grad for o in outputs):
            got = fused(outputs, target)
            if got is not None:
                avgloss, grads = got
                torch.autograd.backward(list(outputs), grads)
                return avgloss
        avgloss = 0
        for output in outputs:
            avgloss = avgloss + criterion(output, target)
        avgloss = 1.0 * avgloss / len(outputs)
    else:
        avgloss = criterion(outputs, target)
    avgloss.backward()
    return avgloss


def train_step(model, optimizer, criterion, inputs, target):
    optimizer.zero_grad()
    outputs = model(inputs)
    avgloss = loss_and_backward(criterion, outputs, target)
    optimizer.step()
    return outputs, avgloss
