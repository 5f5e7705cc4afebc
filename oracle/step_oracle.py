"""CPU oracle for the caller side of the hot path: loss, target synthesis, one step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
  * tools/losses/focal_loss.py:255-301   FocalLoss_BCE_2d (gamma=3, sum / (N*C))
  * tools/misc/helper.py:87-172          create_heatmap (7 keypoints -> 4 channels)
  * trainer/trainer.py:114-136           the body of one training step
"""
from __future__ import annotations

import numpy as np
import torch


def focal_bce_2d_oracle(pred: torch.Tensor, target: torch.Tensor, gamma: float = 3.0) -> torch.Tensor:
    """tools/losses/focal_loss.py:276-301 with size_average=False.

    e = 1 - |p - t| + 1e-20 ; loss = sum(-(1 - e)^gamma * log e) / (N*C)."""
    if pred.dim() > 2:
        pred = pred.view(-1, pred.size(2), pred.size(3))
    target = target.view(-1, target.size(2), target.size(3))
    rows = target.shape[0]
    e = 1 - torch.abs(pred - target) + 1e-20
    return (-1 * (1 - e) ** gamma * torch.log(e)).sum() / rows


def _gauss_of_distance(cx, cy, height, width, radius):
    xs = np.arange(width, dtype=np.float64)[None, :]
    ys = np.arange(height, dtype=np.float64)[:, None]
    dist = np.sqrt((xs - cx) ** 2 + (ys - cy) ** 2)
    return np.exp(-0.5 * dist / radius)


def create_heatmap_oracle(points, height: int, width: int) -> np.ndarray:
    """tools/misc/helper.py:87-172: points [N, C>=5, 2] as (x, y) -> float32 [N,4,H,W].

    ch0 = point 0; ch1 = points 1..3 summed then divided by its max;
    ch2 = point 4; ch3 = points 5..C-1 summed then divided by its max; R = 3."""
    pts = np.asarray(points, dtype=np.float64)
    n, c, _ = pts.shape
    out = np.zeros((n, 4, height, width), dtype=np.float32)
    for b in range(n):
        out[b, 0] = _gauss_of_distance(pts[b, 0, 0], pts[b, 0, 1], height, width, 3)
        for p in range(1, 4):
            out[b, 1] += _gauss_of_distance(pts[b, p, 0], pts[b, p, 1], height, width, 3)
        out[b, 1] = out[b, 1] / np.max(out[b, 1])
        out[b, 2] = _gauss_of_distance(pts[b, 4, 0], pts[b, 4, 1], height, width, 3)
        for p in range(5, c):
            out[b, 3] += _gauss_of_distance(pts[b, p, 0], pts[b, p, 1], height, width, 3)
        out[b, 3] = out[b, 3] / np.max(out[b, 3])
    return out


def train_step_oracle(model, optimizer, inputs, target):
    """trainer/trainer.py:114-136: zero_grad, forward, per-head focal loss,
    mean over heads, backward, optimizer step.  Returns (outputs, loss)."""
    optimizer.zero_grad()
    outputs = model(inputs)
    total = 0
    for out in outputs:
        total = total + focal_bce_2d_oracle(out, target)
    total = 1.0 * total / len(outputs)
    total.backward()
    optimizer.step()
    return outputs, total
