"""Target synthesis of the training step on the device.

``create_heatmap`` mirrors tools/misc/helper.py:87-172 of the reference (which builds the maps with numpy on the CPU
every step, trainer/trainer.py:122-135): key points ``[N, P >= 6, 2]`` as (x, y) -> float32 ``[N, 4, H, W]``;
channel 0 = point 0, channel 1 = points 1..3 (summed, divided by the maximum), channel 2 = point 4, channel 3 =
points 5..P-1 (summed, divided by the maximum); every map is ``exp(-0.5 * distance / 3)``.
"""
from __future__ import annotations

import torch

from . import ops


def create_heatmap(target, image_height: int, image_width: int, radius: float = 3.0) -> torch.Tensor:
    """`target`: key points, tensor or array-like.  Returns a CUDA float32 tensor (no CPU fallback)."""
    pts = torch.as_tensor(target, dtype=torch.float32)
    if not pts.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("create_heatmap runs on the GPU: this path has no CPU fallback")
        pts = pts.cuda()
    return ops.create_heatmap(pts.contiguous(), int(image_height), int(image_width), radius)
