"""Heat maps <-> key points on the device: the reference's ``Heatmap`` helper.

Mirrors the interface of /root/reference/tools/misc/heatmap.py (class ``Heatmap(HeatmapPattern)``: constructor and its
argument checks :21-54, ``create_heatmap`` :203-230, ``extract_points_`` :148-200, ``transfer_points`` :241-263) that the
reference's validation loop drives on the CPU with numpy + OpenCV for every head output (trainer/trainer.py:213-221).
Here the maps never leave the GPU.

**Parity unpinned.**  The reference file imports OpenCV at module level and OpenCV is absent from the build image, so
no vectors could be generated from it; the kernels are checked against oracle/keypoints_oracle.py (a restatement from
the source text and, for the OpenCV calls of region_segment_ (:100-144), from their published algorithms: the 3x3
chamfer ``cv2.distanceTransform`` in fixed point, cores above a tenth of its maximum, ``cv2.connectedComponents`` of the
cores, ``cv2.watershed`` of the binary mask seeded with them).  ``segmentation="watershed"`` (default) is that region
step: blobs that touch through a thin neck are split, a blob without a core is dropped, the frame of the map (first /
last row and column) belongs to no region, as ``cv2.watershed`` leaves it; ``segmentation="components"`` keeps the round-2
stand-in (a region = an 8-connected component of the mask).  The reference's matcher ``match_distmin`` is unfinished and returns ``[]`` (:56-79): ``transfer_points`` returns
the extracted points in peak order instead of an empty tensor.

Known differences of ``"watershed"`` from ``cv2.watershed`` (none of them can be pinned without OpenCV): (1) there is no
level-255 phase -- unknown pixels of a hole narrower than 5 px that is enclosed by core pixels (mask value 0, reachable only
from labelled mask pixels) stay outside the region, where OpenCV floods them with the surrounding label; (2) the flood
runs in synchronous rounds instead of OpenCV's FIFO queue, which can move the boundary line between two touching blobs by
a pixel; (3) the core threshold float32(0.1 * max) assumes the float64 product of NumPy < 2 and OpenCV's fixed-point
(non-IPP) distance path.  Peaks sit inside cores, so (1) and (2) do not move a reported point unless the map's maximum
over a region lies on such a pixel.
"""
from __future__ import annotations

import logging

import torch

from . import ops


class Heatmap:
    def __init__(self, pattern, w, h, radius=3, match_method="match_distmin", segmentation="watershed"):
        # the argument checks of HeatmapPattern.__init__ (heatmap.py:32-49)
        if not isinstance(pattern, list):
            raise TypeError("'pattern' must be list.")
        if len(pattern) > 4:
            logging.warning("Elements of 'pattern' is suggested to be less than 4.")
        for ele in pattern:
            if not isinstance(ele, list):
                raise TypeError("'pattern' must be list with elements as type 'list'. e.g. [[0,1,3],[2,4],[5]]")
        flat = sum(pattern, [])
        if len(set(flat)) != len(flat):
            raise ValueError("Number in 'pattern' must not repeat.")
        if not isinstance(w, int) or not isinstance(h, int):
            raise TypeError("'w'& 'h' must be int")
        if not isinstance(match_method, str):
            raise TypeError("'match_method' must be str")
        self.pattern, self.w, self.h, self.radius = pattern, w, h, radius
        self.match_method = match_method
        if segmentation not in ("watershed", "components"):
            raise ValueError("'segmentation' must be 'watershed' or 'components'")
        self.segmentation = segmentation

    @staticmethod
    def _cuda(t):
        t = torch.as_tensor(t, dtype=torch.float32)
        if not t.is_cuda:
            if not torch.cuda.is_available():
                raise RuntimeError("Heatmap runs on the GPU: this path has no CPU fallback")
            t = t.cuda()
        return t.contiguous()

    def create_heatmap(self, targets) -> torch.Tensor:
        """targets [N, C, 2] as (x, y) -> CUDA float32 [N, len(pattern), H, W]  (heatmap.py:203-230)"""
        return ops.heatmap_pattern(self._cuda(targets), self.pattern, self.h, self.w, float(self.radius))

    def extract_points_(self, pred, num, threshold=0.5):
        """[H, W] heat map -> list of up to `num` [x, y], brightest region first (heatmap.py:148-200)"""
        pred = self._cuda(pred)
        if pred.dim() != 2:
            raise AssertionError("Heatmap assertion failed. It should be [H, W]")
        points, counts = ops.keypoints_extract(pred.unsqueeze(0), int(num), float(threshold), segmentation=self.segmentation)
        n = min(int(counts[0]), int(num))
        # origin size == map size here (the reference rescales by origin/self sizes that are the same numbers, :171-172)
        return [[int(x), int(y)] for x, y in points[0, :n].cpu().tolist()]

    def transfer_points(self, preds, targets=None, threshold=0.5):
        """preds [N, C, H, W] -> (points [N, C, max_num, 2] CUDA float32 as (x, y), -1 padded; counts [N, C] int32 =
        points found per map).  `targets` is accepted for signature compatibility (heatmap.py:241): the reference
        only uses it in its unfinished matcher."""
        preds = self._cuda(preds)
        if preds.dim() != 4:
            raise AssertionError("preds shape should be [N, C, H, W]")
        if preds.shape[1] != len(self.pattern):
            raise AssertionError("2nd dimension of preds must equal the number of maps of the pattern")
        if targets is not None:
            t = torch.as_tensor(targets)
            if t.dim() != 3 or t.shape[2] != 2 or t.shape[0] != preds.shape[0]:
                raise AssertionError("targets shape should be [N, C, 2] with the batch size of preds")
        n, c, h, w = preds.shape
        nums = [len(hmap) for hmap in self.pattern]
        points, counts = ops.keypoints_extract(preds.view(n * c, h, w), max(nums), float(threshold),
                                               segmentation=self.segmentation)
        points, counts = points.view(n, c, max(nums), 2), counts.view(n, c)
        limit = torch.tensor(nums, dtype=torch.int32, device=points.device).view(1, c)
        found = torch.minimum(counts, limit)
        slot = torch.arange(max(nums), device=points.device).view(1, 1, -1)
        points = torch.where((slot < found.unsqueeze(-1)).unsqueeze(-1), points, torch.full_like(points, -1.0))
        return points, found
