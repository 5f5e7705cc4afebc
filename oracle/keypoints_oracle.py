"""CPU oracle for the heat-map side of validation: pattern-driven target maps and heat map -> key points.

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Parity unpinned**: the reference's implementation lives in
tools/misc/heatmap.py, which imports OpenCV at module level (heatmap.py:13) -- cv2 is absent from this image, so the
file cannot be imported and no vectors can be generated from it.  What is restated, from the source text:

  * ``create_heatmap_pattern``  heatmap.py:203-230 -- for every map of ``pattern`` the Gaussians
    exp(-0.5 * distance / radius) of its key points are summed into a float32 array (one rounding per addition, in
    pattern order) and the map is divided by its maximum.  For the pattern [[0], [1, 2, 3], [4], [5, 6]] the multi-point
    maps equal channels 1 and 3 of tools/misc/helper.py:87-172, which IS pinned (tests/golden kp/heatmap).
  * ``extract_points``  heatmap.py:148-200 (extract_points_) with its region step (region_segment_, :100-144) REPLACED:
    the reference seeds cv2.watershed with the cores of cv2.distanceTransform and so splits blobs that touch; here a
    region is an 8-connected component of the same mask (values below the threshold zeroed, 3x3 median, > 0).  For
    separated blobs -- what the network is trained to produce -- both give one region per blob and the same peak.
    Kept as in the reference: regions ordered by their maximum (descending, stable in label order), the first `num`
    taken, each reported as (x, y) of the first pixel in raster order that attains the maximum; when no region exists
    the extraction is retried once at 0.9 * threshold (heatmap.py:176-198).
  * ``transfer_points``  heatmap.py:241-263 up to the matcher: the reference's match_distmin is unfinished and
    returns [] (heatmap.py:56-79); the points are returned per (image, map) in peak order instead.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage


def create_heatmap_pattern(targets, pattern, height: int, width: int, radius: float = 3.0) -> np.ndarray:
    """targets [N, P, 2] as (x, y) -> float32 [N, len(pattern), H, W]  (heatmap.py:203-230)"""
    pts = np.asarray(targets, dtype=np.float64)
    n = pts.shape[0]
    out = np.zeros((n, len(pattern), height, width), dtype=np.float32)
    xs = np.arange(width, dtype=np.float64)[None, :]
    ys = np.arange(height, dtype=np.float64)[:, None]
    for b in range(n):
        for m, hmap in enumerate(pattern):
            for p in hmap:
                dist = np.sqrt((xs - pts[b, p, 0]) ** 2 + (ys - pts[b, p, 1]) ** 2)
                out[b, m] += np.exp(-0.5 * dist / radius)
            out[b, m] = out[b, m] / np.max(out[b, m])
    return out


def region_mask(pred: np.ndarray, threshold: float) -> np.ndarray:
    """values below the threshold zeroed, 3x3 median with replicated borders (cv2.medianBlur's border mode), > 0"""
    heat = pred.copy()
    heat[heat < threshold] = 0
    return ndimage.median_filter(heat, size=3, mode="nearest") > 0


def _extract_once(pred: np.ndarray, num: int, threshold: float):
    heat = pred.astype(np.float32).copy()
    heat[heat < threshold] = 0
    labels, count = ndimage.label(region_mask(pred.astype(np.float32), threshold), structure=np.ones((3, 3), dtype=int))
    regions = []
    for lab in range(1, count + 1):  # scipy labels in raster order of each component's first pixel
        inside = np.where(labels == lab, heat, 0)
        regions.append((float(inside.max()), lab, inside))
    regions.sort(key=lambda r: r[0], reverse=True)  # stable: ties keep label order (heatmap.py:163)
    points = []
    for peak, _, inside in regions[:num]:
        ys, xs = np.where(inside == peak)  # (a region whose values are all zero reports its first zero, as the reference)
        points.append([int(xs[0]), int(ys[0])])
    return points, count


def extract_points(pred: np.ndarray, num: int, threshold: float = 0.5):
    """[H, W] heat map -> up to `num` [x, y] points, brightest region first"""
    assert pred.ndim == 2, "Heatmap assertion failed. It should be [H, W]"
    points, count = _extract_once(pred, num, threshold)
    if count == 0:
        points, count = _extract_once(pred, num, threshold * 0.9)
    return points


def transfer_points(preds: np.ndarray, pattern, threshold: float = 0.5):
    """[N, C, H, W] -> list (image) of list (map) of points"""
    assert preds.ndim == 4, "preds shape should be [N, C, H, W]"
    return [[extract_points(preds[n, c], len(hmap), threshold) for c, hmap in enumerate(pattern)]
            for n in range(preds.shape[0])]
