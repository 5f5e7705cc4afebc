"""CPU oracle for the heat-map side of validation: pattern-driven target maps and heat map -> key points.

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Parity unpinned**: the reference's implementation lives in
tools/misc/heatmap.py, which imports OpenCV at module level (heatmap.py:13) -- cv2 is absent from this image, so the
file cannot be imported and no vectors can be generated from it.  What is restated, from the source text:

  * ``create_heatmap_pattern``  heatmap.py:203-230 -- for every map of ``pattern`` the Gaussians
    exp(-0.5 * distance / radius) of its key points are summed into a float32 array (one rounding per addition, in
    pattern order) and the map is divided by its maximum.  For the pattern [[0], [1, 2, 3], [4], [5, 6]] the multi-point
    maps equal channels 1 and 3 of tools/misc/helper.py:87-172, which IS pinned (tests/golden kp/heatmap).
  * ``extract_points``  heatmap.py:148-200 (extract_points_) with its region step (region_segment_, :100-144) restated
    from the published algorithms of the OpenCV calls it makes (``segmentation="watershed"``, the default since round 5):
    cv2.distanceTransform(mask, DIST_L2, 3) is the two-pass 3x3 chamfer transform in 16-bit fixed point (weights
    round(0.955 * 2^16) and round(1.3693 * 2^16), result x 2^-16 as float32); the cores are the pixels above
    float32(0.1 * max); cv2.connectedComponents labels them 8-connected in raster order; cv2.watershed on the BINARY
    mask image floods, at priority 0, every unknown pixel from labelled 4-neighbours of its own value (the cores grow
    through their blob, the background marker through the two-pixel ring that two dilations put around the mask) and
    marks a pixel whose labelled neighbours disagree as a watershed line -- restated here as synchronous breadth-first
    rounds (OpenCV pops one pixel at a time from a FIFO queue: the two differ, if at all, in which of two equidistant
    pixels becomes the line; ``watershed_fifo`` below is the pixel-by-pixel statement, tests/test_keypoints.py compares
    the two: a handful of border pixels per map, the same points).  Consequences the restatement keeps: blobs that touch
    through a neck no deeper than a tenth of the map's largest distance are SPLIT, a blob whose own distance maximum is
    below that tenth has no core and is NOT a region, the frame of the map (first / last row and column) belongs to no
    region, non-core pixels of a blob next to the labelled ring become lines.  ``segmentation="components"`` is the round-2 stand-in (a region = an 8-connected component
    of the mask).  Kept as in the reference: regions ordered by their maximum (descending, stable in label order), the first `num`
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


HV_DIST, DIAG_DIST = 62587, 89738   # cv2 distanceTransform, DIST_L2 with a 3x3 mask: CV_FLT_TO_FIX(0.955, 16), (1.3693, 16)
DIST_INF = 0x3FFFFFFF


def chamfer_distance(mask: np.ndarray) -> np.ndarray:
    """int64 [H, W]: 3x3 chamfer distance (16-bit fixed point) of every mask pixel to the nearest non-mask pixel, by the
    two sequential passes of Borgefors' algorithm as cv2.distanceTransform runs them; 0 outside the mask"""
    h, w = mask.shape
    d = np.full((h + 2, w + 2), DIST_INF, dtype=np.int64)
    d[1:-1, 1:-1] = np.where(mask, DIST_INF, 0)
    for y in range(1, h + 1):
        for x in range(1, w + 1):
            if d[y, x]:
                d[y, x] = min(d[y, x], d[y - 1, x - 1] + DIAG_DIST, d[y - 1, x] + HV_DIST, d[y - 1, x + 1] + DIAG_DIST,
                              d[y, x - 1] + HV_DIST)
    for y in range(h, 0, -1):
        for x in range(w, 0, -1):
            if d[y, x]:
                d[y, x] = min(d[y, x], d[y + 1, x + 1] + DIAG_DIST, d[y + 1, x] + HV_DIST, d[y + 1, x - 1] + DIAG_DIST,
                              d[y, x + 1] + HV_DIST)
    return np.minimum(d[1:-1, 1:-1], DIST_INF)


def watershed_regions(mask: np.ndarray):
    """region_segment_ (heatmap.py:100-144) on the median mask -> (labels int64 [H, W]: core label >= 1 inside a region,
    0 elsewhere; the labels that own at least one pixel, ascending = raster order of each core's first pixel).
    cv2.watershed frames the marker image with boundary pixels (-1) before it floods, so the outermost rows / columns
    of a map carry no label and belong to no region; a core that lies in the frame only does not become a region."""
    h, w = mask.shape
    mk, _ = region_markers(mask)   # 1 = background marker, 0 = unknown
    mk[0, :] = mk[-1, :] = -1
    mk[:, 0] = mk[:, -1] = -1
    pad_m = np.zeros((h + 2, w + 2), dtype=np.int64)
    pad_v = np.full((h + 2, w + 2), -1, dtype=np.int64)   # value outside the image: matches nothing
    pad_v[1:-1, 1:-1] = mask
    while True:
        pad_m[1:-1, 1:-1] = mk
        nb_l = [pad_m[0:-2, 1:-1], pad_m[2:, 1:-1], pad_m[1:-1, 0:-2], pad_m[1:-1, 2:]]
        nb_v = [pad_v[0:-2, 1:-1], pad_v[2:, 1:-1], pad_v[1:-1, 0:-2], pad_v[1:-1, 2:]]
        active = np.zeros((h, w), dtype=bool)
        lo = np.full((h, w), np.iinfo(np.int64).max)
        hi = np.zeros((h, w), dtype=np.int64)
        for l, v in zip(nb_l, nb_v):
            lab = l > 0
            active |= lab & (v == mask)
            lo = np.where(lab, np.minimum(lo, l), lo)
            hi = np.where(lab, np.maximum(hi, l), hi)
        grow = (mk == 0) & active
        if not grow.any():
            break
        mk = np.where(grow, np.where(lo == hi, lo, -1), mk)
    labels = np.where(mk >= 2, mk - 1, 0)
    return labels, [int(v) for v in np.unique(labels[labels > 0])]


def watershed_fifo(mask: np.ndarray, markers: np.ndarray) -> np.ndarray:
    """cv2.watershed on the three-channel image of a binary mask, restated pixel by pixel from its published algorithm
    (Meyer's flooding with one FIFO queue per grey-level difference; OpenCV modules/imgproc/src/segmentation.cpp):
    the frame of the marker image is set to -1; every unlabelled pixel with a labelled 4-neighbour is queued, in raster
    order, under the smallest difference to such a neighbour; then, always from the lowest non-empty queue, a pixel is
    popped, takes the label its labelled neighbours (left, right, top, bottom) agree on or -1 where they differ, and --
    unless it became -1 -- queues its still unlabelled neighbours under their difference to it.  SEQUENTIAL: the slow,
    order-faithful statement the synchronous rounds of ``watershed_regions`` (and of the device kernel) are compared with
    in tests/test_keypoints.py.  markers: > 0 labels, 0 unknown.  Returns the marker image after flooding."""
    from collections import deque
    h, w = mask.shape
    val = np.where(mask, 255, 0).astype(np.int64)
    m = markers.astype(np.int64).copy()
    m[0, :] = m[-1, :] = -1
    m[:, 0] = m[:, -1] = -1
    IN_QUEUE = -2
    queues = [deque() for _ in range(256)]
    nbrs = ((0, -1), (0, 1), (-1, 0), (1, 0))   # left, right, top, bottom
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            if m[y, x] < 0:
                m[y, x] = 0
            if m[y, x] == 0:
                idx = 256
                for dy, dx in nbrs:
                    if m[y + dy, x + dx] > 0:
                        idx = min(idx, abs(int(val[y, x]) - int(val[y + dy, x + dx])))
                if idx < 256:
                    queues[idx].append((y, x))
                    m[y, x] = IN_QUEUE
    active = 0
    while True:
        if not queues[active]:
            active = next((i for i in range(active + 1, 256) if queues[i]), 256)
            if active == 256:
                break
        y, x = queues[active].popleft()
        lab = 0
        for dy, dx in nbrs:
            t = m[y + dy, x + dx]
            if t > 0:
                lab = t if lab == 0 else (lab if t == lab else -1)
        m[y, x] = lab
        if lab == -1:
            continue
        for dy, dx in nbrs:
            if m[y + dy, x + dx] == 0:
                t = abs(int(val[y, x]) - int(val[y + dy, x + dx]))
                queues[t].append((y + dy, x + dx))
                active = min(active, t)
                m[y + dy, x + dx] = IN_QUEUE
    return m


def region_markers(mask: np.ndarray):
    """the marker image region_segment_ hands to cv2.watershed (heatmap.py:117-137): core labels + 1, background 1
    beyond two 3x3 dilations of the mask, 0 (unknown) in between; and the number of cores"""
    dist_f = (chamfer_distance(mask).astype(np.float32) * np.float32(1.0 / 65536.0)).astype(np.float32)
    thr = np.float32(0.1 * float(dist_f.max())) if mask.any() else np.float32(0)
    core = mask & (dist_f > thr)
    core_lab, count = ndimage.label(core, structure=np.ones((3, 3), dtype=int))
    sure_bg = ndimage.binary_dilation(mask, structure=np.ones((3, 3), dtype=bool), iterations=2)
    return np.where(core, core_lab + 1, np.where(sure_bg, 0, 1)).astype(np.int64), count


def _extract_once(pred: np.ndarray, num: int, threshold: float, segmentation: str = "watershed"):
    heat = pred.astype(np.float32).copy()
    heat[heat < threshold] = 0
    mask = region_mask(pred.astype(np.float32), threshold)
    if segmentation == "watershed":
        labels, present = watershed_regions(mask)
    else:
        labels, count = ndimage.label(mask, structure=np.ones((3, 3), dtype=int))
        present = list(range(1, count + 1))
    regions = []
    for lab in present:  # scipy labels in raster order of each component's first pixel (np.unique(markers), heatmap.py:127)
        inside = np.where(labels == lab, heat, 0)
        regions.append((float(inside.max()), lab, inside))
    regions.sort(key=lambda r: r[0], reverse=True)  # stable: ties keep label order (heatmap.py:163)
    points = []
    for peak, _, inside in regions[:num]:
        ys, xs = np.where(inside == peak)  # (a region whose values are all zero reports its first zero, as the reference)
        points.append([int(xs[0]), int(ys[0])])
    return points, len(present)


def extract_points(pred: np.ndarray, num: int, threshold: float = 0.5, segmentation: str = "watershed"):
    """[H, W] heat map -> up to `num` [x, y] points, brightest region first"""
    assert pred.ndim == 2, "Heatmap assertion failed. It should be [H, W]"
    points, count = _extract_once(pred, num, threshold, segmentation)
    if count == 0:
        points, count = _extract_once(pred, num, threshold * 0.9, segmentation)
    return points


def transfer_points(preds: np.ndarray, pattern, threshold: float = 0.5, segmentation: str = "watershed"):
    """[N, C, H, W] -> list (image) of list (map) of points"""
    assert preds.ndim == 4, "preds shape should be [N, C, H, W]"
    return [[extract_points(preds[n, c], len(hmap), threshold, segmentation) for c, hmap in enumerate(pattern)]
            for n in range(preds.shape[0])]
