"""Wall time of the device-side heat map -> key points extraction (tools/misc/heatmap.py:100-200), both
region steps, on maps of trained-network shape (a few blobs per map).  python tools/probes/keypoints_timing.py [H W maps]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from unet_nested4tiny_objects_keypoints_amd import ops  # noqa: E402


def main():
    h, w, maps = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 256, 16)
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:h, 0:w]
    heat = np.zeros((maps, h, w), dtype=np.float32)
    for m in range(maps):
        for _ in range(3):
            x, y = rng.uniform(8, w - 8), rng.uniform(8, h - 8)
            heat[m] = np.maximum(heat[m], np.exp(-0.5 * np.sqrt((xx - x) ** 2 + (yy - y) ** 2) / 3.0))
    t = torch.from_numpy(heat).cuda()
    for seg in ("components", "watershed"):
        ops.keypoints_extract(t, 3, 0.5, segmentation=seg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pts, cnt = ops.keypoints_extract(t, 3, 0.5, segmentation=seg)
        torch.cuda.synchronize()
        print("%-11s %d maps %dx%d: %.3f ms per call, regions per map %s" % (
            seg, maps, h, w, (time.perf_counter() - t0) * 100, cnt.cpu().tolist()[:4]))


if __name__ == "__main__":
    main()
