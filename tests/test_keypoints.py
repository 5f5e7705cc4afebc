"""Heat maps <-> key points (/root/reference/tools/misc/heatmap.py).  PARITY UNPINNED: the reference
needs OpenCV, absent here, so there are no reference vectors; the only pin is that the pattern maps of
[[0], [1, 2, 3], [4], [5, 6]] reproduce channels 1 and 3 of tools/misc/helper.py's create_heatmap fixture.

CPU part: the oracle (oracle/keypoints_oracle.py) on synthetic blobs.  GPU part (-m gpu): the HIP kernels through the
C ABI against the oracle -- maps within float32 rounding, extracted points and counts EXACTLY."""
import numpy as np
import pytest
import torch

from oracle.keypoints_oracle import create_heatmap_pattern, extract_points, region_mask, transfer_points
from tests.helpers import load_golden

PATTERN = [[0], [1, 2, 3], [4], [5, 6]]


def test_pattern_maps_reproduce_the_pinned_helper_channels():
    z, _ = load_golden("c1_fs4_64x64_b4_seed0")
    got = create_heatmap_pattern(z["kp/points"], PATTERN, 64, 64)
    ref = z["kp/heatmap"]   # produced by the reference's tools/misc/helper.py:87-172
    np.testing.assert_array_equal(got[:, 1], ref[:, 1])
    np.testing.assert_array_equal(got[:, 3], ref[:, 3])
    # single-point maps: heatmap.py normalises them as well, helper.py does not
    np.testing.assert_allclose(got[:, 0], ref[:, 0] / ref[:, 0].max((1, 2), keepdims=True), rtol=0, atol=1e-7)


def test_oracle_recovers_the_key_points_of_its_own_maps():
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(6, 58, (7,)), rng.uniform(6, 58, (7,))], -1)[None].astype(np.float32)
    pts[0, :, 0] = [8, 20, 40, 56, 30, 12, 50]     # far apart inside every map
    pts[0, :, 1] = [10, 50, 12, 40, 30, 20, 55]
    maps = create_heatmap_pattern(pts, PATTERN, 64, 64)
    found = transfer_points(maps, PATTERN)[0]
    for hmap, got in zip(PATTERN, found):
        assert len(got) == len(hmap)
        want = {(int(round(float(pts[0, p, 0]))), int(round(float(pts[0, p, 1])))) for p in hmap}
        assert {tuple(g) for g in got} == want


def test_oracle_edge_cases():
    z = np.zeros((16, 16), dtype=np.float32)
    assert extract_points(z, 3) == []                                  # nothing, even after the retry
    z[5, 5] = 0.9                                                      # an isolated pixel does not survive the median
    assert extract_points(z, 3) == []
    z[4:7, 4:7] = 0.47                                                 # below 0.5, above 0.45: found by the retry
    z[5, 5] = 0.48
    assert extract_points(z, 3) == [[5, 5]]
    two = np.zeros((16, 24), dtype=np.float32)
    two[2:5, 2:5] = 0.6
    two[3, 3] = 0.7
    two[9:13, 15:19] = 0.8                                             # plateau: first pixel in raster order ...
    assert extract_points(two, 2) == [[16, 9], [3, 3]]                 # ... of the REGION (the median drops the corners)
    assert extract_points(two, 1) == [[16, 9]]                         # brightest region first, (x, y)
    assert region_mask(two, 0.5).sum() > 0


# ------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("pattern,h,w,radius", [(PATTERN, 64, 64, 3), ([[2, 0], [1]], 40, 72, 5), ([[0, 1, 2, 3, 4]], 128, 96, 2.5)])
def test_pattern_maps_hip_vs_oracle(dev, pattern, h, w, radius):
    from unet_nested4tiny_objects_keypoints_amd import Heatmap
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(0, w, (3, 7)), rng.uniform(0, h, (3, 7))], -1).astype(np.float32)
    got = Heatmap(pattern, w, h, radius).create_heatmap(pts)
    want = create_heatmap_pattern(pts, pattern, h, w, radius)
    assert got.shape == want.shape and got.dtype == torch.float32
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=3e-7)   # exp/sqrt in float64 on both sides


@pytest.mark.gpu
def test_pattern_maps_hip_vs_reference_fixture(dev):
    from unet_nested4tiny_objects_keypoints_amd import Heatmap
    z, _ = load_golden("c1_fs4_64x64_b4_seed0")
    got = Heatmap(PATTERN, 64, 64).create_heatmap(z["kp/points"]).cpu().numpy()
    np.testing.assert_allclose(got[:, 1], z["kp/heatmap"][:, 1], rtol=0, atol=3e-7)   # the pinned channels
    np.testing.assert_allclose(got[:, 3], z["kp/heatmap"][:, 3], rtol=0, atol=3e-7)


def _random_blob_maps(seed, n, c, h, w):
    rng = np.random.default_rng(seed)
    maps = np.zeros((n, c, h, w), dtype=np.float32)
    ys, xs = np.mgrid[0:h, 0:w]
    for i in range(n):
        for j in range(c):
            for _ in range(rng.integers(0, 7)):
                cx, cy, r, a = rng.uniform(-2, w + 2), rng.uniform(-2, h + 2), rng.uniform(1.5, 6), rng.uniform(0.3, 1.0)
                maps[i, j] = np.maximum(maps[i, j], a * np.exp(-0.5 * np.sqrt((xs - cx) ** 2 + (ys - cy) ** 2) / r))
            maps[i, j] += rng.uniform(0, 0.05, (h, w)).astype(np.float32)      # noise floor below the thresholds
    maps[0, 0] = 0                                                             # an empty map
    return maps


@pytest.mark.gpu
@pytest.mark.parametrize("seed,h,w", [(2, 64, 64), (3, 48, 80), (4, 128, 128)])
def test_extraction_hip_equals_oracle(dev, seed, h, w):
    """random blobs (touching ones, border ones, plateaus from the max-composition, empty maps, retry path included):
    points, their order and the counts must equal the oracle's exactly"""
    from unet_nested4tiny_objects_keypoints_amd import Heatmap
    pattern = [[0], [1, 2, 3], [4, 5]]
    maps = _random_blob_maps(seed, 3, len(pattern), h, w)
    maps[1, 2] = np.where(maps[1, 2] >= 0.5, 0.47, maps[1, 2])                # only the 0.9 * threshold retry finds these
    hm = Heatmap(pattern, w, h)
    points, found = hm.transfer_points(torch.from_numpy(maps))
    want = transfer_points(maps, pattern)
    points, found = points.cpu().numpy(), found.cpu().numpy()
    for n in range(maps.shape[0]):
        for c, hmap in enumerate(pattern):
            ref = want[n][c]
            assert int(found[n, c]) == len(ref), (n, c, ref, points[n, c])
            assert [[int(x), int(y)] for x, y in points[n, c, :len(ref)]] == ref, (n, c)
            assert (points[n, c, len(ref):] == -1).all()
    # the single-map form of the reference's API
    assert hm.extract_points_(maps[2, 1], 3) == extract_points(maps[2, 1], 3)
    assert hm.extract_points_(np.zeros((h, w), dtype=np.float32), 2) == []


@pytest.mark.gpu
def test_extraction_has_no_region_limit(dev):
    """A speckled map (what an untrained network's sigmoid outputs look like after thresholding) has far more regions
    than the ranking buffer holds (max_regions = 4096).  The reference's extract_points_ has no such limit
    (heatmap.py:148-200): the top `num` regions must still come out exactly, in order -- the selection then runs over
    all roots.  Also: the blob maps with a ranking buffer of only `num` entries (every map takes the uncapped path)."""
    from scipy import ndimage

    from oracle.keypoints_oracle import region_mask
    from unet_nested4tiny_objects_keypoints_amd import ops
    rng = np.random.default_rng(9)
    h, w, num, thr = 512, 512, 7, 0.7
    heat = rng.uniform(0, 1, (2, h, w)).astype(np.float32)
    points, counts = ops.keypoints_extract(torch.from_numpy(heat).to(dev), num, thr)
    points, counts = points.cpu().numpy(), counts.cpu().numpy()
    for m in range(2):
        labels, count = ndimage.label(region_mask(heat[m], thr), structure=np.ones((3, 3), dtype=int))
        assert count > 4096, count   # the case this test is about
        assert int(counts[m]) == count
        hz = np.where(heat[m] < thr, 0, heat[m])
        peaks = ndimage.maximum(hz, labels, index=np.arange(1, count + 1))
        order = sorted(range(count), key=lambda k: (-float(peaks[k]), k))[:num]
        for rank, k in enumerate(order):
            ys, xs = np.where((labels == k + 1) & (hz == peaks[k]))
            want = [0, 0] if peaks[k] == 0 else [int(xs[0]), int(ys[0])]
            assert [int(points[m, rank, 0]), int(points[m, rank, 1])] == want, (m, rank)
    maps = _random_blob_maps(5, 2, 3, 64, 64)
    flat = torch.from_numpy(maps.reshape(-1, 64, 64)).to(dev)
    a_pts, a_cnt = ops.keypoints_extract(flat, 2, 0.5)
    b_pts, b_cnt = ops.keypoints_extract(flat, 2, 0.5, max_regions=2)
    assert int(a_cnt.max()) > 2
    assert torch.equal(a_pts, b_pts) and torch.equal(a_cnt, b_cnt)


@pytest.mark.gpu
def test_heads_to_points_end_to_end(dev):
    """eval forward of the HIP network -> its three head outputs -> key points, all on the device (the reference moves
    every output to the CPU and runs OpenCV per map, trainer/trainer.py:213-221)"""
    from unet_nested4tiny_objects_keypoints_amd import Heatmap, UNet_Nested
    torch.manual_seed(0)
    m = UNet_Nested(in_channels=1, n_classes=4, feature_scale=8).to(dev).eval()
    hm = Heatmap(PATTERN, 32, 32)
    with torch.no_grad():
        outs = m(torch.randn(2, 1, 32, 32, device=dev))
    for o in outs:
        points, found = hm.transfer_points(o, threshold=float(o.mean()))
        assert points.shape == (2, 4, 3, 2) and found.shape == (2, 4) and points.is_cuda
        ref = transfer_points(o.cpu().numpy(), PATTERN, threshold=float(o.mean()))
        for n in range(2):
            for c in range(4):
                k = int(found[n, c])
                assert [[int(x), int(y)] for x, y in points[n, c, :k].cpu().tolist()] == ref[n][c]


def test_heatmap_constructor_checks_mirror_the_reference():
    from unet_nested4tiny_objects_keypoints_amd import Heatmap
    with pytest.raises(TypeError):
        Heatmap("012", 8, 8)
    with pytest.raises(TypeError):
        Heatmap([[0], 1], 8, 8)
    with pytest.raises(ValueError):
        Heatmap([[0, 1], [1]], 8, 8)
    with pytest.raises(TypeError):
        Heatmap([[0]], 8.0, 8)
