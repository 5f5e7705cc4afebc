"""Heat maps <-> key points (/root/reference/tools/misc/heatmap.py).  PARITY UNPINNED: the reference
needs OpenCV, absent here, so there are no reference vectors; the only pin is that the pattern maps of
[[0], [1, 2, 3], [4], [5, 6]] reproduce channels 1 and 3 of tools/misc/helper.py's create_heatmap fixture.

CPU part: the oracle (oracle/keypoints_oracle.py) on synthetic blobs.  GPU part (-m gpu): the HIP kernels through the
C ABI against the oracle -- maps within float32 rounding, extracted points and counts EXACTLY."""
import numpy as np
import pytest
import torch

from oracle.keypoints_oracle import (DIAG_DIST, HV_DIST, chamfer_distance, create_heatmap_pattern, extract_points, region_markers,
                                     region_mask, transfer_points, watershed_fifo, watershed_regions)
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


def _discs(h, w, blobs):
    """max-composition of exp(-0.5 * distance / radius) blobs [(x, y, radius, height)]"""
    yy, xx = np.mgrid[0:h, 0:w]
    heat = np.zeros((h, w), dtype=np.float64)
    for x, y, r, a in blobs:
        heat = np.maximum(heat, a * np.exp(-0.5 * np.sqrt((xx - x) ** 2 + (yy - y) ** 2) / r))
    return heat.astype(np.float32)


def _touching_maps():
    """the cases the reference's distance-transform + watershed step exists for (heatmap.py:100-144)"""
    bar = np.zeros((48, 72), dtype=np.float32)          # two 24 x 24 squares joined by a two-pixel neck
    bar[10:34, 6:30] = 0.8
    bar[10:34, 38:62] = 0.7
    bar[21:23, 30:38] = 0.6
    bar[15, 12] = 0.95
    bar[22, 45] = 0.9
    corner = np.zeros((56, 56), dtype=np.float32)       # 8-connected through one corner
    corner[4:28, 4:28] = 0.6
    corner[28:52, 28:52] = 0.7
    corner[10, 10] = 0.8
    corner[30, 38] = 0.9
    edge = np.zeros((12, 20), dtype=np.float32)         # survives the median (replicated border), all core, but in the frame
    edge[0, 3:15] = 0.9
    edge[4:9, 19] = 0.8
    return {
        "lines in the frame": (edge, 0.5),
        "squares touching at a corner": (corner, 0.5),
        "two discs that overlap": (_discs(64, 64, [(20, 30, 4.0, 1.0), (34, 30, 4.0, 0.9)]), 0.3),
        "thick blob next to a thin one": (_discs(64, 96, [(30, 30, 8.0, 1.0), (80, 12, 0.9, 0.9)]), 0.4),
        "squares joined by a neck": (bar, 0.5),
        "three in a row": (_discs(40, 120, [(20, 20, 3.0, 1.0), (42, 20, 3.0, 0.8), (64, 20, 3.0, 0.9), (100, 8, 2.0, 0.7)]), 0.2),
        "blob on the border": (_discs(32, 32, [(0, 0, 4.0, 1.0), (31, 16, 3.0, 0.9)]), 0.4),
        "everything above the threshold": (np.full((24, 24), 0.9, dtype=np.float32), 0.5),
    }


def test_chamfer_distance_is_the_fixed_point_shortest_path():
    """the two sequential passes (cv2.distanceTransform, DIST_L2, 3x3) equal the 8-neighbour shortest path with weights
    HV / DIAG to the nearest non-mask pixel, which is what the device kernel relaxes to"""
    rng = np.random.default_rng(3)
    for h, w, p in [(17, 23, 0.8), (32, 32, 0.95), (9, 40, 0.6)]:
        mask = rng.uniform(size=(h, w)) < p
        mask[h // 2, w // 2] = False
        d = chamfer_distance(mask)
        it = np.where(mask, 0x3FFFFFFF, 0).astype(np.int64)
        while True:
            pad = np.pad(it, 1, constant_values=0x3FFFFFFF)
            new = it.copy()
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dy or dx:
                        new = np.minimum(new, pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] + (DIAG_DIST if dy and dx else HV_DIST))
            new = np.where(mask, new, 0)
            if (new == it).all():
                break
            it = new
        np.testing.assert_array_equal(d, it)
    one = np.ones((5, 7), dtype=bool)
    one[2, 1] = False
    d = chamfer_distance(one)
    assert d[2, 2] == HV_DIST and d[3, 2] == DIAG_DIST and d[4, 3] == 2 * DIAG_DIST and d[2, 6] == 5 * HV_DIST
    assert d[0, 4] == 2 * DIAG_DIST + HV_DIST


def test_oracle_splits_touching_blobs_and_drops_coreless_ones():
    maps = _touching_maps()
    heat, thr = maps["two discs that overlap"]
    assert extract_points(heat, 4, thr, "components") == [[20, 30]]
    assert extract_points(heat, 4, thr) == [[20, 30]]       # a wide overlap stays one region: the cores are everything
    heat, thr = maps["squares joined by a neck"]            # deeper than a TENTH of the maximum distance (heatmap.py:121)
    assert extract_points(heat, 4, thr, "components") == [[12, 15]]               # one component ...
    assert extract_points(heat, 4, thr) == [[12, 15], [45, 22]]                   # ... two regions
    heat, thr = maps["squares touching at a corner"]
    assert extract_points(heat, 4, thr, "components") == [[38, 30]]
    assert extract_points(heat, 4, thr) == [[38, 30], [10, 10]]
    heat, thr = maps["thick blob next to a thin one"]
    assert extract_points(heat, 4, thr, "components") == [[30, 30], [80, 12]]
    assert extract_points(heat, 4, thr) == [[30, 30]]                             # thinner than a tenth of the thickest: no core
    heat, thr = maps["three in a row"]
    assert extract_points(heat, 4, thr) == [[20, 20], [64, 20], [42, 20], [100, 8]]
    labels, present = watershed_regions(region_mask(heat, thr))
    assert present == [1, 2, 3, 4] and set(np.unique(labels)) == {0, 1, 2, 3, 4}
    heat, thr = maps["everything above the threshold"]
    assert extract_points(heat, 2, thr, "components") == [[0, 0]]                 # one region, plateau: first pixel
    assert extract_points(heat, 2, thr) == [[1, 1]]                               # cv2.watershed frames the map with -1
    heat, thr = maps["blob on the border"]
    assert extract_points(heat, 2, thr, "components") == [[0, 0], [31, 16]]
    assert extract_points(heat, 2, thr) == [[1, 1], [30, 16]]
    edge = np.zeros((12, 20), dtype=np.float32)                                   # a line along the first row survives the
    edge[0, 3:15] = 0.9                                                           # median (replicated border) and is all core,
    assert region_mask(edge, 0.5)[0, 4:14].all()                                  # but it lies in the frame: no region, and
    assert extract_points(edge, 2, 0.5, "components") == [[4, 0]]                 # the retry finds none either
    assert extract_points(edge, 2, 0.5) == []


def test_synchronous_flood_against_the_sequential_watershed():
    """The device floods in synchronous rounds; cv2.watershed pops one pixel at a time from FIFO queues (restated in
    watershed_fifo).  Same regions up to a few pixels where two fronts meet in the same round -- and the same points."""
    cases = dict(_touching_maps())
    rb = _random_blob_maps(11, 2, 2, 96, 80)
    for i in range(2):
        for j in range(2):
            cases["random blobs %d/%d" % (i, j)] = (rb[i, j], 0.5)
    for name, (heat, thr) in cases.items():
        mask = region_mask(heat, thr)
        labels, present = watershed_regions(mask)
        markers, _ = region_markers(mask)
        flooded = watershed_fifo(mask, markers)
        fifo = np.where(flooded >= 2, flooded - 1, 0)
        assert [int(v) for v in np.unique(fifo[fifo > 0])] == present, name
        differ = int((fifo != labels).sum())
        assert differ <= max(4, int(0.01 * (labels > 0).sum())), (name, differ)
        assert not ((fifo > 0) & (labels > 0) & (fifo != labels)).any(), name    # never a pixel in two different regions
        hz = np.where(heat < thr, 0, heat)
        for lab in present:                                                        # the same peak, at the same pixel
            a, b = np.where(fifo == lab, hz, 0), np.where(labels == lab, hz, 0)
            assert a.max() == b.max() and np.argmax(a) == np.argmax(b), (name, lab)


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
@pytest.mark.parametrize("segmentation", ["watershed", "components"])
@pytest.mark.parametrize("seed,h,w", [(2, 64, 64), (3, 48, 80), (4, 128, 128)])
def test_extraction_hip_equals_oracle(dev, seed, h, w, segmentation):
    """random blobs (touching ones, border ones, plateaus from the max-composition, empty maps, retry path included):
    points, their order and the counts must equal the oracle's exactly"""
    from unet_nested4tiny_objects_keypoints_amd import Heatmap
    pattern = [[0], [1, 2, 3], [4, 5]]
    maps = _random_blob_maps(seed, 3, len(pattern), h, w)
    maps[1, 2] = np.where(maps[1, 2] >= 0.5, 0.47, maps[1, 2])                # only the 0.9 * threshold retry finds these
    hm = Heatmap(pattern, w, h, segmentation=segmentation)
    points, found = hm.transfer_points(torch.from_numpy(maps))
    want = transfer_points(maps, pattern, segmentation=segmentation)
    points, found = points.cpu().numpy(), found.cpu().numpy()
    for n in range(maps.shape[0]):
        for c, hmap in enumerate(pattern):
            ref = want[n][c]
            assert int(found[n, c]) == len(ref), (n, c, ref, points[n, c])
            assert [[int(x), int(y)] for x, y in points[n, c, :len(ref)]] == ref, (n, c)
            assert (points[n, c, len(ref):] == -1).all()
    # the single-map form of the reference's API
    assert hm.extract_points_(maps[2, 1], 3) == extract_points(maps[2, 1], 3, segmentation=segmentation)
    assert hm.extract_points_(np.zeros((h, w), dtype=np.float32), 2) == []


def _same_partition(got, want_labels, present):
    """device labels (root index, -1 outside) describe the oracle's regions (labels `present`, 0 outside), in the same order"""
    roots = np.unique(got[got >= 0])
    assert len(roots) == len(present)
    for lab, r in zip(present, roots):  # oracle labels follow the raster order of the cores' first pixels, as the roots do
        np.testing.assert_array_equal(got == r, want_labels == lab)
    np.testing.assert_array_equal(got < 0, want_labels == 0)


@pytest.mark.gpu
def test_region_step_hip_equals_oracle(dev):
    """the reference's region step on its own (region_segment_, heatmap.py:100-144): chamfer distance EXACTLY (integers),
    then the regions pixel by pixel, on the touching / coreless / border cases and on random blobs; then the points"""
    from unet_nested4tiny_objects_keypoints_amd import ops
    cases = dict(_touching_maps())
    rb = _random_blob_maps(11, 2, 2, 96, 80)
    for i in range(2):
        for j in range(2):
            cases["random blobs %d/%d" % (i, j)] = (rb[i, j], 0.5)
    for name, (heat, thr) in cases.items():
        t = torch.from_numpy(heat).to(dev).unsqueeze(0)
        labels, dist = ops.keypoints_regions(t, thr)
        mask = region_mask(heat, thr)
        np.testing.assert_array_equal(dist[0].cpu().numpy(), chamfer_distance(mask), err_msg=name)
        want_labels, present = watershed_regions(mask)
        _same_partition(labels[0].cpu().numpy(), want_labels, present)
        comp, _ = ops.keypoints_regions(t, thr, segmentation="components")
        from scipy import ndimage
        cl, cc = ndimage.label(mask, structure=np.ones((3, 3), dtype=int))
        _same_partition(comp[0].cpu().numpy(), cl, list(range(1, cc + 1)))
        for seg in ("watershed", "components"):
            points, counts = ops.keypoints_extract(t, 5, thr, segmentation=seg)
            ref = extract_points(heat, 5, thr, seg)
            k = min(int(counts[0]), 5)
            assert [[int(x), int(y)] for x, y in points[0, :k].cpu().tolist()] == ref, (name, seg)
    # all maps of the batch at once give what they give one by one (grid y = map)
    same = [c for c in cases.values() if c[0].shape == (96, 80)]
    batch = torch.from_numpy(np.stack([c[0] for c in same])).to(dev)
    labels, dist = ops.keypoints_regions(batch, 0.5)
    for m, (heat, thr) in enumerate(same):
        one, d1 = ops.keypoints_regions(batch[m:m + 1], 0.5)
        assert torch.equal(one[0], labels[m]) and torch.equal(d1[0], dist[m])


@pytest.mark.gpu
def test_extraction_has_no_region_limit(dev):
    """A speckled map (what an untrained network's sigmoid outputs look like after thresholding) has far more regions
    than the ranking buffer holds (max_regions = 4096).  The reference's extract_points_ has no such limit
    (heatmap.py:148-200): the top `num` regions must still come out exactly, in order -- the selection then runs over
    all roots.  Also: the blob maps with a ranking buffer of only `num` entries (every map takes the uncapped path)."""
    from scipy import ndimage

    from unet_nested4tiny_objects_keypoints_amd import ops
    rng = np.random.default_rng(9)
    h, w, num, thr = 512, 512, 7, 0.7
    heat = rng.uniform(0, 1, (2, h, w)).astype(np.float32)
    w_points, w_counts = ops.keypoints_extract(torch.from_numpy(heat).to(dev), num, thr)     # the reference's region step
    for m in range(2):
        _, present = watershed_regions(region_mask(heat[m], thr))
        assert len(present) > 4096 and int(w_counts[m]) == len(present)
        ref = extract_points(heat[m], num, thr)
        assert [[int(x), int(y)] for x, y in w_points[m].cpu().tolist()] == ref
    points, counts = ops.keypoints_extract(torch.from_numpy(heat).to(dev), num, thr, segmentation="components")
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
