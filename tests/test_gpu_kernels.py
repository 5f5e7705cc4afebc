"""Kernel-level parity: each C-ABI entry point against a CPU fp32/fp64 PyTorch statement of the same op.

GPU only (pytest -m gpu).  Tolerance: 1e-4 relative to the largest reference magnitude (north_star),
most ops land at ~1e-6.
"""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def nhwc(t):  # NCHW cpu -> NHWC gpu
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):  # NHWC gpu -> NCHW cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


@pytest.fixture(params=["winograd", "direct"])
def conv_algo(request):
    """3x3 fast path: Winograd F(2x2,3x3) (default) or direct summation (UNETPP_GEMM_DIRECT)."""
    from unet_nested4tiny_objects_keypoints_amd import ops
    old = ops.USE_WINOGRAD
    ops.USE_WINOGRAD = request.param == "winograd"
    yield request.param
    ops.USE_WINOGRAD = old


@pytest.mark.parametrize("shape", [
    # (B, H, W, [cin per source], cout)
    (2, 16, 16, [8], 8),
    (1, 24, 40, [3], 4),          # rgb first layer, ragged tiles
    (2, 32, 32, [32], 32),
    (1, 64, 64, [32, 32, 32, 32], 32),   # X_03.conv1-like virtual concat
    (2, 8, 8, [16, 8, 8], 40),    # odd column tile
    (1, 4, 4, [5, 3], 7),         # scalar-load path, tile much larger than the image
    (1, 16, 48, [1], 33),
    (2, 32, 64, [1], 32),         # first-layer VALU kernel (1 -> 32 channels)
    (1, 24, 40, [3], 8),          # first-layer VALU kernel, rgb, ragged patches
    (1, 16, 16, [4], 64),
    (5, 256, 256, [1], 32),       # first layer, more patches than persistent workgroups: prefetch + both LDS buffers
    (5, 250, 250, [3], 16),       # the same with ragged border patches, three input channels
    (5, 256, 256, [4], 64),       # four input channels (three workgroups per CU), sixteen channel quads
])
def test_conv3x3_fwd_multiview(dev, shape, conv_algo):
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, cins, co = shape
    g = torch.Generator().manual_seed(1)
    srcs = [torch.randn(b, c, h, w, generator=g) for c in cins]
    wt = torch.randn(co, sum(cins), 3, 3, generator=g) * 0.2
    bias = torch.randn(co, generator=g)
    ref = F.conv2d(torch.cat(srcs, 1).double(), wt.double(), bias.double(), padding=1).float()
    out = torch.full((b, h, w, co), float("nan"), device=dev)
    ops.gemm_fwd(b, h, w, 9, [V(nhwc(s)) for s in srcs], [V(out)], engine.pack_conv_fwd(wt.cuda()), bias.cuda())
    assert rel_err(nchw(out), ref) < TOL
    # ReLU epilogue + BN partial sums
    out2 = torch.empty_like(out)
    blocks = ops.gemm_pixel_blocks(b, h, w)
    part = torch.empty(blocks * co * 2, device=dev)
    ops.gemm_fwd(b, h, w, 9, [V(nhwc(s)) for s in srcs], [V(out2)], engine.pack_conv_fwd(wt.cuda()), bias.cuda(), part)
    sums = part.view(blocks, co, 2).double().sum(0).cpu()
    assert rel_err(sums[:, 0], ref.double().sum((0, 2, 3))) < TOL * 10  # plain sums cancel; looser
    assert rel_err(sums[:, 1], (ref.double() ** 2).sum((0, 2, 3))) < TOL


@pytest.mark.parametrize("shape", [
    # (B, H, W, [cin per source], cout): which kernel takes the statistics, and who finalizes
    (32, 64, 64, [32], 32),           # Winograd lean kernel, 512 persistent workgroups with 1 unit each
    (8, 128, 128, [32, 32], 64),      # Winograd, two column groups per patch, several units per workgroup
    (3, 40, 24, [16], 24),            # ragged patches, partial column tile
    (16, 256, 256, [1], 32),          # first-layer VALU kernel, 1024 persistent workgroups, 8 patches each
    (2, 24, 40, [3], 8),              # first layer, rgb, fewer patches than workgroups
    (1, 16, 16, [64], 288),           # more columns than a workgroup row takes: per-block rows
    (2, 8, 8, [5, 3], 7),             # generic kernel: per-block rows
])
@pytest.mark.parametrize("fold", [False, True])
def test_fused_batchnorm_finalize_matches_separate_launch(dev, shape, fold, conv_algo):
    """unetpp_bn_fused: the convolution call that takes the statistics also leaves mean / invstd / scale / shift and
    the running statistics behind (per-workgroup rows in the persistent kernels, per-block rows in the others; the
    finalize is enqueued by the library).  Against the float64 statistics of the stored tensor, against the two-call
    path, three times in a row into a NaN-filled workspace, with other work in flight before it."""
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, cins, co = shape
    g = torch.Generator().manual_seed(3)
    srcs = [nhwc(torch.randn(b, c, h, w, generator=g)) for c in cins]
    wt = (torch.randn(co, sum(cins), 3, 3, generator=g) * 0.2).cuda()
    bias = torch.randn(co, generator=g).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(co, generator=g)).cuda(), (0.1 * torch.randn(co, generator=g)).cuda()
    kw = {}
    if fold and cins[0] >= 8 and len(cins) == 1:   # a folded BatchNorm + ReLU on the input view (encoder conv2)
        kw = dict(scale=(1 + 0.1 * torch.randn(cins[0], generator=g)).cuda(), shift=(0.1 * torch.randn(cins[0], generator=g)).cuda(), relu=True)
    ins = [V(s, **kw) for s in srcs]
    wp = engine.pack_conv_fwd(wt)
    count = b * h * w
    # two-launch reference
    y_ref = torch.empty(b, h, w, co, device=dev)
    blocks = ops.gemm_pixel_blocks(b, h, w)
    part = torch.empty(blocks * co * 2, device=dev)
    ops.gemm_fwd(b, h, w, 9, ins, [V(y_ref)], wp, bias, part)
    rm_ref, rv_ref = torch.zeros(co, device=dev), torch.ones(co, device=dev)
    ref = ops.bn_finalize(part, blocks, co, count, gamma, beta, 1e-5, 0.1, rm_ref, rv_ref)
    # fused
    rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
    rows = ops.gemm_stats_rows(b, h, w)
    busy = torch.randn(1 << 22, device=dev)
    for rep in range(3):
        y = torch.full((b, h, w, co), float("nan"), device=dev)
        workspace = torch.full((rows * co * 2,), float("nan"), device=dev)   # stale rows must not matter
        fin = ops.BatchNormFinish(gamma, beta, rm, rv, 1e-5, 0.1, count)
        for _ in range(4):
            busy = busy * 1.0001 + 0.5      # other kernels in front of the launch: its workgroups start unevenly
        ops.gemm_fwd(b, h, w, 9, ins, [V(y)], wp, bias, workspace, bn=fin)
        assert torch.equal(y, y_ref)
        yd = y.double().view(-1, co)
        mean, var = yd.mean(0), yd.var(0, unbiased=False)
        assert float((fin.mean.cpu().double() - mean.cpu()).abs().max()) < 1e-5 * max(1.0, float(mean.abs().max()))
        assert rel_err(fin.invstd.cpu(), (1 / (var + 1e-5).sqrt()).cpu()) < 1e-5
        for got, want in zip((fin.mean, fin.invstd, fin.scale, fin.shift), ref):
            assert rel_err(got.cpu(), want.cpu()) < 1e-5
    # three updates of the running statistics with momentum 0.1 == three two-launch updates
    for _ in range(2):
        ops.bn_finalize(part, blocks, co, count, gamma, beta, 1e-5, 0.1, rm_ref, rv_ref)
    assert rel_err(rm.cpu(), rm_ref.cpu()) < 1e-5 and rel_err(rv.cpu(), rv_ref.cpu()) < 1e-5


@pytest.mark.parametrize("ck", [12, 16])  # 12: partial 8-channel chunk (general kernel); 16: the lean fold kernel
def test_conv3x3_load_transform_and_slices(dev, conv_algo, ck):
    """affine + ReLU applied on load (zero padding AFTER the transform), channel-sliced views, store gate/accumulate."""
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w = 2, 16, 16
    g = torch.Generator().manual_seed(2)
    big = torch.randn(b, 28, h, w, generator=g)          # use channels 8..8+ck
    scale, shift = torch.randn(ck, generator=g), torch.randn(ck, generator=g)
    wt = torch.randn(16, ck, 3, 3, generator=g) * 0.2
    xin = F.relu(big[:, 8:8 + ck] * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    ref = F.conv2d(xin.double(), wt.double(), None, padding=1).float()
    gate = torch.randn(b, 16, h, w, generator=g)
    prev = torch.randn(b, 16, h, w, generator=g)
    expect = prev + ref * (gate > 0)
    out = nhwc(prev)
    ops.gemm_fwd(b, h, w, 9, [V(nhwc(big), c_off=8, c_len=ck, scale=scale.cuda(), shift=shift.cuda(), relu=True)],
                 [V(out, gate=nhwc(gate), accumulate=True)], engine.pack_conv_fwd(wt.cuda()))
    assert rel_err(nchw(out), expect) < TOL
    # gate_sum: the ReLU mask is applied to the accumulated sum (the "last contributor" form); plain input
    # views so the fast kernel runs, then the same through the generic kernel
    plain = nhwc(xin)
    for fast in (True, False):
        ops.USE_FAST_GEMM = fast
        try:
            out = nhwc(prev)
            ops.gemm_fwd(b, h, w, 9, [V(plain)], [V(out, gate=nhwc(gate), accumulate=True, gate_sum=True)],
                         engine.pack_conv_fwd(wt.cuda()))
        finally:
            ops.USE_FAST_GEMM = True
        assert rel_err(nchw(out), (prev + ref) * (gate > 0)) < TOL


@pytest.mark.parametrize("hw", [(24, 40), (64, 64), (5, 7)])
def test_conv3x3_mixed_folded_views(dev, hw, conv_algo):
    """Virtual concat of a plain view, a view with affine + ReLU on load and a view with the affine alone (whole
    8-channel chunks: the lean kernel with the fold), ragged patches: the padding must be zero AFTER the transform."""
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, (h, w) = 2, hw
    g = torch.Generator().manual_seed(12)
    xs = [torch.randn(b, c, h, w, generator=g) for c in (8, 16, 8)]
    sc1, sh1 = torch.randn(16, generator=g), torch.randn(16, generator=g) + 0.5
    sc2, sh2 = torch.randn(8, generator=g), torch.randn(8, generator=g) - 0.5
    wt = torch.randn(24, 32, 3, 3, generator=g) * 0.2
    bias = torch.randn(24, generator=g)
    x1 = F.relu(xs[1] * sc1.view(1, -1, 1, 1) + sh1.view(1, -1, 1, 1))
    x2 = xs[2] * sc2.view(1, -1, 1, 1) + sh2.view(1, -1, 1, 1)
    ref = F.conv2d(torch.cat([xs[0], x1, x2], 1).double(), wt.double(), bias.double(), padding=1).float()
    out = torch.full((b, h, w, 24), float("nan"), device=dev)
    ops.gemm_fwd(b, h, w, 9, [V(nhwc(xs[0])), V(nhwc(xs[1]), scale=sc1.cuda(), shift=sh1.cuda(), relu=True),
                              V(nhwc(xs[2]), scale=sc2.cuda(), shift=sh2.cuda())],
                 [V(out)], engine.pack_conv_fwd(wt.cuda()), bias.cuda())
    assert rel_err(nchw(out), ref) < TOL


@pytest.mark.parametrize("shape", [(2, 32, 32, [32], 32), (1, 64, 64, [32, 32, 32, 32], 32), (2, 8, 8, [16, 8, 8], 40),
                                   (1, 24, 40, [8], 4), (3, 16, 16, [64, 32], 96)])
def test_fast_and_generic_gemm_agree(dev, shape):
    """The register-prefetched kernel and the generic kernel are the same function of their inputs (bitwise:
    same MFMA order), including the BatchNorm partial sums."""
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, cins, co = shape
    g = torch.Generator().manual_seed(9)
    srcs = [nhwc(torch.randn(b, c, h, w, generator=g)) for c in cins]
    wp = engine.pack_conv_fwd((torch.randn(co, sum(cins), 3, 3, generator=g) * 0.2).cuda())
    bias = torch.randn(co, generator=g).cuda()
    res = []
    for fast in (True, False):
        ops.USE_FAST_GEMM = fast
        try:
            out = torch.empty(b, h, w, co, device=dev)
            part = torch.empty(ops.gemm_pixel_blocks(b, h, w) * co * 2, device=dev)
            ops.gemm_fwd(b, h, w, 9, [V(s) for s in srcs], [V(out)], wp, bias, part, direct=True)
        finally:
            ops.USE_FAST_GEMM = True
        res.append((out, part))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("shape", [(2, 32, 32, [32], 32), (1, 64, 64, [32, 32, 32, 32], 32), (2, 8, 8, [16, 8, 8], 40),
                                   (1, 24, 40, [8], 4), (3, 16, 16, [64, 32], 96), (2, 3, 5, [12], 16),
                                   (1, 37, 21, [20, 4], 36), (1, 128, 128, [64], 64),
                                   (2, 32, 32, [16], 16), (1, 64, 32, [32, 8], 16)])  # <= 16 columns: half-tile kernel
def test_winograd_matches_direct(dev, shape):
    """Winograd F(2x2,3x3) and direct summation are the same convolution up to fp32 rounding: outputs within 2e-6 of
    the largest magnitude (odd image sizes, ragged patches, partial column tiles and K chunks included), BatchNorm
    partial sums within 1e-5, and the library reports which kernel ran."""
    from unet_nested4tiny_objects_keypoints_amd import _lib, engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, cins, co = shape
    g = torch.Generator().manual_seed(11)
    srcs = [nhwc(torch.randn(b, c, h, w, generator=g)) for c in cins]
    wp = engine.pack_conv_fwd((torch.randn(co, sum(cins), 3, 3, generator=g) * 0.2).cuda())
    bias = torch.randn(co, generator=g).cuda()
    res, names = [], []
    for direct in (False, True):
        out = torch.full((b, h, w, co), float("nan"), device=dev)
        part = torch.empty(ops.gemm_pixel_blocks(b, h, w) * co * 2, device=dev)
        ops.gemm_fwd(b, h, w, 9, [V(s) for s in srcs], [V(out, relu=True)], wp, bias, part, direct=direct)
        names.append(_lib.lib().unetpp_last_kernel_name().decode())
        res.append((out, part))
    assert names == ["gemm_wino_kernel", "gemm_fast_kernel<9>"]
    scale = res[1][0].abs().max().item()
    assert (res[0][0] - res[1][0]).abs().max().item() <= 2e-6 * scale
    assert rel_err(res[0][1].cpu(), res[1][1].cpu()) < 1e-5


def test_conv3x3_dgrad_matches_autograd(dev, conv_algo):
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, ci, co = 2, 16, 24, 20, 12
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, ci, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64)
    dy = torch.randn(b, co, h, w, generator=g, dtype=torch.float64)
    F.conv2d(x, wt, None, padding=1).backward(dy)
    # split the input gradient over two destination tensors (virtual concat backward)
    d0 = torch.empty(b, h, w, 8, device=dev)
    d1 = torch.empty(b, h, w, 12, device=dev)
    ops.gemm_fwd(b, h, w, 9, [V(nhwc(dy.float()))], [V(d0), V(d1)], engine.pack_conv_dgrad(wt.float().cuda()))
    got = torch.cat([nchw(d0), nchw(d1)], 1)
    assert rel_err(got, x.grad.float()) < TOL


@pytest.mark.parametrize("shape", [(2, 8, 8, 16, 8), (1, 16, 32, 64, 32), (1, 4, 4, 6, 5)])
def test_deconv2x2_fwd_dgrad_wgrad(dev, shape):
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, ci, co = shape
    g = torch.Generator().manual_seed(4)
    x = torch.randn(b, ci, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(ci, co, 2, 2, generator=g, dtype=torch.float64, requires_grad=True)
    bias = torch.randn(co, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(x, wt, bias, stride=2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xg, wg, bg = nhwc(x.detach().float()), wt.detach().float().cuda(), bias.detach().float().cuda()
    up = torch.full((b, 2 * h, 2 * w, co), float("nan"), device=dev)
    ops.gemm_fwd(b, h, w, 1, [V(xg)], engine._phase_views(up), engine.pack_deconv_fwd(wg), engine.tile_bias4(bg))
    assert rel_err(nchw(up), y.detach().float()) < TOL
    d_up = nhwc(dy.float())
    dx = torch.empty_like(xg)
    ops.gemm_fwd(b, h, w, 1, engine._phase_views(d_up), [V(dx)], engine.pack_deconv_dgrad(wg))
    assert rel_err(nchw(dx), x.grad.float()) < TOL
    dw, db = torch.empty_like(wg), torch.empty_like(bg)
    ops.wgrad(b, h, w, 1, [V(xg)], engine._phase_views(d_up), dw, (0, 4 * co, 4, 1), db, n_inner=co)
    assert rel_err(dw.cpu(), wt.grad.float()) < TOL
    assert rel_err(db.cpu(), bias.grad.float()) < TOL


@pytest.mark.parametrize("shape", [(2, 16, 16, [8], 16), (1, 32, 64, [32, 16], 64), (2, 24, 40, [8], 12),
                                   (1, 37, 21, [20, 4], 36), (1, 3, 5, [4], 6), (2, 64, 64, [64], 32)])
@pytest.mark.parametrize("mode", ["plain", "relu", "gate", "gate_sum", "accumulate", "sliced", "stats"])
def test_pointwise_fast_and_generic_agree(dev, shape, mode):
    """Pointwise (1x1) GEMM: the swapped-operand register-direct epilogue of the fast kernel against the generic
    kernel, bitwise, for every store mode (ReLU, gate before/after the sum, accumulate, channel-sliced output in a
    wider tensor, unaligned column counts -> scalar stores, statistics launches -> pixel-major epilogue) and against
    the fp64 statement."""
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, cins, co = shape
    g = torch.Generator().manual_seed(21)
    srcs = [torch.randn(b, c, h, w, generator=g) for c in cins]
    wt = torch.randn(co, sum(cins), 1, 1, generator=g) * 0.3
    bias = torch.randn(co, generator=g)
    gate = torch.randn(b, co, h, w, generator=g)
    prev = torch.randn(b, co, h, w, generator=g)
    ref = F.conv2d(torch.cat(srcs, 1).double(), wt.double(), bias.double())
    if mode == "relu":
        ref = F.relu(ref)
    elif mode == "gate":
        ref = prev.double() + ref * (gate > 0)
    elif mode == "gate_sum":
        ref = (prev.double() + ref) * (gate > 0)
    elif mode == "accumulate":
        ref = prev.double() + ref
    wp = engine.pack_conv_fwd(wt.cuda())
    res = []
    for fast in (True, False):
        ops.USE_FAST_GEMM = fast
        try:
            with ops._lib.debug_switch("PW_DIRECT", 0):   # (gemm_pw.hip sums K in another order: test_pointwise_direct_kernel)
                part = None
                if mode == "sliced":
                    wide = torch.full((b, h, w, co + 12), 7.0, device=dev)
                    outs = [V(wide, c_off=8, c_len=co)]
                else:
                    wide = nhwc(prev)
                    kw = {"relu": {"relu": True}, "gate": {"gate": nhwc(gate), "accumulate": True},
                          "gate_sum": {"gate": nhwc(gate), "accumulate": True, "gate_sum": True},
                          "accumulate": {"accumulate": True}}.get(mode, {})
                    outs = [V(wide, **kw)]
                    if mode == "stats":
                        part = torch.empty(ops.gemm_pixel_blocks(b, h, w) * co * 2, device=dev)
                ops.gemm_fwd(b, h, w, 1, [V(nhwc(s)) for s in srcs], outs, wp, bias.cuda(), part)
        finally:
            ops.USE_FAST_GEMM = True
        res.append((wide, part))
    assert torch.equal(res[0][0], res[1][0])
    if mode == "stats":
        assert torch.equal(res[0][1], res[1][1])
    got = res[0][0]
    if mode == "sliced":
        assert bool((got[..., :8] == 7.0).all()) and bool((got[..., 8 + co:] == 7.0).all())
        got = got[..., 8:8 + co]
    assert rel_err(nchw(got), ref.float()) < TOL


@pytest.mark.parametrize("case", [
    # (B, H, W, cin per input view, cout per output view, form)
    (2, 16, 32, [64], [32] * 4, "deconv_fwd"),        # level 0 of the depth-4 network: K 64, N 128
    (1, 16, 16, [128], [64] * 4, "deconv_fwd"),       # level 1: K 128, N 256 = two column passes, one 128 KB image
    (2, 8, 16, [32], [32] * 4, "deconv_fwd"),         # K 32
    (1, 32, 48, [16], [32] * 4, "deconv_fwd"),        # K 16, three tiles per row (division instead of a shift)
    (2, 16, 32, [32] * 4, [64], "deconv_dgrad"),      # K 128 (four phase views), N 64, gate on the sum + accumulate
    (1, 16, 16, [64] * 4, [128], "deconv_dgrad"),     # K 256 = two K chunks
    (1, 8, 16, [128] * 4, [64], "deconv_dgrad"),      # K 512 = four K chunks
    (2, 16, 16, [32, 32], [64], "plain"),             # 1x1 convolution over a virtual concatenation
    (2, 16, 16, [32, 16], [64], "plain"),             # K = 48: three 16-channel groups -> gemm_fast
    (1, 32, 64, [64], [16], "relu"),
    (1, 16, 16, [32], [48], "plain"),                 # three 16-column blocks: not a block count of the kernel -> gemm_fast
    (2, 16, 32, [16], [32], "gate"),
    (2, 16, 32, [64], [64], "accumulate"),
    (1, 16, 16, [32], [32], "sliced"),
    (1, 64, 64, [256], [128], "plain"),               # 128 KB image, one workgroup of sixteen waves per CU
    (1, 8, 16, [256], [128] * 4, "deconv_fwd"),       # level 2: K 256, N 512 = 512 KB of weights: beyond the LDS -> gemm_fast
    (1, 8, 16, [128] * 4, [256], "deconv_dgrad"),     # K 512, N 256: the same
    (2, 16, 16, [64], [192], "relu"),                 # 12 column blocks: no block count of the kernel -> gemm_fast
])
def test_pointwise_direct_kernel(dev, case):
    """gemm_pw.hip (weights resident in LDS, activations loaded straight into the MFMA operand registers, register-direct
    16-byte stores) in every form the network launches -- transposed convolution forward into four phase views and its
    input gradient out of them, 1x1 convolutions with ReLU / gate / accumulate / sliced outputs -- against the float64
    statement and against gemm_fast_kernel<1> (same arithmetic in another summation order: 2e-6 of the largest value)."""
    from unet_nested4tiny_objects_keypoints_amd import _lib, engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, cins, couts, form = case
    g = torch.Generator().manual_seed(33)
    ci, co = sum(cins), sum(couts)

    def run(direct):
        with _lib.debug_switch("PW_DIRECT", 1 if direct else 0):
            if form == "deconv_fwd":
                x = torch.randn(b, ci, h, w, generator=torch.Generator().manual_seed(1))
                wt = torch.randn(ci, couts[0], 2, 2, generator=torch.Generator().manual_seed(2)) * 0.2
                bias = torch.randn(couts[0], generator=torch.Generator().manual_seed(3))
                ref = F.conv_transpose2d(x.double(), wt.double(), bias.double(), stride=2)
                up = torch.full((b, 2 * h, 2 * w, couts[0]), float("nan"), device=dev)
                ops.gemm_fwd(b, h, w, 1, [V(nhwc(x))], engine._phase_views(up), engine.pack_deconv_fwd(wt.cuda()),
                             engine.tile_bias4(bias.cuda()))
                return nchw(up), ref
            if form == "deconv_dgrad":
                cu = cins[0]
                x = torch.randn(b, co, h, w, generator=torch.Generator().manual_seed(1), dtype=torch.float64, requires_grad=True)
                wt = torch.randn(co, cu, 2, 2, generator=torch.Generator().manual_seed(2), dtype=torch.float64) * 0.2
                dy = torch.randn(b, cu, 2 * h, 2 * w, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
                F.conv_transpose2d(x, wt, None, stride=2).backward(dy)
                gate = torch.randn(b, co, h, w, generator=torch.Generator().manual_seed(4))
                prev = torch.randn(b, co, h, w, generator=torch.Generator().manual_seed(5))
                ref = (prev.double() + x.grad) * (gate > 0)
                dx = nhwc(prev)
                ops.gemm_fwd(b, h, w, 1, engine._phase_views(nhwc(dy.float())),
                             [V(dx, accumulate=True, gate=nhwc(gate), gate_sum=True)], engine.pack_deconv_dgrad(wt.float().cuda()))
                return nchw(dx), ref
            srcs = [torch.randn(b, c, h, w, generator=torch.Generator().manual_seed(10 + i)) for i, c in enumerate(cins)]
            wt = torch.randn(co, ci, 1, 1, generator=torch.Generator().manual_seed(2)) * 0.3
            bias = torch.randn(co, generator=torch.Generator().manual_seed(3))
            gate = torch.randn(b, co, h, w, generator=torch.Generator().manual_seed(4))
            prev = torch.randn(b, co, h, w, generator=torch.Generator().manual_seed(5))
            ref = F.conv2d(torch.cat(srcs, 1).double(), wt.double(), bias.double())
            if form == "relu":
                ref = F.relu(ref)
            elif form == "gate":
                ref = prev.double() + ref * (gate > 0)
            elif form == "accumulate":
                ref = prev.double() + ref
            if form == "sliced":
                wide = torch.full((b, h, w, co + 12), 7.0, device=dev)
                outs = [V(wide, c_off=8, c_len=co)]
            else:
                wide = nhwc(prev)
                outs = [V(wide, **{"relu": {"relu": True}, "gate": {"gate": nhwc(gate), "accumulate": True},
                                   "accumulate": {"accumulate": True}}.get(form, {}))]
            ops.gemm_fwd(b, h, w, 1, [V(nhwc(s)) for s in srcs], outs, engine.pack_conv_fwd(wt.cuda()), bias.cuda())
            if form == "sliced":
                assert bool((wide[..., :8] == 7.0).all()) and bool((wide[..., 8 + co:] == 7.0).all())
                wide = wide[..., 8:8 + co]
            return nchw(wide), ref

    got, ref = run(True)
    name = _lib.lib().unetpp_last_kernel_name().decode()
    def blocks_ok(c):
        return (c // 16) in (1, 2, 4, 8) or (c // 16) % 8 == 0
    takes = blocks_ok(ci) and blocks_ok(co) and (ci * co + co) * 4 <= 148 * 1024
    assert name == ("gemm_pw_kernel" if takes else "gemm_fast_kernel<1>"), name
    assert rel_err(got, ref.float()) < TOL
    other, _ = run(False)
    assert _lib.lib().unetpp_last_kernel_name().decode() == "gemm_fast_kernel<1>"
    assert (got - other).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("case", [
    # (B, H, W, cin, cout of the transposed convolution | [cin per x view], cout of a 1x1 convolution), target_blocks
    (2, 16, 32, 64, 32, 256),        # level 0: one 64 x 128 block, eight pixel ranges per workgroup
    (2, 16, 32, 64, 32, 3),          # three workgroups: uneven row shares
    (1, 16, 16, 128, 64, 256),       # level 1: 2 x 2 blocks, two pixel ranges each
    (1, 8, 8, 256, 128, 256),        # level 2: 16 blocks = two workgroup rows, one pixel range each
    (1, 4, 12, 64, 32, 1024),        # more workgroups than image rows: empty pixel ranges contribute zeros
    (2, 8, 16, [32, 32], 128, 256),  # 1x1 convolution over a virtual concatenation, plain dy view
    (1, 8, 8, 64, 16, 256),          # N = 64: not a block shape of the kernel -> wgrad_dma_kernel<1>
])
def test_pointwise_wgrad_direct_kernel(dev, case):
    """wgrad_pw.hip (both operands loaded straight into the MFMA operand registers, pixels as the contraction index)
    against float64 autograd: transposed-convolution weight / bias gradients at the three block structures of the
    network's levels, uneven and empty pixel ranges, and a 1x1 convolution over two x views."""
    from unet_nested4tiny_objects_keypoints_amd import _lib, engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, ci, co, target_blocks = case
    g = torch.Generator().manual_seed(17)
    if isinstance(ci, list):
        xs = [torch.randn(b, c, h, w, generator=g, dtype=torch.float64) for c in ci]
        wt = torch.randn(co, sum(ci), 1, 1, generator=g, dtype=torch.float64, requires_grad=True)
        bias = torch.zeros(co, dtype=torch.float64, requires_grad=True)
        dy = torch.randn(b, co, h, w, generator=g, dtype=torch.float64)
        F.conv2d(torch.cat(xs, 1), wt, bias).backward(dy)
        dw, db = torch.empty(co, sum(ci), 1, 1, device=dev), torch.empty(co, device=dev)
        ops.wgrad(b, h, w, 1, [V(nhwc(x.float())) for x in xs], [V(nhwc(dy.float()))], dw, (0, 1, sum(ci), 0), db,
                  target_blocks=target_blocks)
        k, n = sum(ci), co
    else:
        x = torch.randn(b, ci, h, w, generator=g, dtype=torch.float64)
        wt = torch.randn(ci, co, 2, 2, generator=g, dtype=torch.float64, requires_grad=True)
        bias = torch.zeros(co, dtype=torch.float64, requires_grad=True)
        dy = torch.randn(b, co, 2 * h, 2 * w, generator=g, dtype=torch.float64)
        F.conv_transpose2d(x, wt, bias, stride=2).backward(dy)
        dw, db = torch.empty(ci, co, 2, 2, device=dev), torch.empty(co, device=dev)
        ops.wgrad(b, h, w, 1, [V(nhwc(x.float()))], engine._phase_views(nhwc(dy.float())), dw, (0, 4 * co, 4, 1), db,
                  n_inner=co, target_blocks=target_blocks)
        k, n = ci, 4 * co
    name = _lib.lib().unetpp_last_kernel_name().decode()
    assert name == ("wgrad_pw_kernel" if (k % 64 == 0 and n % 128 == 0) else "wgrad_dma_kernel<1>"), name
    assert rel_err(dw.cpu(), wt.grad.float()) < TOL
    assert rel_err(db.cpu(), bias.grad.float()) < TOL
    with _lib.debug_switch("PW_DIRECT", 0):   # the LDS-staged kernel on the same operands: same sums up to their order
        dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
        if isinstance(ci, list):
            ops.wgrad(b, h, w, 1, [V(nhwc(x.float())) for x in xs], [V(nhwc(dy.float()))], dw2, (0, 1, sum(ci), 0), db2)
        else:
            ops.wgrad(b, h, w, 1, [V(nhwc(x.float()))], engine._phase_views(nhwc(dy.float())), dw2, (0, 4 * co, 4, 1), db2, n_inner=co)
        assert _lib.lib().unetpp_last_kernel_name().decode() == "wgrad_dma_kernel<1>"
    assert (dw - dw2).abs().max().item() <= 2e-5 * wt.grad.abs().max().item()
    assert (db - db2).abs().max().item() <= 2e-5 * bias.grad.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 16, 16, [8], 8), (1, 32, 32, [32, 32, 32], 32), (2, 24, 40, [3], 4),
                                   (1, 8, 8, [40, 5], 33), (3, 64, 64, [16], 16), (2, 32, 64, [1], 32),
                                   (1, 24, 40, [3], 8)])
def test_conv3x3_wgrad(dev, shape):
    from unet_nested4tiny_objects_keypoints_amd import ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, cins, co = shape
    g = torch.Generator().manual_seed(5)
    srcs = [torch.randn(b, c, h, w, generator=g, dtype=torch.float64) for c in cins]
    wt = torch.randn(co, sum(cins), 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    act = torch.randn(b, co, h, w, generator=g, dtype=torch.float64)  # ReLU gate
    dy = torch.randn(b, co, h, w, generator=g, dtype=torch.float64)
    F.conv2d(torch.cat(srcs, 1), wt, bias, padding=1).backward(dy * (act > 0))
    dw = torch.empty(co, sum(cins), 3, 3, device=dev)
    db = torch.empty(co, device=dev)
    ci = sum(cins)
    for target_blocks in (1024, 3):  # many slabs / few slabs with a long per-block pixel loop
        ops.wgrad(b, h, w, 9, [V(nhwc(s.float())) for s in srcs], [V(nhwc(dy.float()), gate=nhwc(act.float()))],
                  dw, (1, 9, ci * 9, 0), db, target_blocks=target_blocks)
        assert rel_err(dw.cpu(), wt.grad.float()) < TOL
        assert rel_err(db.cpu(), bias.grad.float()) < TOL


@pytest.mark.parametrize("shape", [(1, 32, 32, [32, 32, 32], 32), (3, 64, 64, [16], 16), (2, 24, 40, [8], 12),
                                   (1, 37, 21, [20, 12], 36), (2, 16, 16, [8], 8), (1, 128, 128, [64], 64),
                                   (2, 6, 18, [12], 40),
                                   (1, 16, 24, [256], 256), (1, 8, 8, [128, 192], 224)])   # wide: the transposing finish
@pytest.mark.parametrize("direct", [False, True])
def test_conv3x3_wgrad_plain_views(dev, shape, direct):
    """Plain views: the LDS-DMA kernels.  Winograd F(2x2,3x3) weight gradient (images at least 17 wide) and the direct
    sum both match autograd in fp64 within 1e-4 (observed ~1e-6), for many and for few slabs."""
    from unet_nested4tiny_objects_keypoints_amd import _lib, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, cins, co = shape
    g = torch.Generator().manual_seed(6)
    srcs = [torch.randn(b, c, h, w, generator=g, dtype=torch.float64) for c in cins]
    wt = torch.randn(co, sum(cins), 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(b, co, h, w, generator=g, dtype=torch.float64)
    F.conv2d(torch.cat(srcs, 1), wt, bias, padding=1).backward(dy)
    ci = sum(cins)
    for target_blocks in (1024, 3):
        dw = torch.full((co, ci, 3, 3), float("nan"), device=dev)
        db = torch.full((co,), float("nan"), device=dev)
        ops.wgrad(b, h, w, 9, [V(nhwc(s.float())) for s in srcs], [V(nhwc(dy.float()))], dw, (1, 9, ci * 9, 0), db,
                  target_blocks=target_blocks, direct=direct)
        name = _lib.lib().unetpp_last_kernel_name().decode()
        assert name == ("wgrad_wino_kernel" if (not direct and w > 16) else "wgrad_dma_kernel<9>")
        assert rel_err(dw.cpu(), wt.grad.float()) < TOL
        assert rel_err(db.cpu(), bias.grad.float()) < TOL


@pytest.mark.parametrize("shape", [(2, 32, 32, 32, 32), (1, 24, 40, 12, 20), (2, 37, 21, 24, 8), (1, 64, 64, 64, 64)])
@pytest.mark.parametrize("direct", [False, True])
def test_conv3x3_wgrad_folded_batchnorm_view(dev, shape, direct):
    """x view with the producer's BatchNorm apply + ReLU folded into the operand read (scale/shift + relu, zero padding
    AFTER the transform; negative, zero and positive scales): Winograd kernel (NaN padding trick) and direct kernel."""
    from unet_nested4tiny_objects_keypoints_amd import _lib, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, ci, co = shape
    g = torch.Generator().manual_seed(8)
    y1 = torch.randn(b, ci, h, w, generator=g, dtype=torch.float64)
    scale = torch.randn(ci, generator=g, dtype=torch.float64)
    scale[0] = 0.0
    shift = torch.randn(ci, generator=g, dtype=torch.float64)
    xin = F.relu(y1 * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    wt = torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(b, co, h, w, generator=g, dtype=torch.float64)
    F.conv2d(xin, wt, bias, padding=1).backward(dy)
    dw = torch.full((co, ci, 3, 3), float("nan"), device=dev)
    db = torch.full((co,), float("nan"), device=dev)
    ops.wgrad(b, h, w, 9, [V(nhwc(y1.float()), scale=scale.float().cuda(), shift=shift.float().cuda(), relu=True)],
              [V(nhwc(dy.float()))], dw, (1, 9, ci * 9, 0), db, direct=direct)
    name = _lib.lib().unetpp_last_kernel_name().decode()
    assert name == ("wgrad_fast_kernel<9>" if direct else "wgrad_wino_kernel")
    assert rel_err(dw.cpu(), wt.grad.float()) < TOL
    assert rel_err(db.cpu(), bias.grad.float()) < TOL


@pytest.mark.parametrize("shape", [(2, 32, 64, 1, 32), (1, 24, 40, 3, 8), (3, 16, 16, 4, 128), (2, 8, 8, 2, 4)])
def test_first_layer_wgrad(dev, shape):
    """The 1..4-channel first convolution's dedicated weight-gradient kernel (plain dy, no gate)."""
    from unet_nested4tiny_objects_keypoints_amd import ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, ci, co = shape
    g = torch.Generator().manual_seed(15)
    x = torch.randn(b, ci, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(b, co, h, w, generator=g, dtype=torch.float64)
    F.conv2d(x, wt, bias, padding=1).backward(dy)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    db = torch.empty(co, device=dev)
    ops.wgrad(b, h, w, 9, [V(nhwc(x.float()))], [V(nhwc(dy.float()))], dw, (1, 9, ci * 9, 0), db)
    assert rel_err(dw.cpu(), wt.grad.float()) < TOL
    assert rel_err(db.cpu(), bias.grad.float()) < TOL


@pytest.mark.parametrize("c", [32, 8, 3])
def test_batchnorm_relu_pool_fwd_bwd(dev, c):
    from unet_nested4tiny_objects_keypoints_amd import ops
    b, h, w = 3, 16, 24
    g = torch.Generator().manual_seed(6)
    y = (torch.randn(b, c, h, w, generator=g, dtype=torch.float64) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(c, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(c, generator=g, dtype=torch.float64)).requires_grad_(True)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    act = F.relu(F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5))
    pooled, idx = F.max_pool2d(act, 2, return_indices=True)
    d_act = torch.randn(act.shape, generator=g, dtype=torch.float64)
    d_pool = torch.randn(pooled.shape, generator=g, dtype=torch.float64)
    (act * d_act).sum().backward(retain_graph=True)
    gy_act, gg, gb = y.grad.clone(), gamma.grad.clone(), beta.grad.clone()
    y.grad = None
    (pooled * d_pool).sum().backward()
    gy_pool = y.grad.clone()  # includes BN backward of the pooled path; used via d_act accumulation below

    yg = nhwc(y.detach().float())
    # statistics come from the conv epilogue in the product; here: per-"block" sums computed on the host
    part = torch.stack([yg.view(-1, c).sum(0), (yg.view(-1, c) ** 2).sum(0)], 1).reshape(1, c, 2).contiguous()
    rmg, rvg = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    mean, invstd, scale, shift = ops.bn_finalize(part.view(-1), 1, c, b * h * w, gamma.detach().float().cuda(),
                                                 beta.detach().float().cuda(), 1e-5, 0.1, rmg, rvg)
    assert rel_err(rmg.cpu(), rm.float()) < TOL and rel_err(rvg.cpu(), rv.float()) < TOL
    act_g = torch.empty_like(yg)
    pooled_g = torch.empty(b, h // 2, w // 2, c, device=dev)
    idx_g = torch.empty(b, h // 2, w // 2, c, dtype=torch.uint8, device=dev)
    ops.affine_relu_pool(yg, scale, shift, True, act_g, pooled_g, idx_g)
    assert rel_err(nchw(act_g), act.detach().float()) < TOL
    assert rel_err(nchw(pooled_g), pooled.detach().float()) < TOL
    # backward: d_act (+ pool gradient routed to the argmax) -> BN backward
    dact_g = nhwc(d_act.float())
    dy_g = torch.empty_like(dact_g)
    dgam, dbet = ops.bn_backward(dact_g, yg, scale, shift, mean, invstd, gamma.detach().float().cuda(), dy_g)
    assert rel_err(nchw(dy_g), gy_act.float()) < TOL
    assert rel_err(dgam.cpu(), gg.float()) < TOL and rel_err(dbet.cpu(), gb.float()) < TOL
    acc = torch.zeros_like(dact_g)
    ops.maxpool_bwd(nhwc(d_pool.float()), idx_g, acc)
    ops.bn_backward(acc, yg, scale, shift, mean, invstd, gamma.detach().float().cuda(), acc)
    assert rel_err(nchw(acc), gy_pool.float()) < TOL


@pytest.mark.parametrize("b,h,w,c,affine,relu", [(2, 8, 64, 32, True, True), (3, 6, 40, 20, True, True),
                                                 (1, 4, 128, 64, False, False), (2, 4, 8, 8, True, True),
                                                 (2, 5, 7, 8, True, True), (1, 9, 6, 3, True, True),    # odd sizes: floor
                                                 (1, 6, 5, 16, False, False)])
def test_pool_argmax_and_routing_exact(dev, b, h, w, c, affine, relu):
    """Max-pool value, argmax (first maximum in scan order, ties from the ReLU zeros included) and the gradient
    routing, bit-exact against torch on the same activation; covers the row-structured and the generic kernels."""
    from unet_nested4tiny_objects_keypoints_amd import ops
    g = torch.Generator().manual_seed(17)
    y = torch.randn(b, h, w, c, generator=g).to(dev)
    scale = (1 + 0.2 * torch.randn(c, generator=g)).to(dev) if affine else None
    shift = (0.3 * torch.randn(c, generator=g)).to(dev) if affine else None
    act = torch.empty_like(y)
    pooled = torch.empty(b, h // 2, w // 2, c, device=dev)
    idx = torch.empty(b, h // 2, w // 2, c, dtype=torch.uint8, device=dev)
    ops.affine_relu_pool(y, scale, shift, relu, act, pooled, idx)
    # the kernel's fused multiply-add: exact product, one rounding (via float64 here)
    want = (y.double() * scale.double() + shift.double()).float() if affine else y.clone()
    if relu:
        want = want.clamp_min(0)
    assert torch.equal(act, want)
    ref_pool, ref_idx = F.max_pool2d(want.permute(0, 3, 1, 2), 2, return_indices=True)
    assert torch.equal(pooled.permute(0, 3, 1, 2), ref_pool)
    iy, ix = ref_idx // w - 2 * torch.arange(h // 2, device=dev).view(1, 1, -1, 1), ref_idx % w - 2 * torch.arange(
        w // 2, device=dev).view(1, 1, 1, -1)
    assert torch.equal(idx.permute(0, 3, 1, 2).long(), iy * 2 + ix)
    d_pool = torch.randn(pooled.shape, generator=g).to(dev)
    base = torch.randn(y.shape, generator=g).to(dev)
    acc = base.clone()
    ops.maxpool_bwd(d_pool, idx, acc)
    ref = base.permute(0, 3, 1, 2).contiguous()
    ref.view(b, c, -1).scatter_add_(2, ref_idx.view(b, c, -1), d_pool.permute(0, 3, 1, 2).reshape(b, c, -1))
    assert torch.equal(acc.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("b,h,w,c", [(2, 8, 16, 32), (1, 6, 10, 8), (1, 7, 9, 12), (2, 4, 8, 20)])
def test_affine_relu_pool_without_winner_buffer(dev, b, h, w, c):
    """unetpp_affine_relu_pool takes pool_idx = NULL (forward-only callers need no winners): the pooled
    values and the activation are those of the call that records the winners -- row-structured, generic and odd-size
    (floor) paths."""
    from unet_nested4tiny_objects_keypoints_amd import ops
    g = torch.Generator().manual_seed(19)
    y = torch.randn(b, h, w, c, generator=g).to(dev)
    scale, shift = (1 + 0.2 * torch.randn(c, generator=g)).to(dev), (0.3 * torch.randn(c, generator=g)).to(dev)
    act0, act1 = torch.empty_like(y), torch.empty_like(y)
    p0, p1 = (torch.empty(b, h // 2, w // 2, c, device=dev) for _ in range(2))
    idx = torch.empty(b, h // 2, w // 2, c, dtype=torch.uint8, device=dev)
    ops.affine_relu_pool(y, scale, shift, True, act0, p0, idx)
    ops.affine_relu_pool(y, scale, shift, True, act1, p1, None)
    assert torch.equal(act0, act1) and torch.equal(p0, p1)
    ref = F.max_pool2d(act0.permute(0, 3, 1, 2), 2)
    assert torch.equal(p1.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("b,h,w,c", [(2, 8, 16, 32), (1, 4, 64, 128), (3, 6, 10, 8), (2, 4, 8, 20)])
def test_batchnorm_backward_with_pool_routing(dev, b, h, w, c):
    """bn_backward(pool=...) == maxpool_bwd into d_act followed by the plain bn_backward: the fused kernels where the
    shape allows (power-of-two channel quads), the two-pass fallback otherwise (c = 20)."""
    from unet_nested4tiny_objects_keypoints_amd import ops
    g = torch.Generator().manual_seed(23)
    y = (torch.randn(b, h, w, c, generator=g) * 1.5 + 0.2).to(dev)
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(dev)
    beta = (0.1 * torch.randn(c, generator=g)).to(dev)
    part = torch.stack([y.view(-1, c).sum(0), (y.view(-1, c) ** 2).sum(0)], 1).reshape(1, c, 2).contiguous()
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    mean, invstd, scale, shift = ops.bn_finalize(part.view(-1), 1, c, b * h * w, gamma, beta, 1e-5, 0.1, rm, rv)
    act = torch.empty_like(y)
    pooled = torch.empty(b, h // 2, w // 2, c, device=dev)
    idx = torch.empty(b, h // 2, w // 2, c, dtype=torch.uint8, device=dev)
    ops.affine_relu_pool(y, scale, shift, True, act, pooled, idx)
    d_act = torch.randn(y.shape, generator=g).to(dev)
    d_pool = torch.randn(pooled.shape, generator=g).to(dev)
    # reference: two passes
    ref_in = d_act.clone()
    ops.maxpool_bwd(d_pool, idx, ref_in)
    ref_dy = torch.empty_like(y)
    ref_dg, ref_db = ops.bn_backward(ref_in, y, scale, shift, mean, invstd, gamma, ref_dy)
    # fused (in place, as the engine calls it)
    got = d_act.clone()
    dg, db = ops.bn_backward(got, y, scale, shift, mean, invstd, gamma, got, pool=(d_pool, idx))
    assert rel_err(got.cpu(), ref_dy.cpu()) < 1e-5
    assert rel_err(dg.cpu(), ref_dg.cpu()) < 1e-5 and rel_err(db.cpu(), ref_db.cpu()) < 1e-5


@pytest.mark.parametrize("c,ncls", [(32, 4), (8, 5), (6, 3)])
def test_head_fwd_bwd_with_mask(dev, c, ncls):
    from unet_nested4tiny_objects_keypoints_amd import ops
    b, h, w = 2, 16, 24
    g = torch.Generator().manual_seed(7)
    x = torch.randn(b, c, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = (torch.randn(ncls, c, 1, 1, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    bias = torch.randn(ncls, generator=g, dtype=torch.float64, requires_grad=True)
    keep = (torch.rand(b, c, h, w, generator=g) < 0.6)
    for use_mask in (False, True):
        for t in (x, wt, bias):
            t.grad = None
        xin = x * keep.double() / 0.6 if use_mask else x
        out = torch.sigmoid(F.conv2d(xin, wt, bias))
        d_out = torch.randn(out.shape, generator=g, dtype=torch.float64)
        out.backward(d_out)
        xg = nhwc(x.detach().float())
        mask = keep.permute(0, 2, 3, 1).contiguous().to(torch.uint8).cuda() if use_mask else None
        p = 0.4 if use_mask else 0.0
        o = torch.empty(b, ncls, h, w, device=dev)
        wv = wt.detach().float().cuda().view(ncls, c)
        ops.head_fwd(xg, wv, bias.detach().float().cuda(), p, 0, mask, o)
        assert rel_err(o.cpu(), out.detach().float()) < TOL
        dx = torch.empty_like(xg)
        dw, db = ops.head_bwd(d_out.float().cuda(), o, xg, wv, p, 0, mask, dx, False)
        assert rel_err(nchw(dx), x.grad.float()) < TOL
        assert rel_err(dw.cpu(), wt.grad.float()) < TOL
        assert rel_err(db.cpu(), bias.grad.float()) < TOL
        # accumulate into an existing gradient, then ReLU-gate the sum by (x > 0)
        prev = torch.randn(b, h, w, c, generator=g)
        dx2 = prev.clone().cuda()
        ops.head_bwd(d_out.float().cuda(), o, xg, wv, p, 0, mask, dx2, True, gate_x=True)
        want = (prev.permute(0, 3, 1, 2) + x.grad.float()) * (x.detach() > 0)
        assert rel_err(nchw(dx2), want) < TOL


def test_head_dropout_generator_is_consistent(dev):
    """In-kernel dropout: keep rate ~ 1-p, identical mask in forward and backward, seed-dependent."""
    from unet_nested4tiny_objects_keypoints_amd import ops
    b, h, w, c, ncls = 2, 64, 64, 32, 4
    x = torch.ones(b, h, w, c, device=dev)
    wt = torch.zeros(ncls, c, device=dev)
    wt[0] = 1.0 / c
    bias = torch.zeros(ncls, device=dev)
    o = torch.empty(b, ncls, h, w, device=dev)
    ops.head_fwd(x, wt, bias, 0.4, 1234, None, o)
    logit = torch.log(o[:, 0] / (1 - o[:, 0]))        # = kept fraction / 0.6
    keep_rate = float(logit.mean()) * 0.6
    assert abs(keep_rate - 0.6) < 0.01
    # backward regenerates the same mask: dx is nonzero exactly where the element was kept
    d_out = torch.ones_like(o)
    dx = torch.empty_like(x)
    ops.head_bwd(d_out, o, x, wt, 0.4, 1234, None, dx, False)
    kept_per_pixel = (dx != 0).sum(-1).float()
    assert torch.allclose(kept_per_pixel / c / 0.6, logit, atol=1e-4)
    o2 = torch.empty_like(o)
    ops.head_fwd(x, wt, bias, 0.4, 99, None, o2)
    assert not torch.equal(o, o2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_head_dropout_generator_independence(dev, dtype):
    """one splitmix64 hash per 4 channels gives four 16-bit uniforms.  The keep flags must be
    Bernoulli(0.6) per channel position of the quad, uncorrelated between the channels of a quad, between neighbouring
    pixels, and between the seeds the three heads use (engine._dropout_config: base + odd constant * head)."""
    from unet_nested4tiny_objects_keypoints_amd import ops
    b, h, w, c, ncls = 2, 96, 96, 32, 4
    x = torch.ones(b, h, w, c, device=dev, dtype=dtype)
    wt = torch.zeros(ncls, c, device=dev)
    bias = torch.zeros(ncls, device=dev)
    o = torch.zeros(b, ncls, h, w, device=dev)

    def mask_of(seed):  # backward regenerates the mask: dx != 0 exactly where the element was kept (logit of class 0 ~ 1)
        wt.zero_()
        wt[0] = 1.0 / c
        ops.head_fwd(x, wt, bias, 0.4, seed, None, o)
        dx = torch.empty_like(x)
        ops.head_bwd(torch.ones_like(o), o, x, wt, 0.4, seed, None, dx, False)
        return (dx.float() != 0).float().view(-1, c)

    base = 0x1234567
    seeds = [(base + 0x632BE59BD9B4E019 * (j + 1)) & 0xFFFFFFFFFFFFFFFF for j in range(3)]
    masks = [mask_of(s) for s in seeds]
    n = masks[0].shape[0]
    sigma = (0.6 * 0.4 / n) ** 0.5
    for m in masks:
        rate = m.mean(0)                                    # per channel: each of the 4 positions of every quad
        assert float((rate - 0.6).abs().max()) < 5 * sigma, rate
        centred = m - 0.6
        cov = (centred.t() @ centred) / n / 0.24            # correlation matrix of the 32 channels
        off = cov - torch.diag(torch.diag(cov))
        assert float(off.abs().max()) < 5 / n ** 0.5, float(off.abs().max())       # incl. channels of one quad
        img = m.view(b, h, w, c)
        horiz = ((img[:, :, 1:] - 0.6) * (img[:, :, :-1] - 0.6)).mean() / 0.24      # neighbouring pixels
        assert abs(float(horiz)) < 5 / (n * c) ** 0.5
    for i in range(3):
        for j in range(i + 1, 3):                           # the three heads draw independent masks
            corr = ((masks[i] - 0.6) * (masks[j] - 0.6)).mean() / 0.24
            assert abs(float(corr)) < 5 / (n * c) ** 0.5, (i, j, float(corr))


@pytest.mark.parametrize("shape", [(2, 8, 8, 6), (1, 5, 7, 4), (1, 1, 1, 3)])
def test_bilinear2x_align_corners(dev, shape):
    from unet_nested4tiny_objects_keypoints_amd import ops
    b, h, w, c = shape
    g = torch.Generator().manual_seed(8)
    x = torch.randn(b, c, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    yg = torch.empty(b, 2 * h, 2 * w, c, device=dev)
    ops.bilinear2x_fwd(nhwc(x.detach().float()), yg)
    assert rel_err(nchw(yg), y.detach().float()) < TOL
    dx = torch.ones(b, h, w, c, device=dev)
    ops.bilinear2x_bwd(nhwc(dy.float()), dx, True)
    assert rel_err(nchw(dx) - 1.0, x.grad.float()) < TOL


def test_layout_converters(dev):
    from unet_nested4tiny_objects_keypoints_amd import ops
    x = torch.randn(2, 3, 8, 12)
    assert torch.equal(ops.nchw_to_nhwc(x.cuda()).cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(x.permute(0, 2, 3, 1).contiguous().cuda()).cpu(), x)


@pytest.mark.parametrize("shape", [(2, 32, 32, [32], 32), (1, 64, 64, [32, 32, 32, 32], 32), (2, 24, 40, [8, 16], 24),
                                   (1, 5, 7, [8], 8), (3, 16, 16, [64, 32], 96), (1, 128, 128, [64], 64)])
@pytest.mark.parametrize("fold", [False, True])
def test_wino_lean_and_general_kernels_agree(dev, shape, fold, monkeypatch):
    """The lean instantiations of the Winograd GEMM (buffer loads with hardware zero padding, staging inside the MFMA
    groups; mode 2 with the BatchNorm fold) and the general one are the same function of their inputs, bit for bit:
    output with bias + ReLU, gated accumulation, and the BatchNorm partial sums."""
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, cins, co = shape
    g = torch.Generator().manual_seed(17)
    srcs = [nhwc(torch.randn(b, c, h, w, generator=g)) for c in cins]
    coef = [(torch.randn(c, generator=g).to(dev), torch.randn(c, generator=g).to(dev)) for c in cins]
    wp = engine.pack_conv_fwd((torch.randn(co, sum(cins), 3, 3, generator=g) * 0.2).cuda())
    bias = torch.randn(co, generator=g).cuda()
    gate = nhwc(torch.randn(b, co, h, w, generator=g))
    prev = nhwc(torch.randn(b, co, h, w, generator=g))

    def views():
        if not fold:
            return [V(s) for s in srcs]
        return [V(s, scale=sc, shift=sh, relu=(i % 2 == 0)) if i != 1 else V(s) for i, (s, (sc, sh)) in enumerate(zip(srcs, coef))]

    res = []
    for general in (False, True):
        with ops._lib.debug_switch("WINO_NO_LEAN", 1 if general else 0):
            out = torch.empty(b, h, w, co, device=dev)
            part = torch.empty(ops.gemm_pixel_blocks(b, h, w) * co * 2, device=dev)
            ops.gemm_fwd(b, h, w, 9, views(), [V(out, relu=True)], wp, bias, part)
            acc = prev.clone()
            ops.gemm_fwd(b, h, w, 9, views(), [V(acc, gate=gate, accumulate=True)], wp)
        res.append((out, part, acc))
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c)


def test_conv3x3_view_larger_than_2gb_takes_the_general_kernel(dev):
    """The lean Winograd instantiations address a view through a buffer resource with 32-bit byte offsets (<= 2 GB);
    a 2.7 GB input (below the 2^31-element limit of the fast path) must run through the general instantiation and give
    the same numbers: checked on two crops (top-left border, last image) against the fp64 statement."""
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    b, h, w, ci, co = 5, 1024, 1024, 128, 8
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(b, h, w, ci, device=dev, generator=g)
    assert x.numel() * 4 > 2 ** 31 and x.numel() < 2 ** 31
    wt = (torch.randn(co, ci, 3, 3, generator=torch.Generator().manual_seed(4)) * 0.1)
    out = torch.empty(b, h, w, co, device=dev)
    ops.gemm_fwd(b, h, w, 9, [V(x)], [V(out)], engine.pack_conv_fwd(wt.cuda()), None)
    for (n, y0, x0) in ((0, 0, 0), (b - 1, h - 40, w - 48)):
        ys, xs = slice(max(y0 - 1, 0), min(y0 + 41, h)), slice(max(x0 - 1, 0), min(x0 + 49, w))
        crop = x[n, ys, xs].permute(2, 0, 1).unsqueeze(0).double().cpu()
        pad = (1 if x0 == 0 else 0, 1 if x0 + 48 == w else 0, 1 if y0 == 0 else 0, 1 if y0 + 40 == h else 0)
        ref = F.conv2d(F.pad(crop, pad), wt.double())
        got = out[n, y0:y0 + 40, x0:x0 + 48].permute(2, 0, 1).unsqueeze(0).cpu()
        assert rel_err(got, ref.float()) < TOL

