"""Kernel-level repeatability: the same launch, ten times, bit-identical results -- for every kernel family and shape class
of the path (3x3 GEMM forward / input gradient in its Winograd, direct, bf16 register and bf16 LDS-DMA forms; pointwise
GEMMs of the transposed / 1x1 convolutions with even and odd numbers of column tiles; weight gradients in their Winograd,
LDS-DMA, pair and quad forms), in fp32 and bf16 storage, every other launch from an idle device.

The parity tests compare one launch with a float64 statement inside a tolerance; a kernel whose result depends on timing
-- the round-4 out-of-slot weight DMA of the bf16 pointwise GEMM erred in ~1e-3 of its units, by an amount every bf16
tolerance covers -- passes them almost always.  Nothing on this path may depend on timing (no floating-point atomics, fixed
summation orders), so bit-equality across launches is the property to hold every kernel to.  Self-comparison: runs after
the parity files (tests/conftest.py).  Reference layers: models/unet.py:132,140 (3x3), :187 (transposed), :191 (1x1).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
RUNS = 10


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def bits(t):
    return t.view(torch.int16) if t.dtype == BF else t


def repeat(fn):
    """fn() -> list of tensors; RUNS launches, odd ones from an idle device; all bit-identical to the first"""
    ref = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    for i in range(1, RUNS):
        if i % 2:
            torch.cuda.synchronize()
        got = fn()
        for k, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(bits(a), bits(b)), "launch %d, output %d: %d elements differ" % (
                i, k, int((bits(a) != bits(b)).sum()))


def act(g, dev, dtype, *shape):
    return torch.randn(*shape, generator=g).to(dtype).to(dev)


GEMM_SHAPES = [
    (2, 40, 72, (32,), 32),             # one chunk (bf16) into one tile, ragged border patches
    (1, 24, 40, (64, 64, 64), 64),      # dense-skip concatenation, two column tiles
    (3, 16, 16, (32, 32), 96),          # 16 x 16 patches, three tiles
    (2, 8, 8, (128,), 32),              # 8 x 32 patches, long K into one tile
    (1, 70, 33, (32,), 64),             # odd sizes
    (2, 64, 64, (32, 32, 32, 32), 32),  # level-0 decoder shape class
    (1, 48, 96, (256,), 128),           # deep-level shape class
    (4, 192, 192, (32,), 64),           # more 512-pixel units than CUs (8-wave form of the bf16 LDS-DMA kernel)
]


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("mode", ["relu", "stats", "dgrad"])
@pytest.mark.parametrize("b,h,w,cins,cout", GEMM_SHAPES)
def test_conv3x3_gemm_is_repeatable(dev, dtype, mode, b, h, w, cins, cout):
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    g = torch.Generator().manual_seed(41)
    if mode == "dgrad":
        wt = (torch.randn(cout, sum(cins), 3, 3, generator=g) * 0.05).to(dev)
        dy = act(g, dev, dtype, b, h, w, cout)
        old = [act(g, dev, dtype, b, h, w, c) for c in cins]
        gates = [act(g, dev, dtype, b, h, w, c) for c in cins]

        def run():
            outs = [o.clone() for o in old]
            views = []
            for i, o in enumerate(outs):
                kind = i % 4
                views.append(V(o, accumulate=True, gate=gates[i], gate_sum=True) if kind == 0 else V(o, accumulate=True)
                             if kind == 1 else V(o) if kind == 2 else V(o, gate=gates[i]))
            ops.gemm_fwd(b, h, w, 9, [V(dy)], views, engine.pack_conv_dgrad(wt))
            return outs
    else:
        xs = [act(g, dev, dtype, b, h, w, c) for c in cins]
        wt = (torch.randn(cout, sum(cins), 3, 3, generator=g) * (2.0 / (9 * sum(cins))) ** 0.5).to(dev)
        bias = (torch.randn(cout, generator=g) * 0.1).to(dev)

        def run():
            y = torch.full((b, h, w, cout), float("nan"), dtype=dtype, device=dev)
            part = None
            if mode == "stats":
                part = torch.zeros(ops.gemm_pixel_blocks(b, h, w) * cout * 2, device=dev)
            ops.gemm_fwd(b, h, w, 9, [V(x) for x in xs], [V(y, relu=(mode == "relu"))], engine.pack_conv_fwd(wt), bias, part)
            return [y] + ([part] if part is not None else [])
    repeat(run)


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("b,hs,ws,ci,co", [
    (2, 12, 20, 64, 32),      # forward 4 tiles; input gradient into 64 channels (2 tiles)
    (3, 24, 40, 32, 32),      # input gradient into 32 channels: ONE column tile (the launch class of GPUTEST_r04's cause)
    (2, 64, 64, 64, 32),      # many units per workgroup
    (1, 32, 32, 256, 128),    # long K
    (2, 16, 16, 96, 96),      # three tiles (odd) both ways
])
def test_transposed_convolution_gemms_are_repeatable(dev, dtype, b, hs, ws, ci, co):
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    g = torch.Generator().manual_seed(42)
    x = act(g, dev, dtype, b, hs, ws, ci)
    wt = (torch.randn(ci, co, 2, 2, generator=g) * 0.1).to(dev)
    bias = (torch.randn(co, generator=g) * 0.1).to(dev)
    d_up = act(g, dev, dtype, b, 2 * hs, 2 * ws, co)
    old = act(g, dev, dtype, b, hs, ws, ci)

    def fwd():
        up = torch.full((b, 2 * hs, 2 * ws, co), float("nan"), dtype=dtype, device=dev)
        ops.gemm_fwd(b, hs, ws, 1, [ops.V(x)], engine._phase_views(up), engine.pack_deconv_fwd(wt), engine.tile_bias4(bias))
        return [up]

    def dgrad():
        dx = old.clone()
        ops.gemm_fwd(b, hs, ws, 1, engine._phase_views(d_up), [ops.V(dx, accumulate=True, gate=x, gate_sum=True)],
                     engine.pack_deconv_dgrad(wt))
        return [dx]

    def wgrad():
        dw, db = torch.empty(ci, co, 2, 2, device=dev), torch.empty(co, device=dev)
        ops.wgrad(b, hs, ws, 1, [ops.V(x)], engine._phase_views(d_up), dw, (0, 4 * co, 4, 1), db, n_inner=co)
        return [dw, db]
    for fn in (fwd, dgrad, wgrad):
        repeat(fn)


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("b,h,w,ci,co", [(2, 32, 32, 64, 32), (4, 128, 128, 128, 32), (2, 48, 80, 128, 96), (1, 64, 64, 64, 64)])
def test_conv1x1_of_the_bilinear_path_is_repeatable(dev, dtype, b, h, w, ci, co):
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    g = torch.Generator().manual_seed(43)
    x = act(g, dev, dtype, b, h, w, ci)
    wt = (torch.randn(co, ci, 1, 1, generator=g) * 0.1).to(dev)
    bias = (torch.randn(co, generator=g) * 0.1).to(dev)
    dy = act(g, dev, dtype, b, h, w, co)

    def fwd():
        y = torch.full((b, h, w, co), float("nan"), dtype=dtype, device=dev)
        ops.gemm_fwd(b, h, w, 1, [ops.V(x)], [ops.V(y)], engine.pack_conv_fwd(wt), bias)
        return [y]

    def dgrad():
        dx = torch.full((b, h, w, ci), float("nan"), dtype=dtype, device=dev)
        ops.gemm_fwd(b, h, w, 1, [ops.V(dy)], [ops.V(dx)], engine.pack_conv_dgrad(wt))
        return [dx]
    repeat(fwd)
    repeat(dgrad)


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("b,h,w,cis,co", [
    (2, 32, 32, (32, 32, 32), 32),     # Winograd / pair kernel, three views
    (1, 64, 64, (64,), 64),            # quad kernel (bf16)
    (2, 24, 40, (32,), 32),            # ragged patches
    (1, 48, 48, (128, 64), 64),        # quad kernel, two views
    (3, 16, 16, (64,), 128),           # image narrower than 17: the direct kernels (fp32)
    (2, 96, 96, (32, 32), 64),         # many tiles per workgroup
])
def test_conv3x3_weight_gradient_is_repeatable(dev, dtype, b, h, w, cis, co):
    from unet_nested4tiny_objects_keypoints_amd import ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    g = torch.Generator().manual_seed(44)
    xs = [act(g, dev, dtype, b, h, w, c) for c in cis]
    dy = act(g, dev, dtype, b, h, w, co)
    ci = sum(cis)

    def run():
        dw, db = torch.empty(co, ci, 3, 3, device=dev), torch.empty(co, device=dev)
        ops.wgrad(b, h, w, 9, [V(x) for x in xs], [V(dy)], dw, (1, 9, ci * 9, 0), db)
        return [dw, db]
    repeat(run)
    if dtype == torch.float32:   # the folded-BatchNorm x view of the encoder's conv2 (its own instantiations)
        sc = (torch.rand(cis[0], generator=g) + 0.5).to(dev)
        sh = (torch.randn(cis[0], generator=g) * 0.1).to(dev)

        def run_fold():
            dw, db = torch.empty(co, cis[0], 3, 3, device=dev), torch.empty(co, device=dev)
            ops.wgrad(b, h, w, 9, [V(xs[0], scale=sc, shift=sh, relu=True)], [V(dy)], dw, (1, 9, cis[0] * 9, 0), db)
            return [dw, db]
        repeat(run_fold)
