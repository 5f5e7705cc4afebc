"""bf16-storage twins (BASELINE configs[3]/[4]; UNETPP_GEMM_BF16) through the C ABI.

Kernel level: every bf16 entry point against a float64 PyTorch statement of the same operation evaluated on the SAME
bf16-rounded operands (inputs, weights rounded to bf16 as the kernels' weight images are; fp32 bias / BatchNorm
coefficients).  What remains is the kernel's own arithmetic: fp32 accumulation (<= 1e-5 of the result scale) and ONE
rounding of the stored result to bf16 (half an ulp = 2^-9 relative), so the bar is
    |got - want| <= 2^-8 * |want| + 1e-5 * max|want|                    (TOL_ULP)
for bf16 outputs and 1e-4 relative for fp32 outputs (weight gradients, statistics).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rb(t):
    """round to bf16, keep as float64 (the operand values the kernels see)"""
    return t.to(BF).double()


def close_bf16(got, want, what=""):
    got, want = got.double().cpu(), want.double().cpu()
    err = (got - want).abs()
    bound = 2.0 ** -8 * want.abs() + 1e-5 * float(want.abs().max())
    bad = err > bound
    assert not bool(bad.any()), (what, int(bad.sum()), float(err.max()), float(want.abs().max()))


def close_f32(got, want, tol=1e-4, what=""):
    got, want = got.double().cpu(), want.double().cpu()
    assert float((got - want).abs().max()) <= tol * float(want.abs().max()) + 1e-30, (
        what, float((got - want).abs().max()), float(want.abs().max()))


def nhwc(t):  # NCHW float64 -> NHWC
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("b,h,w,cins,cout,affine", [
    (2, 16, 16, (32,), 32, False),
    (1, 24, 40, (64, 32), 64, False),      # two-view concatenation, ragged patches
    (2, 8, 8, (16, 8, 40), 24, False),     # partial chunks and column tiles
    (1, 32, 32, (32,), 64, True),          # BatchNorm apply + ReLU folded into the load
])
def test_conv3x3_bf16_forward(dev, b, h, w, cins, cout, affine):
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(b, c, h, w, generator=g) for c in cins]
    wt = torch.randn(cout, sum(cins), 3, 3, generator=g) * (2.0 / (9 * sum(cins))) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    scale = (1 + 0.2 * torch.randn(cins[0], generator=g)) if affine else None
    shift = (0.3 * torch.randn(cins[0], generator=g)) if affine else None
    xd = [nhwc(x).to(BF).to(dev) for x in xs]
    views = [V(t) for t in xd]
    xin = [rb(x) for x in xs]
    if affine:
        views[0] = V(xd[0], scale=scale.to(dev), shift=shift.to(dev), relu=True)
        a = torch.relu(xin[0].float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))  # fp32 fma, then bf16
        xin[0] = rb((xin[0] * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).clamp_min(0).float())
        del a
    y = torch.empty(b, h, w, cout, dtype=BF, device=dev)
    blocks = ops.gemm_pixel_blocks(b, h, w)
    partial = torch.empty(blocks * cout * 2, device=dev)
    wd = wt.to(dev)
    ops.gemm_fwd(b, h, w, 9, views, [V(y, relu=True)], engine.pack_conv_fwd(wd), bias.to(dev), partial)
    want = F.conv2d(torch.cat(xin, 1), rb(wt), bias.double(), padding=1).clamp_min(0)
    close_bf16(y.permute(0, 3, 1, 2), want, "conv output")
    # BatchNorm partial sums describe the STORED tensor
    part = partial.view(blocks, cout, 2).double().sum(0).cpu()
    yd = y.double().cpu()
    close_f32(part[:, 0], yd.sum((0, 1, 2)), 1e-4, "sum")
    close_f32(part[:, 1], (yd * yd).sum((0, 1, 2)), 1e-4, "sum of squares")
    assert ops._lib.lib().unetpp_last_kernel_name() == b"gemm_bf16_kernel<9>"


def test_conv3x3_bf16_input_gradient_targets(dev):
    """dgrad form: rotated weights, three output views: plain store, accumulate, accumulate + gate of the sum."""
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    g = torch.Generator().manual_seed(2)
    b, h, w, co = 2, 16, 24, 32
    cis = (32, 16, 48)
    dy = torch.randn(b, co, h, w, generator=g)
    wt = torch.randn(co, sum(cis), 3, 3, generator=g) * 0.05
    old = [torch.randn(b, c, h, w, generator=g) for c in cis]
    gate = torch.randn(b, cis[2], h, w, generator=g)
    outs = [nhwc(o).to(BF).to(dev) for o in old]
    gate_d = nhwc(gate).to(BF).to(dev)
    ops.gemm_fwd(b, h, w, 9, [V(nhwc(dy).to(BF).to(dev))],
                 [V(outs[0]), V(outs[1], accumulate=True), V(outs[2], accumulate=True, gate=gate_d, gate_sum=True)],
                 engine.pack_conv_dgrad(wt.to(dev)))
    full = F.conv_transpose2d(rb(dy), rb(wt), padding=1)  # = conv(dy, rot180(w)^T)
    parts = torch.split(full, cis, 1)
    close_bf16(outs[0].permute(0, 3, 1, 2), parts[0], "store")
    # the contribution itself is rounded to bf16 before it is added (it passes through the same epilogue)
    want1 = rb(parts[1].float()) + rb(old[1])
    close_bf16(outs[1].permute(0, 3, 1, 2), want1, "accumulate")
    want2 = (rb(parts[2].float()) + rb(old[2])) * (rb(gate) > 0)
    close_bf16(outs[2].permute(0, 3, 1, 2), want2, "gate of the sum")


def test_deconv2x2_bf16_forward_and_input_gradient(dev):
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    g = torch.Generator().manual_seed(3)
    b, hs, ws, ci, co = 2, 12, 20, 64, 32
    x = torch.randn(b, ci, hs, ws, generator=g)
    wt = torch.randn(ci, co, 2, 2, generator=g) * 0.1
    bias = torch.randn(co, generator=g) * 0.1
    up = torch.empty(b, 2 * hs, 2 * ws, co, dtype=BF, device=dev)
    ops.gemm_fwd(b, hs, ws, 1, [V(nhwc(x).to(BF).to(dev))], engine._phase_views(up), engine.pack_deconv_fwd(wt.to(dev)),
                 engine.tile_bias4(bias.to(dev)))
    want = F.conv_transpose2d(rb(x), rb(wt), bias.double(), stride=2)
    close_bf16(up.permute(0, 3, 1, 2), want, "deconv forward")
    assert ops._lib.lib().unetpp_last_kernel_name() == b"gemm_bf16_kernel<1>"
    d_up = torch.randn(b, co, 2 * hs, 2 * ws, generator=g)
    d_up_d = nhwc(d_up).to(BF).to(dev)
    dx = torch.empty(b, hs, ws, ci, dtype=BF, device=dev)
    ops.gemm_fwd(b, hs, ws, 1, engine._phase_views(d_up_d), [V(dx)], engine.pack_deconv_dgrad(wt.to(dev)))
    want_dx = F.conv2d(rb(d_up), rb(wt), stride=2)  # [ci, co, 2, 2] read as an (out = ci, in = co) stride-2 convolution
    close_bf16(dx.permute(0, 3, 1, 2), want_dx, "deconv input gradient")


@pytest.mark.parametrize("b,h,w,cis,co,affine", [
    (2, 16, 16, (32,), 32, False),
    (3, 24, 40, (64, 32), 64, False),
    (2, 8, 16, (16, 40), 24, False),
    (2, 32, 32, (32,), 32, True),
])
def test_conv3x3_bf16_weight_gradient(dev, b, h, w, cis, co, affine):
    from unet_nested4tiny_objects_keypoints_amd import ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    g = torch.Generator().manual_seed(4)
    xs = [torch.randn(b, c, h, w, generator=g) for c in cis]
    dy = torch.randn(b, co, h, w, generator=g)
    scale = (1 + 0.2 * torch.randn(cis[0], generator=g)) if affine else None
    shift = (0.3 * torch.randn(cis[0], generator=g)) if affine else None
    xd = [nhwc(x).to(BF).to(dev) for x in xs]
    views = [V(t) for t in xd]
    xin = [rb(x) for x in xs]
    if affine:
        views[0] = V(xd[0], scale=scale.to(dev), shift=shift.to(dev), relu=True)
        xin[0] = rb((xin[0] * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).clamp_min(0).float())
    ci = sum(cis)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    db = torch.empty(co, device=dev)
    ops.wgrad(b, h, w, 9, views, [V(nhwc(dy).to(BF).to(dev))], dw, (1, 9, ci * 9, 0), db)
    assert ops._lib.lib().unetpp_last_kernel_name() == b"wgrad_bf16_kernel<9>"
    xcat, dyr = torch.cat(xin, 1), rb(dy)
    want = torch.nn.grad.conv2d_weight(xcat, (co, ci, 3, 3), dyr, padding=1)
    close_f32(dw, want, 1e-4, "dW")
    close_f32(db, dyr.sum((0, 2, 3)), 1e-4, "db")


def test_deconv2x2_bf16_weight_gradient(dev):
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    g = torch.Generator().manual_seed(5)
    b, hs, ws, ci, co = 2, 12, 20, 64, 32
    x = torch.randn(b, ci, hs, ws, generator=g)
    d_up = torch.randn(b, co, 2 * hs, 2 * ws, generator=g)
    dw = torch.empty(ci, co, 2, 2, device=dev)
    db = torch.empty(co, device=dev)
    ops.wgrad(b, hs, ws, 1, [V(nhwc(x).to(BF).to(dev))], engine._phase_views(nhwc(d_up).to(BF).to(dev)), dw,
              (0, 4 * co, 4, 1), db, n_inner=co)
    xr, dr = rb(x), rb(d_up)
    want = torch.einsum("nchw,nkhawb->ckab", xr, dr.view(b, co, hs, 2, ws, 2))
    close_f32(dw, want, 1e-4, "deconv dW")
    close_f32(db, dr.sum((0, 2, 3)), 1e-4, "deconv db")


def test_first_layer_bf16(dev):
    """1..3-channel fp32 network input -> bf16 activations (VALU kernel), and its weight gradient from a bf16 dy."""
    from unet_nested4tiny_objects_keypoints_amd import engine, ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    g = torch.Generator().manual_seed(6)
    for cin in (1, 3):
        b, h, w, co = 2, 24, 40, 32
        x = torch.randn(b, cin, h, w, generator=g)
        wt = torch.randn(co, cin, 3, 3, generator=g) * 0.3
        bias = torch.randn(co, generator=g) * 0.1
        xd = nhwc(x).to(dev)
        y = torch.empty(b, h, w, co, dtype=BF, device=dev)
        blocks = ops.gemm_pixel_blocks(b, h, w)
        partial = torch.empty(blocks * co * 2, device=dev)
        ops.gemm_fwd(b, h, w, 9, [V(xd)], [V(y)], engine.pack_conv_fwd(wt.to(dev)), bias.to(dev), partial)
        assert ops._lib.lib().unetpp_last_kernel_name() == b"small_cin_fwd_kernel"
        want = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
        close_bf16(y.permute(0, 3, 1, 2), want, "first layer")
        part = partial.view(blocks, co, 2).double().sum(0).cpu()
        close_f32(part[:, 0], y.double().cpu().sum((0, 1, 2)), 1e-4)
        dy = torch.randn(b, co, h, w, generator=g)
        dw = torch.empty(co, cin, 3, 3, device=dev)
        db = torch.empty(co, device=dev)
        ops.wgrad(b, h, w, 9, [V(xd)], [V(nhwc(dy).to(BF).to(dev))], dw, (1, 9, cin * 9, 0), db)
        want_dw = torch.nn.grad.conv2d_weight(x.double(), (co, cin, 3, 3), rb(dy), padding=1)
        close_f32(dw, want_dw, 1e-4, "first layer dW")
        close_f32(db, rb(dy).sum((0, 2, 3)), 1e-4)
