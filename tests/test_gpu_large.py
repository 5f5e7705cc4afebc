"""Tensors above 2 GB through the weight-gradient kernels that address their operands with per-image buffer resources
(csrc/wgrad_bf16.hip, csrc/wgrad_wino.hip): a buffer offset is 31 bits, so the resource is rebuilt per IMAGE and only
an image has to stay below 2 GB.  Size-independent property instead of an oracle (the oracle would take hours here):
the gradient over the whole batch equals the sum of the gradients over its two halves, each of which is below 2 GB and
starts at offset zero of its own tensor view.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _wgrad(ops, V, x, dy, direct=False):
    b, h, w, ci = x.shape
    co = dy.shape[3]
    dw = torch.empty(co, ci, 3, 3, device=x.device)
    db = torch.empty(co, device=x.device)
    ops.wgrad(b, h, w, 9, [V(x)], [V(dy)], dw, (1, 9, ci * 9, 0), db, direct=direct)
    return dw, db, ops._lib.lib().unetpp_last_kernel_name().decode()


@pytest.mark.parametrize("dtype,b,hw,c,kernel", [
    (torch.bfloat16, 36, 512, 128, "wgrad_bf16_quad_kernel<9>"),   # 2.4 GB per operand, 64-multiples: the quad kernel
    (torch.bfloat16, 48, 512, 96, "wgrad_bf16_kernel<9>"),         # 2.4 GB per operand, 96 channels: the pair kernel
    (torch.float32, 36, 512, 64, "wgrad_wino_kernel"),             # 2.4 GB per operand, fp32 Winograd weight gradient
])
def test_weight_gradient_over_2gb_equals_sum_of_halves(dev, dtype, b, hw, c, kernel):
    from unet_nested4tiny_objects_keypoints_amd import ops
    from unet_nested4tiny_objects_keypoints_amd.ops import V
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.empty(b, hw, hw, c, device=dev, dtype=dtype)
    dy = torch.empty(b, hw, hw, c, device=dev, dtype=dtype)
    for t in (x, dy):   # filled image by image: torch.randn of the whole tensor would need a 4-byte copy of it
        for i in range(b):
            t[i] = torch.randn(hw, hw, c, device=dev, generator=g).to(dtype)
    assert x.numel() * x.element_size() > 2 ** 31
    dw, db, name = _wgrad(ops, V, x, dy)
    assert name == kernel
    h = b // 2
    dw0, db0, _ = _wgrad(ops, V, x[:h], dy[:h])
    dw1, db1, _ = _wgrad(ops, V, x[h:], dy[h:])
    torch.cuda.synchronize()
    scale = float(dw.abs().max())
    assert float((dw - (dw0 + dw1)).abs().max()) <= 2e-5 * scale
    assert float((db - (db0 + db1)).abs().max()) <= 2e-5 * float(db.abs().max()) + 1e-3
