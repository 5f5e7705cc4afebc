"""Caller-side kernels: fused FocalLoss_BCE_2d value + gradient and on-device create_heatmap,
against the oracle restatements and the reference's own outputs recorded in the golden fixtures.  GPU only."""
import numpy as np
import pytest
import torch

from oracle.step_oracle import create_heatmap_oracle, focal_bce_2d_oracle
from tests.helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(2, 4, 16, 16), (3, 5, 24, 40), (1, 4, 7, 9), (4, 4, 64, 64)])
@pytest.mark.parametrize("gamma", [3, 2.5])
def test_focal_loss_value_and_gradient(dev, shape, gamma):
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d
    g = torch.Generator().manual_seed(3)
    pred = torch.rand(shape, generator=g) * 0.98 + 0.01
    target = torch.rand(shape, generator=g)
    pred[0, 0, 0, :3] = target[0, 0, 0, :3]          # exact hits: e = 1 + 1e-20, abs'(0) = 0
    target[0, 1, 1, 0], pred[0, 1, 1, 0] = 1.0, 0.0  # |p - t| = 1: e = 1e-20, log e = -46 -- finite
    p64 = pred.double().requires_grad_(True)
    ref = focal_bce_2d_oracle(p64, target.double(), gamma=gamma)
    ref.backward()
    pg = pred.to(dev).requires_grad_(True)
    crit = FocalLoss_BCE_2d(gamma=gamma, size_average=False)
    loss = crit(pg, target.to(dev))
    (2.0 * loss).backward()  # upstream scale flows through
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    # fp32 evaluation of 1 - |p - t| loses relative accuracy near e -> 1; compare against the gradient scale
    assert rel_err(pg.grad.cpu() / 2.0, p64.grad.float()) < 1e-4
    # the plain torch statement of the module (CPU tensors) is the same function
    cpu = crit(pred.clone().requires_grad_(True), target)
    assert abs(cpu.item() - ref.item()) <= 1e-5 * abs(ref.item())


def test_focal_loss_size_average(dev):
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d
    g = torch.Generator().manual_seed(4)
    pred, target = torch.rand(2, 4, 8, 8, generator=g), torch.rand(2, 4, 8, 8, generator=g)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=True)
    got = crit(pred.to(dev), target.to(dev)).item()
    want = crit(pred, target).item()
    assert abs(got - want) <= 1e-5 * abs(want)


def test_create_heatmap_matches_reference_fixture(dev):
    """kp/points -> kp/heatmap as the REFERENCE's create_heatmap produced them (tests/golden/make_golden.py)."""
    from unet_nested4tiny_objects_keypoints_amd import create_heatmap
    z, _ = load_golden("c1_fs4_64x64_b4_seed0")
    got = create_heatmap(z["kp/points"], 64, 64).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == z["kp/heatmap"].shape
    np.testing.assert_allclose(got, z["kp/heatmap"], rtol=0, atol=3e-7)


@pytest.mark.parametrize("case", [(3, 7, 40, 24), (2, 9, 33, 65), (1, 6, 300, 200)])
def test_create_heatmap_matches_oracle(dev, case):
    from unet_nested4tiny_objects_keypoints_amd import create_heatmap
    n, p, h, w = case
    rng = np.random.default_rng(5)
    pts = np.stack([rng.uniform(-2, w + 2, (n, p)), rng.uniform(-2, h + 2, (n, p))], axis=-1).astype(np.float32)
    got = create_heatmap(torch.from_numpy(pts).to(dev), h, w).cpu().numpy()
    want = create_heatmap_oracle(pts, h, w)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-7)


def test_fused_loss_on_hip_heads_matches_reference_loss(dev):
    """Forward of the HIP model (train mode, dropout off as in the fixture) + fused criterion on the 3 heads = the
    loss the reference computed for the same state, input and target."""
    from tests.helpers import sub
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested
    z, ctor = load_golden("c1_fs4_64x64_b4_seed0")
    model = UNet_Nested(**ctor).to(dev)
    model.load_state_dict(sub(z, "state0"))
    model.train()
    model.drop_out.p = 0.0
    outs = model(torch.from_numpy(z["x"]).to(dev))
    target = torch.from_numpy(z["target"]).to(dev)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    loss = sum(crit(o, target) for o in outs) / len(outs)
    assert abs(loss.item() - float(z["loss"])) <= 2e-5 * abs(float(z["loss"]))
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
