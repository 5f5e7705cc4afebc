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


@pytest.mark.parametrize("heads,shape,gamma,size_average", [(3, (4, 4, 64, 64), 3, False), (4, (2, 5, 24, 40), 3, False),
                                                            (2, (1, 4, 7, 9), 2.5, False), (8, (2, 4, 16, 16), 3, True)])
def test_mean_over_heads_is_the_trainers_loop_bit_for_bit(dev, heads, shape, gamma, size_average):
    """FocalLoss_BCE_2d.mean_over_heads (one launch: value, mean over heads, gradients) against the loop body of
    trainer/trainer.py:122-135 run with the same module and autograd: same loss bits, same gradient bits."""
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d
    g = torch.Generator().manual_seed(11)
    target = torch.rand(shape, generator=g).to(dev)
    preds = [(torch.rand(shape, generator=g) * 0.98 + 0.01).to(dev).requires_grad_(True) for _ in range(heads)]
    crit = FocalLoss_BCE_2d(gamma=gamma, size_average=size_average)
    avg = 0
    for p in preds:
        avg = avg + crit(p, target)
    avg = 1.0 * avg / len(preds)
    avg.backward()
    got = crit.mean_over_heads(tuple(p.detach().requires_grad_(True) for p in preds), target)
    assert got is not None
    loss, grads = got
    assert loss.dim() == 0 and torch.equal(loss, avg.detach())
    for p, gr in zip(preds, grads):
        assert torch.equal(p.grad, gr)
    # heads that do not qualify fall back to the loop (None)
    assert crit.mean_over_heads(tuple(p.detach().cpu() for p in preds), target.cpu()) is None
    assert crit.mean_over_heads((preds[0].detach(),), target) is None


def test_train_step_with_fused_head_loss_equals_the_written_loop(dev):
    """train_step (fused loss + direct start of the network's backward) against the loop body written out with autograd:
    same loss, same parameter gradients, bit for bit -- on a deep-supervision network with a transposed-convolution up path
    (its four-phase bias rows and the grouped input-gradient weights come from the pack plan's one copy launch from the
    second step on: both steps are compared)."""
    import copy
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, train_step
    torch.manual_seed(5)
    model = UNet_Nested(in_channels=1, n_classes=4, feature_scale=4, is_deconv=True, is_batchnorm=True, is_ds=True).to(dev).train()
    model.drop_out.p = 0.0
    twin = copy.deepcopy(model)
    x = torch.randn(2, 1, 64, 64, device=dev)
    target = torch.rand(2, 4, 64, 64, device=dev)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    opt2 = torch.optim.SGD(twin.parameters(), lr=0.05)
    for step in range(3):
        outs, loss = train_step(model, opt, crit, x, target)
        opt2.zero_grad()
        outs2 = twin(x)
        assert isinstance(outs2, tuple) and len(outs2) >= 2
        avg = 0
        for o in outs2:
            avg = avg + crit(o, target)
        avg = 1.0 * avg / len(outs2)
        avg.backward()
        assert torch.equal(loss, avg.detach()), step
        for (n, p), q in zip(model.named_parameters(), twin.parameters()):
            assert torch.equal(p.grad, q.grad), (step, n)
        opt2.step()
    plan = model.__dict__["_pack_plan"]
    assert plan.copy_launches >= 4 and len(plan.copies) > 0   # (fwd + bwd of steps 2 and 3)
    assert any(k[0] == "bias4" for k in plan.copies) and any(k[0] == "grouped" for k in plan.copies)


def test_copy_jobs_kernel(dev):
    """unetpp_copy_jobs: strided row copies (vector and scalar paths, broadcast rows, ragged sizes) in one launch"""
    import ctypes as C
    from unet_nested4tiny_objects_keypoints_amd import _lib
    from unet_nested4tiny_objects_keypoints_amd.ops import _ptr, _stream, check
    g = torch.Generator().manual_seed(2)
    cases = [(4, 32, 0, 32), (32, 288, 864, 288), (7, 9, 27, 11), (1, 5, 5, 5), (64, 64 * 9, 192 * 9, 64 * 9), (3, 1000, 1003, 1001)]
    jobs, want, dsts = [], [], []
    for (no, ni, ss, ds) in cases:
        src = torch.randn(max(1, (no - 1) * ss + ni) + 3, generator=g).to(dev)
        dst = torch.full(((no - 1) * ds + ni + 5,), -7.0, device=dev)
        ref = dst.clone()
        for o in range(no):
            ref[o * ds:o * ds + ni] = src[o * ss:o * ss + ni]
        j = _lib.CopyJob()
        j.src, j.dst, j.n_outer, j.n_inner, j.src_stride, j.dst_stride = src.data_ptr(), dst.data_ptr(), no, ni, ss, ds
        jobs.append(j)
        want.append(ref)
        dsts.append((dst, src))
    arr = (_lib.CopyJob * len(jobs))(*jobs)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    check(_lib.lib().unetpp_copy_jobs(_ptr(table), len(jobs), max(c[0] * c[1] for c in cases), _stream()), "unetpp_copy_jobs")
    torch.cuda.synchronize()
    for (dst, _), ref in zip(dsts, want):
        assert torch.equal(dst, ref)
    assert _lib.lib().unetpp_copy_jobs(None, 1, 4, None) == -1
    assert _lib.lib().unetpp_copy_jobs(_ptr(table), 0, 4, None) == -1
