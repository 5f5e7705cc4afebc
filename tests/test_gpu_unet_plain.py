"""The reference's classic U-Net (models/unet.py:8-117) on the HIP path, through the C ABI:
  (1) golden fixtures produced by the reference itself (a narrow instance wired from its own blocks, and its UNet()
      with a seeded state), (2) the CPU oracle on fresh inputs at larger shapes (gradients against the float64 oracle
      run with the HIP forward's ReLU gates and max-pool winners, tests/helpers.py).
GPU only.  Bar: <= 1e-4 relative fp32."""
import pytest
import torch

from tests.helpers import check_flips, install_hip_gates_plain, load_golden, rel_err, seeded_state, sub
from tests.test_oracle_golden import plain_grads_close

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _state_of(z, ctor):
    from oracle.unet_plain_oracle import UNetOracle
    seed = int(z["meta/seeded_state"])
    return seeded_state(UNetOracle(**ctor), seed) if seed >= 0 else sub(z, "state0")


def _hip(ctor, state, dev):
    from unet_nested4tiny_objects_keypoints_amd import UNet
    m = UNet(**ctor)
    m.load_state_dict(state, strict=True)
    return m.to(dev)


def _gate_aware_grads(m, ctor, state, x, target, label, dtype=torch.float32):
    from oracle.step_oracle import focal_bce_2d_oracle
    from oracle.unet_plain_oracle import UNetOracle
    ref = UNetOracle(**ctor)
    ref.load_state_dict(state)
    ref = ref.to(dtype).train()
    gated = install_hip_gates_plain(ref, m._debug_saved)
    out = ref(x.to(dtype))
    focal_bce_2d_oracle(out, target.to(dtype)).backward()
    flips = check_flips(gated, label)
    return flips, {k: p.grad for k, p in ref.named_parameters()}


def _pre_bn_bias(k):
    return k.endswith(".bias") and ".conv." in k and k.split(".")[-2] in ("0", "3")


@pytest.mark.parametrize("name", ["unet_w8_rgb5_32x48_b2", "unet_ref_rgb5_64x64_b1", "unet_w8_rgb5_40x56_b2"])
def test_plain_unet_golden(dev, name):
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d
    z, ctor = load_golden(name)
    state = _state_of(z, ctor)
    m = _hip(ctor, state, dev).eval()
    x, target = torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["target"]).to(dev)
    with torch.no_grad():
        out = m(x)
    assert out.shape == z["eval_out/0"].shape and rel_err(out.cpu(), z["eval_out/0"]) < TOL
    m.train()
    m._debug_keep_saved = True
    out = m(x)
    loss = FocalLoss_BCE_2d(gamma=3, size_average=False)(out, target)  # trainer/trainer.py:133 (non-tuple output)
    loss.backward()
    assert rel_err(out.detach().cpu(), z["train_out/0"]) < TOL
    assert abs(float(loss.detach()) - float(z["loss"])) <= TOL * abs(float(z["loss"]))
    got = {k: p.grad.cpu() for k, p in m.named_parameters()}
    flips, want = _gate_aware_grads(m, ctor, state, x.cpu(), target.cpu(), "plain-golden:" + name)
    bad = [(k, rel_err(got[k], w)) for k, w in want.items() if not _pre_bn_bias(k) and not rel_err(got[k], w) < TOL]
    assert not bad, bad
    if flips == 0:  # then the reference's own gradients must match as well
        plain_grads_close(got, z, TOL)
    bufs = sub(z, "state1_buffers")
    for k, b in m.named_buffers():
        if b.dtype.is_floating_point:
            assert rel_err(b.cpu(), bufs[k]) < TOL, k
        else:
            assert int(b) == int(bufs[k]), k


@pytest.mark.parametrize("case", [
    (dict(n_classes=5, n_channels=3, widths=(16, 32, 64, 128, 128)), 2, 64, 96, torch.float32),
    (dict(n_classes=5, n_channels=3), 2, 128, 128, torch.float64),   # the reference's widths at a real size
    (dict(n_classes=4, n_channels=1, widths=(32, 64, 128, 256, 256)), 1, 256, 256, torch.float64),
], ids=lambda c: "w%d-b%d-%dx%d" % (c[0].get("widths", (64,))[0], c[1], c[2], c[3]))
def test_plain_unet_train_step_vs_oracle(dev, case):
    from oracle.step_oracle import focal_bce_2d_oracle
    from oracle.unet_plain_oracle import UNetOracle
    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d
    ctor, b, h, w, grad_dtype = case
    torch.manual_seed(41)
    ref = UNetOracle(**ctor).train()
    state = seeded_state(ref, 42)
    ref.load_state_dict(state)
    x = torch.randn(b, ctor["n_channels"], h, w)
    target = torch.rand(b, ctor["n_classes"], h, w)
    m = _hip(ctor, state, dev).train()
    m._debug_keep_saved = True
    with torch.no_grad():
        ro = ref(x)  # advances the oracle's running statistics once, like the HIP step below
        rl = focal_bce_2d_oracle(ro, target)
    out = m(x.to(dev))
    loss = FocalLoss_BCE_2d(gamma=3, size_average=False)(out, target.to(dev))
    loss.backward()
    assert rel_err(out.detach().cpu(), ro) < TOL
    assert abs(float(loss.detach()) - float(rl)) <= TOL * abs(float(rl))
    got = {k: p.grad.cpu() for k, p in m.named_parameters()}
    _, want = _gate_aware_grads(m, ctor, state, x, target, "plain-oracle:%s b%d %dx%d" % (sorted(ctor.items()), b, h, w),
                                dtype=grad_dtype)
    bad = [(k, rel_err(got[k], g)) for k, g in want.items() if not _pre_bn_bias(k) and not rel_err(got[k], g) < TOL]
    assert not bad, bad
    for (k, bh), (_, br) in zip(m.named_buffers(), ref.named_buffers()):
        if bh.dtype.is_floating_point:
            assert rel_err(bh.cpu(), br) < TOL, k


def test_plain_unet_module_protocol(dev):
    """state_dict round trip under the reference's key names, input gradient, eval determinism, error behaviour."""
    from oracle.unet_plain_oracle import UNetOracle
    from unet_nested4tiny_objects_keypoints_amd import UNet
    ctor = dict(n_classes=5, n_channels=3, widths=(8, 16, 32, 64, 64))
    ref = UNetOracle(**ctor).train()
    ref.load_state_dict(seeded_state(ref, 5))
    m = _hip(ctor, ref.state_dict(), dev).train()
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(6))
    xr, xg = x.clone().requires_grad_(True), x.to(dev).requires_grad_(True)
    ref(xr).square().sum().backward()
    m(xg).square().sum().backward()
    assert rel_err(xg.grad.cpu(), xr.grad) < TOL
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    UNetOracle(**ctor).load_state_dict(sd, strict=True)
    assert list(sd.keys()) == list(ref.state_dict().keys())
    m.eval()
    with torch.no_grad():
        a, b = m(x.to(dev)), m(x.to(dev))
    assert torch.equal(a, b) and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert UNet().n_classes == 5 and UNet().n_channels == 3   # zero-argument constructor (models/unet.py:95)
    with pytest.raises(ValueError):
        m(torch.randn(1, 3, 12, 32, device=dev))   # four poolings need at least 16 pixels a side
    with torch.no_grad():                          # sizes that are not multiples of 16: floor pooling + zero padding
        odd = torch.randn(1, 3, 24, 37)
        ref.eval()
        assert rel_err(m(odd.to(dev)).cpu(), ref(odd)) < TOL
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 3, 32, 32))               # CPU tensor: no fallback
