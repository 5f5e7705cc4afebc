"""The oracle (CPU restatement) replayed against fixtures produced by the reference itself.

Pins oracle/unet_nested_oracle.py and oracle/step_oracle.py to
/root/reference/models/unet.py, tools/losses/focal_loss.py, tools/misc/helper.py
(see tests/golden/make_golden.py).  CPU only.
"""
import numpy as np
import pytest
import torch

from oracle.step_oracle import create_heatmap_oracle, focal_bce_2d_oracle, train_step_oracle
from oracle.unet_nested_oracle import UNetNestedOracle
from tests.helpers import DEPTH_CASES, GOLDEN_CASES, assert_grads_close, is_pre_bn_bias, load_golden, rel_err, sub

TOL_FWD = 2e-6   # same library on both sides; only thread-count / ISA reduction order may differ
TOL_GRAD = 2e-5


def _build(z, ctor):
    model = UNetNestedOracle(**ctor)
    state = sub(z, "state0")
    if ctor.get("depth", 4) == 4:
        assert list(model.state_dict().keys()) == list(state.keys()), "state-dict key order/names differ"
    else:  # the fixture-time level-5 subclass registers its extra modules last: names must agree, order cannot
        assert sorted(model.state_dict().keys()) == sorted(state.keys()), "state-dict key names differ"
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(state[k].shape), k
    model.load_state_dict(state)
    return model


@pytest.mark.parametrize("name", GOLDEN_CASES + DEPTH_CASES)
def test_eval_forward_matches_reference(name):
    z, ctor = load_golden(name)
    model = _build(z, ctor).eval()
    with torch.no_grad():
        outs = model(torch.from_numpy(z["x"]))
    assert isinstance(outs, tuple) and len(outs) == ctor.get("depth", 4) - 1
    for i, o in enumerate(outs):
        assert rel_err(o, z["eval_out/%d" % i]) < TOL_FWD


@pytest.mark.parametrize("name", GOLDEN_CASES + DEPTH_CASES)
def test_train_step_matches_reference(name):
    z, ctor = load_golden(name)
    model = _build(z, ctor).train()
    model.drop_out.eval()
    x, target = torch.from_numpy(z["x"]), torch.from_numpy(z["target"])
    outs = model(x)
    loss = sum(focal_bce_2d_oracle(o, target) for o in outs) / len(outs)
    loss.backward()
    for i, o in enumerate(outs):
        assert rel_err(o.detach(), z["train_out/%d" % i]) < TOL_FWD
    assert abs(float(loss) - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    assert_grads_close({k: p.grad for k, p in model.named_parameters()}, sub(z, "grad"), ctor, TOL_GRAD)
    bufs = sub(z, "state1_buffers")
    for k, b in model.named_buffers():
        if b.dtype.is_floating_point:
            assert rel_err(b, bufs[k]) < TOL_FWD, k
        else:
            assert int(b) == int(bufs[k]), k


@pytest.mark.parametrize("opt_name", ["adam", "sgd"])
def test_optimizer_step_matches_reference(opt_name):
    z, ctor = load_golden("c1_fs4_64x64_b4_seed0")
    model = _build(z, ctor).train()
    model.drop_out.eval()
    if opt_name == "adam":
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    else:
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    train_step_oracle(model, opt, torch.from_numpy(z["x"]), torch.from_numpy(z["target"]))
    after = sub(z, "after_" + opt_name)
    for k, p in model.named_parameters():
        # Adam's first step moves every weight by ~lr regardless of gradient size, so compare
        # the update itself with an absolute bound well below lr.
        if opt_name == "adam" and is_pre_bn_bias(k, ctor):
            continue  # zero-gradient parameter: Adam turns the sign of rounding noise into +-lr
        assert float((p.detach() - after[k]).abs().max()) < 2e-5, k


def test_create_heatmap_matches_reference():
    z, _ = load_golden("c1_fs4_64x64_b4_seed0")
    got = create_heatmap_oracle(z["kp/points"], 64, 64)
    assert got.dtype == np.float32 and got.shape == z["kp/heatmap"].shape
    np.testing.assert_allclose(got, z["kp/heatmap"], rtol=0, atol=2e-7)


@pytest.mark.parametrize("depth", [2, 3, 5])
def test_generalised_depth_shapes(depth):
    model = UNetNestedOracle(in_channels=1, n_classes=4, feature_scale=8, depth=depth).eval()
    side = 2 ** (depth - 1) * 2
    with torch.no_grad():
        outs = model(torch.randn(1, 1, side, side))
    assert len(outs) == depth - 1
    assert all(o.shape == (1, 4, side, side) for o in outs)
    assert ("up_concat04.up.weight" in model.state_dict()) == (depth == 5)


# ---------------------------------------------------------------- classic U-Net
def _plain_from_golden(name):
    from oracle.unet_plain_oracle import UNetOracle
    from tests.helpers import seeded_state
    z, ctor = load_golden(name)
    model = UNetOracle(**ctor)
    seed = int(z["meta/seeded_state"])
    state = seeded_state(model, seed) if seed >= 0 else sub(z, "state0")
    assert list(model.state_dict().keys()) == list(state.keys()), "state-dict key order/names differ"
    model.load_state_dict(state)
    return z, ctor, model


def plain_grads_close(got, z, tol):
    """got: name -> gradient tensor; z: fixture with full ('grad/') or subsampled ('grad_sub/', 'grad_norm/') gradients."""
    from tests.helpers import GRAD_STRIDE
    bad = []
    for k in [f[len("grad/"):] for f in z.files if f.startswith("grad/")]:
        if k.endswith(".bias") and ".conv." in k and k.split(".")[-2] in ("0", "3"):
            continue  # conv bias in front of a BatchNorm: analytically zero gradient (tests/helpers.is_pre_bn_bias)
        if not rel_err(got[k], z["grad/" + k]) < tol:
            bad.append((k, rel_err(got[k], z["grad/" + k])))
    for k in [f[len("grad_sub/"):] for f in z.files if f.startswith("grad_sub/")]:
        sub_got = got[k].reshape(-1)[::GRAD_STRIDE]
        scale = float(z["grad_norm/" + k]) / got[k].numel() ** 0.5  # rms of the whole gradient
        err = float((sub_got.double() - torch.from_numpy(z["grad_sub/" + k]).double()).abs().max()) / scale
        nrm = abs(float(got[k].double().norm()) - float(z["grad_norm/" + k])) / float(z["grad_norm/" + k])
        if not (err < 10 * tol and nrm < tol):
            bad.append((k, err, nrm))
    assert not bad, bad


@pytest.mark.parametrize("name", ["unet_w8_rgb5_32x48_b2", "unet_ref_rgb5_64x64_b1"])
def test_plain_unet_oracle_matches_reference(name):
    z, ctor, model = _plain_from_golden(name)
    model.eval()
    with torch.no_grad():
        out = model(torch.from_numpy(z["x"]))
    assert rel_err(out, z["eval_out/0"]) < TOL_FWD
    model.train()
    out = model(torch.from_numpy(z["x"]))
    loss = focal_bce_2d_oracle(out, torch.from_numpy(z["target"]))
    loss.backward()
    assert rel_err(out.detach(), z["train_out/0"]) < TOL_FWD
    assert abs(float(loss) - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    plain_grads_close({k: p.grad for k, p in model.named_parameters()}, z, TOL_GRAD)
    bufs = sub(z, "state1_buffers")
    for k, b in model.named_buffers():
        if b.dtype.is_floating_point:
            assert rel_err(b, bufs[k]) < TOL_FWD, k
