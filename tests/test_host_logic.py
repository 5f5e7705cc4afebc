"""CPU-only checks of the host side that mirrors the reference's module API (models/unet.py:204-300):
constructor defaults, attributes, state-dict contract, initialisation statistics, loss and step helpers."""
import math

import pytest
import torch

from oracle.step_oracle import focal_bce_2d_oracle
from oracle.unet_nested_oracle import UNetNestedOracle
from tests.helpers import GOLDEN_CASES, load_golden, sub
from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, count_param


def test_zero_arg_constructor_matches_reference_defaults():
    m = UNet_Nested()  # trainer/trainer.py:337 builds it with no arguments
    assert (m.in_channels, m.feature_scale, m.is_deconv, m.is_batchnorm, m.is_ds) == (3, 2, True, True, True)
    assert count_param(m) == 553260          # the "2.2MB" weights file of README.md:12
    assert count_param(UNet_Nested(1, 4, feature_scale=1)) == 2207244
    assert count_param(UNet_Nested(1, 4, feature_scale=4)) == 138828
    assert isinstance(m.drop_out, torch.nn.Dropout) and m.drop_out.p == 0.4


@pytest.mark.parametrize("kw", [dict(), dict(is_deconv=False), dict(is_batchnorm=False), dict(in_channels=1, n_classes=5),
                                dict(depth=5, feature_scale=4), dict(depth=2), dict(feature_scale=0.5, depth=3)])
def test_state_dict_contract_equals_oracle(kw):
    a, b = UNet_Nested(**kw).state_dict(), UNetNestedOracle(**kw).state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_reference_checkpoints_load_strictly(name):
    z, ctor = load_golden(name)
    m = UNet_Nested(**ctor)
    missing, unexpected = m.load_state_dict(sub(z, "state0"), strict=True)
    assert not missing and not unexpected
    # and back: the oracle (== reference layout) accepts what the HIP module saves
    UNetNestedOracle(**ctor).load_state_dict(m.state_dict(), strict=True)


def test_initialisation_statistics():
    torch.manual_seed(0)
    m = UNet_Nested(in_channels=1, n_classes=4, feature_scale=1)
    w = m.up_concat03.conv.conv1[0].weight if hasattr(m.up_concat03.conv.conv1, "__getitem__") else \
        getattr(m.up_concat03.conv.conv1, "0").weight
    fan_in = w.shape[1] * 9
    assert abs(float(w.std()) - math.sqrt(2.0 / fan_in)) < 0.05 * math.sqrt(2.0 / fan_in)   # kaiming normal, fan_in
    bn = getattr(m.conv00.conv1, "1")
    assert abs(float(bn.weight.mean()) - 1.0) < 0.02 and float(bn.bias.abs().max()) == 0.0
    assert float(bn.running_var.min()) == 1.0 and int(bn.num_batches_tracked) == 0
    up = m.up_concat01.up.weight
    assert abs(float(up.std()) - math.sqrt(2.0 / (up.shape[1] * 4))) < 0.05


def test_loss_matches_oracle_and_trainer_semantics():
    torch.manual_seed(1)
    p, t = torch.rand(3, 4, 8, 8), torch.rand(3, 4, 8, 8)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    assert torch.allclose(crit(p, t), focal_bce_2d_oracle(p, t), rtol=1e-6)
    # sum over everything / (N*C): doubling the batch with the same content keeps the value
    assert torch.allclose(crit(torch.cat([p, p]), torch.cat([t, t])), crit(p, t), rtol=1e-6)


def test_invalid_arguments():
    with pytest.raises(ValueError):
        UNet_Nested(depth=1)
    with pytest.raises(ValueError):
        UNet_Nested(feature_scale=64)
    m = UNet_Nested(in_channels=1, feature_scale=8)
    m.dropout_masks = [None]
    with pytest.raises(RuntimeError):   # CPU tensor: no fallback
        m(torch.randn(1, 1, 16, 16))


def test_bench_spawns_its_ranks_and_propagates_failure():
    """`python bench.py --gpus 2` without a launcher starts two rank processes itself (it used to run one GPU
    silently).  Without a GPU every rank refuses to run, and the parent must report that with a non-zero exit code --
    which exercises the spawn path, the environment it hands to the ranks and the failure propagation on CPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check (on a GPU box the ranks would run the benchmark)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr and "bench.py needs a GPU" in r.stderr
    assert r.stderr.count("bench.py needs a GPU") == 2   # both ranks were started
