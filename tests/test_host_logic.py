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


def test_graphed_train_step_refuses_cpu_tensors_and_eval_models():
    """graph.GraphedTrainStep has no CPU form (this path has no CPU fallback): CPU tensors and eval-mode models are
    refused up front, before anything touches the library."""
    import torch

    from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, GraphedTrainStep, UNet_Nested
    m = UNet_Nested(in_channels=1, n_classes=4, feature_scale=8).train()
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    x, t = torch.randn(1, 1, 32, 32), torch.rand(1, 4, 32, 32)
    with pytest.raises(RuntimeError, match="GPU"):
        GraphedTrainStep(m, opt, crit, x, t)


def test_bench_other_configs_name_the_baseline_configurations():
    """bench.py's default run attaches BASELINE.json's configs[3] and configs[4] (as single-GPU geometries) to the headline
    line; the table that says which is host logic."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    names = [n for n, _ in bench.OTHER_CONFIGS]
    assert names[0].startswith("configs[3]") and names[1].startswith("configs[4]")
    c3, c4 = bench.OTHER_CONFIGS[0][1], bench.OTHER_CONFIGS[1][1]
    assert (c3["dtype"], c3["size"], c3["depth"], c3["in_channels"], c3["n_classes"]) == ("bf16", 512, 4, 1, 4)
    assert (c4["dtype"], c4["size"], c4["depth"], c4["in_channels"], c4["n_classes"], c4["feature_scale"]) == ("bf16", 384, 5, 3, 5, 0.5)
    base = json.load(open(os.path.join(root, "BASELINE.json")))["configs"]
    assert "512" in base[3] and "bf16" in base[3] and "384" in base[4] and "depth=5" in base[4]
    import argparse
    ns = argparse.Namespace(dtype="f32", size=256, batch=32, feature_scale=1, depth=4, in_channels=1, n_classes=4)
    assert bench.is_headline(ns)
    ns.batch = 8
    assert not bench.is_headline(ns)     # other shapes asked for on the command line stay single-configuration runs


def test_bench_live_pmc_falls_back_without_a_gpu(tmp_path):
    """bench.live_pmc() starts children of bench.py under `rocprofv3 --pmc` before the benchmark touches the GPU.  Whatever goes
    wrong there (here: no GPU, so the first child exits non-zero; elsewhere: no rocprofv3, a timeout) must cost only the
    live counters -- the function returns None, leaves no scratch directory behind, and pmc_counters() then serves the
    committed summary if and only if it was made from this build."""
    import argparse
    import glob
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module_pmc", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ns = argparse.Namespace(dtype="f32", size=256, batch=32, feature_scale=1, depth=4, in_channels=1, n_classes=4)
    before = set(glob.glob("/tmp/unetpp_pmc_*"))
    os.environ["UNETPP_BENCH_PMC_TIMEOUT"] = "120"
    try:
        assert bench.live_pmc(ns) is None
    finally:
        os.environ.pop("UNETPP_BENCH_PMC_TIMEOUT", None)
    assert set(glob.glob("/tmp/unetpp_pmc_*")) == before
    key = bench.config_key(ns)
    assert key == "f32_d4_fs1_s256_b32_i1_c4"
    assert bench.pmc_counters("gemm_wino_kernel", "not-a-build-hash", key) == (None, None, None)
    bench._LIVE_PMC[key] = {"build_hash": "h", "kernels": [
        {"kernel": "gemm_wino_kernel<5, 2, 1, false>", "launches": 3, "fetch_bytes_x2_per_launch": 100, "write_bytes_per_launch": 20,
         "sq": {"mfma_busy": 0.5, "avg_us": 10.0, "dispatches": 3}},
        {"kernel": "gemm_wino_kernel<5, 2, 2, true>", "launches": 1, "fetch_bytes_x2_per_launch": 60, "write_bytes_per_launch": 20,
         "sq": {"mfma_busy": 0.25, "avg_us": 10.0, "dispatches": 1}}]}
    traffic, busy, source = bench.pmc_counters("gemm_wino_kernel", "h", key)
    assert traffic == round((120 * 3 + 80) / 4) and abs(busy - (0.5 * 30 + 0.25 * 10) / 40) < 1e-4 and source.startswith("live")
    assert bench.pmc_counters("gemm_wino_kernel", "other", key) == (None, None, None)   # a live summary of another build: refused too
