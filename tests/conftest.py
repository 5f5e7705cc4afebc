import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# Order of the GPU suite (the driver runs `pytest -x`): parity against the oracle / the reference's golden fixtures comes
# first -- per kernel, then whole network, then the benchmarked block, the caller, the bf16 twins, the classic UNet, key
# points, the >2 GB operands -- then the data-parallel path, and LAST the tests that only compare the library with
# itself (repeatability of every kernel, workspace poisoning, HIP-graph replay against the eager step): a failure there must never hide a parity row
# of SURVEY.md section 8 (reference path models/unet.py:121-300).
_GPU_ORDER = [
    "test_gpu_kernels", "test_gpu_model", "test_gpu_x00_block", "test_gpu_caller", "test_gpu_bf16",
    "test_gpu_unet_plain", "test_keypoints", "test_gpu_large", "test_gpu_dp", "test_gpu_repeat", "test_gpu_workspaces",
    "test_gpu_graph",
]
_LAST_KEYWORDS = ("graphed", "GraphedForward", "graph_replay")   # self-comparison cases inside parity files


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _rank(item):
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    try:
        r = _GPU_ORDER.index(mod)
    except ValueError:
        r = _GPU_ORDER.index("test_gpu_dp") - 0.5 if item.get_closest_marker("gpu") else -1   # CPU tests keep their place
    if item.get_closest_marker("gpu") and any(k in item.name for k in _LAST_KEYWORDS) and mod != "test_gpu_graph":
        r = len(_GPU_ORDER) - 1.5
    return r


def pytest_collection_modifyitems(config, items):
    items.sort(key=_rank)   # stable: file order inside a rank is kept


def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
