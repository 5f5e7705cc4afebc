"""Shared helpers for the parity tests (test infrastructure)."""
import ast
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

GOLDEN_CASES = [
    "c1_fs4_64x64_b4_seed0",
    "c1_fs4_64x64_b2_seed1",
    "fs8_bilinear_32x32_b2",
    "fs8_nobn_32x32_b2",
    "fs8_rgb5_24x40_b2",
]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    ctor = dict(ast.literal_eval(str(z["meta/ctor"])))
    return z, ctor


def sub(z, prefix):
    """All entries under 'prefix/' as torch tensors keyed by the remainder."""
    plen = len(prefix) + 1
    return {k[plen:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(prefix + "/")}


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'relative fp32' measure used for the 1e-4 bar."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    denom = float(b.abs().max())
    if denom == 0.0:
        return float((a - b).abs().max())
    return float((a - b).abs().max()) / denom


def is_pre_bn_bias(key, ctor):
    """Conv biases that feed a BatchNorm have an analytically zero gradient (the batch
    mean removes them); both sides hold rounding noise there, so they are compared
    against the size of the sibling weight gradient instead of against each other."""
    if not ctor.get("is_batchnorm", True):
        return False
    return key.endswith(".0.bias") and key.startswith("conv") and ".conv" in key


def assert_grads_close(got, want, ctor, tol):
    """got / want: dict name -> tensor."""
    for k, w in want.items():
        g = got[k]
        assert tuple(g.shape) == tuple(w.shape), k
        if is_pre_bn_bias(k, ctor):
            scale = float(want[k.replace(".bias", ".weight")].abs().max())
            assert float(g.abs().max()) <= 1e-3 * scale + 1e-30, (k, float(g.abs().max()), scale)
        else:
            assert rel_err(g, w) < tol, (k, rel_err(g, w))
