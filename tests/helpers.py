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


# depth != 4, also produced by the reference: level 5 from its commented-out lines switched on in a
# fixture-time subclass, depth 2/3 as truncations of its graph (tests/golden/make_golden.py)
DEPTH_CASES = [
    "d5_fs8_rgb5_32x32_b2",
    "d5_fs8_bilinear_32x48_b2",
    "d3_fs8_32x48_b2",
    "d2_fs4_64x64_b4",
]


# the reference's classic U-Net: a small-width instance built from the reference's own blocks with
# its full state, and the reference's UNet() itself (13.4 M parameters: the state comes from `seeded_state`, the
# fixture keeps the seed, outputs, loss, BatchNorm buffers and subsampled gradients)
PLAIN_CASES = ["unet_w8_rgb5_32x48_b2", "unet_ref_rgb5_64x64_b1", "unet_w8_rgb5_40x56_b2"]
GRAD_STRIDE = 1009  # large gradients are stored as flat[::GRAD_STRIDE]


def seeded_state(module, seed):
    """A deterministic, well-scaled state_dict for `module` (same values for any module with the same keys/shapes):
    conv weights N(0, 2/fan_in), biases and BatchNorm offsets N(0, 0.1^2), gamma 1 + N(0, 0.1^2), running_var in [1, 1.1)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in module.state_dict().items():
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            out[k] = 1 + 0.1 * torch.rand(v.shape, generator=g)
        elif v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            out[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith(".weight"):  # BatchNorm gamma
            out[k] = 1 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
    return out


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
    """got / want: dict name -> tensor.  Reports every parameter out of tolerance, not just the first."""
    bad = []
    for k, w in want.items():
        g = got[k]
        assert tuple(g.shape) == tuple(w.shape), k
        if is_pre_bn_bias(k, ctor):
            scale = float(want[k.replace(".bias", ".weight")].abs().max())
            if not float(g.abs().max()) <= 1e-3 * scale + 1e-30:
                bad.append((k, "pre-BN bias", float(g.abs().max()), scale))
        elif not rel_err(g, w) < tol:
            bad.append((k, rel_err(g, w)))
    assert not bad, bad


# ---------------------------------------------------------------------------------------------
# ReLU gates.  A ReLU network's gradient is discontinuous where a pre-activation crosses zero.
# Two fp32 implementations agree on activations to ~1e-6, so among ~1e6 gated elements a handful
# sit within rounding of zero and come out "0" on one side and "+4e-7" on the other; each such
# flip changes the gradients of everything upstream by far more than 1e-4 although both sides
# are right.  The gradient parity tests therefore run the oracle's BACKWARD with the gate pattern
# of the HIP forward (values of the forward are untouched), which makes the comparison exact
# again, and separately count how many gates differ from the oracle's own.
class _GatedReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate):
        ctx.save_for_backward(gate)
        return x.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (gate,) = ctx.saved_tensors
        return g * gate.to(g.dtype), None


class GatedReLU(torch.nn.Module):
    def __init__(self, gate):
        super().__init__()
        self.gate = gate
        self.flips = 0
        self.total = int(gate.numel())
        self.flip_mag = 0.0   # largest |oracle pre-activation| at a flipped gate, relative to the tensor's max

    def forward(self, x):
        y = _GatedReLUFn.apply(x, self.gate)
        flipped = (y > 0) != self.gate
        self.flips = int(flipped.sum())
        if self.flips:
            self.flip_mag = float(x.detach()[flipped].abs().max() / x.detach().abs().max())
        return y


FLIP_FRACTION_MAX = 2e-5   # of all gated activations
FLIP_MAGNITUDE_MAX = 2e-5  # a gate may only differ where the oracle's pre-activation is rounding noise around zero


def check_flips(gated, label=""):
    """Bound and report the ReLU gates / max-pool winners on which the HIP forward and the oracle differ: few, and
    only where the oracle's own pre-activation is within rounding of zero (the two window candidates within rounding
    of each other) -- a kernel that mis-gates or mis-routes real activations fails here."""
    import json
    flips, total = sum(g.flips for g in gated), sum(g.total for g in gated)
    mag = max([g.flip_mag for g in gated] + [0.0])
    line = {"case": label, "flips": flips, "gates": total, "fraction": flips / max(1, total), "max_rel_preactivation": mag}
    print("relu-gate flips:", json.dumps(line))
    out = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "relu_gate_flips.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
    assert flips <= max(2, FLIP_FRACTION_MAX * total), line
    assert mag <= FLIP_MAGNITUDE_MAX, line
    return flips


# Max-pool routing is the second discontinuity: when the two largest values of a 2x2 window agree to rounding, the
# two implementations may pick different winners and send the window's gradient to different pixels (a few windows in
# 4e6 at BASELINE configs[1] geometry).  Same treatment: the oracle's pool keeps its own forward VALUES but routes the
# gradient to the HIP forward's argmax; the windows where the winners differ are counted, and the two candidates'
# values must agree to rounding there (check_flips).
class _RoutedPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat_idx):
        ctx.save_for_backward(flat_idx)
        ctx.shape = x.shape
        return torch.nn.functional.max_pool2d(x, 2)

    @staticmethod
    def backward(ctx, g):
        (flat_idx,) = ctx.saved_tensors
        b, c, h, w = ctx.shape
        out = torch.zeros(b, c, h * w, dtype=g.dtype)
        out.scatter_(2, flat_idx.view(b, c, -1), g.reshape(b, c, -1))
        return out.view(b, c, h, w), None


class RoutedMaxPool(torch.nn.Module):
    """Stands in for the oracle's shared nn.MaxPool2d(2): call k uses the HIP forward's argmax of encoder level k."""

    def __init__(self, hip_idx_per_level):
        super().__init__()
        self.hip_idx = hip_idx_per_level  # uint8 [b, h/2, w/2, c], value = 2*iy + ix inside the window
        self.calls = 0
        self.flips = 0
        self.total = 0
        self.flip_mag = 0.0

    def forward(self, x):
        idx = self.hip_idx[self.calls].permute(0, 3, 1, 2).long()
        self.calls += 1
        b, c, h, w = x.shape
        oy = 2 * torch.arange(h // 2).view(1, 1, -1, 1) + idx // 2
        ox = 2 * torch.arange(w // 2).view(1, 1, 1, -1) + idx % 2
        flat = (oy * w + ox).contiguous()
        y = _RoutedPoolFn.apply(x, flat)
        with torch.no_grad():
            at_hip = x.detach().reshape(b, c, -1).gather(2, flat.view(b, c, -1)).view_as(y)
            differ = at_hip != y.detach()  # the HIP winner is not a maximum of the oracle's window
            self.flips += int(differ.sum())
            self.total += y.numel()
            if bool(differ.any()):
                self.flip_mag = max(self.flip_mag, float((y.detach() - at_hip)[differ].abs().max() / x.detach().abs().max()))
        return y


def install_hip_gates(oracle_model, hip_saved):
    """Swap every ReLU of the oracle for one whose backward uses the HIP forward's gate pattern, and its max-pool for
    one that routes gradients to the HIP forward's argmax.  Returns the list of stand-in modules (read .flips after
    the oracle forward)."""
    gated = []

    def nchw_mask(t):
        return (t.permute(0, 3, 1, 2) > 0).cpu()

    d = oracle_model.depth
    for (i, j), rec in hip_saved.pairs.items():
        pair = getattr(oracle_model, "conv%d0" % i) if j == 0 else getattr(oracle_model, "up_concat%d%d" % (i, j)).conv
        a1 = rec.a1
        if a1 is None:  # BatchNorm pairs fold BN1-apply + ReLU into the consumer's load: rebuild the gate from y1
            scale, shift = rec.bn1[2].double(), rec.bn1[3].double()
            a1 = (rec.y1.double() * scale + shift).float()  # only the sign is used
        for seq, act in ((pair.conv1, a1), (pair.conv2, rec.out)):
            mod = GatedReLU(nchw_mask(act))
            seq[len(seq) - 1] = mod
            gated.append(mod)
    assert len(gated) == 2 * (d + d * (d - 1) // 2)
    pool = RoutedMaxPool([hip_saved.pairs[(i, 0)].pool_idx.cpu() for i in range(d - 1)])
    oracle_model.maxpool = pool
    gated.append(pool)
    return gated


def install_hip_gates_plain(oracle_model, hip_saved):
    """The same stand-ins for the classic U-Net oracle (oracle/unet_plain_oracle.py): the ReLUs sit at indices 2 and 5
    of every double_conv's Sequential, every `down` owns its MaxPool2d (mpconv[0])."""
    gated = []

    def nchw_mask(t):
        return (t.permute(0, 3, 1, 2) > 0).cpu()

    enc = [oracle_model.inc.conv, oracle_model.down1.mpconv[1], oracle_model.down2.mpconv[1],
           oracle_model.down3.mpconv[1], oracle_model.down4.mpconv[1]]
    dec = [oracle_model.up1.conv, oracle_model.up2.conv, oracle_model.up3.conv, oracle_model.up4.conv]
    for dc, rec in list(zip(enc, hip_saved.enc)) + list(zip(dec, hip_saved.dec)):
        scale, shift = rec.bn1[2].double(), rec.bn1[3].double()
        a1 = (rec.y1.double() * scale + shift).float()  # BN1-apply + ReLU is folded into conv2's load: only the sign
        for idx, act in ((2, a1), (5, rec.out)):
            mod = GatedReLU(nchw_mask(act))
            dc.conv[idx] = mod
            gated.append(mod)
    for i, d in enumerate((oracle_model.down1, oracle_model.down2, oracle_model.down3, oracle_model.down4)):
        pool = RoutedMaxPool([hip_saved.enc[i].pool_idx.cpu()])
        d.mpconv[0] = pool
        gated.append(pool)
    return gated
