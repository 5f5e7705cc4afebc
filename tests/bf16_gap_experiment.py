"""CPU experiment (not a test; not collected): what moves the bf16 path's parameter gradients away from the fp32 oracle's?

    python tests/bf16_gap_experiment.py [size] [batch] [base] [depth]

Re-runs the forward of oracle/bf16_sim.py (no HIP code; same rounding points) with the roundings switched on one class
at a time and prints, per variant, the worst and the median relative L2 distance of a parameter gradient from the
float64 oracle's (pre-BatchNorm conv biases excluded: their analytic gradient is zero), plus the worst parameters.
The last variants separate VALUES from ROUTING: float64 values with the bf16 forward's ReLU gates / pool winners, and
bf16 values with the float64 forward's.  Results: DESIGN.md section 2 (round 4).
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import bf16_sim as S
from oracle.step_oracle import focal_bce_2d_oracle
from oracle.unet_nested_oracle import UNetNestedOracle


def forward(model, x, R, routing=None, record=None):
    """R: set of rounding classes in force: 'w' weights, 'ency' encoder pre-BN conv outputs, 'enca' encoder activations,
    'dec' decoder conv outputs, 'up' transposed-conv outputs.  routing: {(i, j): (gate1, gate2)}, {i: pool winners}
    used in backward; record: dict that receives this forward's own."""
    d = model.depth
    gates_in, pools_in = routing if routing is not None else ({}, {})
    rb = S.rb
    stats = dict(gates=0, gate_flips=0, windows=0, pool_flips=0)

    def relu(v, key, k):
        g = gates_in.get(key)
        out = S._relu(v, None if g is None else g[k], stats)
        if record is not None:
            record.setdefault("gates", {}).setdefault(key, [None, None])[k] = (out.detach() > 0)
        return out

    def pair(blk, xx, with_bn, key):
        for k, name in enumerate(("conv1", "conv2")):
            seq = getattr(blk, name)
            conv = seq[0]
            w = conv.weight
            if conv.in_channels > 4 and "w" in R:
                w = rb(w)
            y = F.conv2d(xx, w, conv.bias, padding=1)
            if with_bn:
                if "ency" in R:
                    y = rb(y)
                mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
                scale = seq[1].weight / torch.sqrt(var + seq[1].eps)
                shift = seq[1].bias - mean * scale
                xx = relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), key, k)
                if "enca" in R:
                    xx = rb(xx)
            else:
                xx = relu(y, key, k)
                if "dec" in R:
                    xx = rb(xx)
        return xx

    def pool(v, i):
        idx = pools_in.get(i)
        if record is not None:
            b, c, h, w = v.shape
            win = v.detach().view(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, c, h // 2, w // 2, 4)
            record.setdefault("pools", {})[i] = win.argmax(-1).permute(0, 2, 3, 1).to(torch.uint8).contiguous()
        return S._pool(v, idx, stats)

    X = [[None] * d for _ in range(d)]
    X[0][0] = pair(model.conv00, x, True, (0, 0))
    for i in range(1, d):
        X[i][0] = pair(getattr(model, "conv%d0" % i), pool(X[i - 1][0], i - 1), True, (i, 0))
    for j in range(1, d):
        for i in range(d - j):
            up = getattr(model, "up_concat%d%d" % (i, j))
            w = rb(up.up.weight) if "w" in R else up.up.weight
            u = F.conv_transpose2d(X[i + 1][j - 1], w, up.up.bias, stride=2)
            if "up" in R:
                u = rb(u)
            X[i][j] = pair(up.conv, torch.cat([u] + X[i][:j], 1), False, (i, j))
    outs = []
    for j in range(1, d):
        head = getattr(model, "final_%d" % j)
        outs.append(torch.sigmoid(F.conv2d(X[0][j], head.weight, head.bias)))
    return tuple(outs), stats


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    base = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    depth = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = UNetNestedOracle(in_channels=1, n_classes=4, feature_scale=32 / base, depth=depth).double().train()
    model.drop_out.eval()
    x = torch.randn(batch, 1, size, size, dtype=torch.float64)
    target = torch.rand(batch, 4, size, size, dtype=torch.float64)

    def grads(R, routing=None, record=None):
        model.zero_grad()
        outs, stats = forward(model, x, R, routing, record)
        loss = sum(focal_bce_2d_oracle(o, target) for o in outs) / len(outs)
        loss.backward()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}, float(loss.detach()), stats

    rec64, recbf = {}, {}
    ref, l0, _ = grads(set(), record=rec64)
    ALL = {"w", "ency", "enca", "dec", "up"}
    route = lambda rec: ({k: tuple(v) for k, v in rec["gates"].items()}, rec["pools"])   # noqa: E731
    _, _, _ = grads(ALL, record=recbf)
    variants = [("shipped: every stored tensor + weights bf16", ALL, None),
                ("encoder pre-BN outputs fp32 (VERDICT item 2)", ALL - {"ency"}, None),
                ("whole encoder fp32 (y and activations)", ALL - {"ency", "enca"}, None),
                ("only weights bf16", {"w"}, None),
                ("only encoder pre-BN outputs bf16", {"ency"}, None),
                ("only encoder activations bf16", {"enca"}, None),
                ("only decoder outputs bf16", {"dec"}, None),
                ("only transposed-conv outputs bf16", {"up"}, None),
                ("fp64 values, ROUTING of the bf16 forward", set(), route(recbf)),
                ("bf16 values, ROUTING of the fp64 forward", ALL, route(rec64))]
    for name, R, routing in variants:
        g, l, stats = grads(R, routing)
        rel = {k: float((g[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-300)) for k in ref
               if not re.fullmatch(r"conv\d0\.conv[12]\.0\.bias", k)}
        vals = sorted(rel.values())
        worst = sorted(rel.items(), key=lambda kv: -kv[1])[:3]
        flips = "" if routing is None else "  gate flips %d / %d" % (stats["gate_flips"], stats["gates"])
        print("%-48s worst %.3f  median %.4f   %s%s"
              % (name, vals[-1], vals[len(vals) // 2], ", ".join("%s %.3f" % kv for kv in worst), flips), flush=True)


if __name__ == "__main__":
    main()
