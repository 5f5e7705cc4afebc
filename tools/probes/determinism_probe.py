"""Which launch of the EAGER step is not reproducible?  (follow-up of GPUTEST_r04)

tools/probes/graph_case_losses.py settled which side of the round-4 failure was wrong: 752.95263671875 -- the value the
GRAPH replay gave on the driver's box -- is what both runners give on a box where the test passes, so the EAGER
train_step of the second iteration (750.267578125) was the one that erred.  That iteration is the first whose forward
packs its weight images through the batched launch (ops.PackPlan builds its device job table then).  This probe re-runs
that eager step from identical state and compares EVERY tensor the step produces -- all activations kept for backward,
BatchNorm coefficients, outputs, loss, every gradient -- bit for bit against the first run, in three regimes: back to
back, from an idle device, and with a fresh PackPlan each time (table upload + first batched pack every run).  Prints
the first tensor (in forward order) that differs, per run.  Bounded: `runs` (default 40) runs per regime.
"""
import copy
import gc
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from unet_nested4tiny_objects_keypoints_amd import FocalLoss_BCE_2d, UNet_Nested, ops  # noqa: E402


def tensors_of(model, outs, loss):
    s = model._debug_saved
    out = []
    d = model.depth
    order = [(i, 0) for i in range(d)] + [(i, j) for j in range(1, d) for i in range(d - j)]   # forward order (engine.py)
    for key in order:
        if key in s.ups:
            u = s.ups[key]
            for name in ("interp", "up"):
                t = getattr(u, name)
                if t is not None:
                    out.append(("up%d%d.%s" % (key[0], key[1], name), t))
        r = s.pairs[key]
        for name in ("y1", "a1", "y2", "out", "pooled", "pool_idx"):
            t = getattr(r, name)
            if t is not None:
                out.append(("X%d%d.%s" % (key[0], key[1], name), t))
        for bn_name in ("bn1", "bn2"):
            bn = getattr(r, bn_name)
            if bn is not None:
                for nm, t in zip(("mean", "invstd", "scale", "shift"), bn):
                    if t is not None:
                        out.append(("X%d%d.%s.%s" % (key[0], key[1], bn_name, nm), t))
    for i, o in enumerate(outs):
        out.append(("out%d" % i, o.detach()))
    out.append(("loss", loss.detach()))
    return out


def one_step(model, crit, x, t):
    for p in model.parameters():
        p.grad = None
    outs = model(x)
    loss = sum(crit(o, t) for o in outs) / len(outs)
    fwd = [(k, v.clone()) for k, v in tensors_of(model, outs, loss)]
    loss.backward()
    grads = [("grad/" + k, p.grad.clone()) for k, p in model.named_parameters()]
    bufs = [("buf/" + k, b.clone()) for k, b in model.named_buffers()]
    torch.cuda.synchronize()
    return fwd + grads + bufs


def bits(t):
    return t.view(torch.int16) if t.dtype == torch.bfloat16 else t


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    only = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device("cuda:0")
    cases = [
        ("bf16-bilinear-fs2-64 (GPUTEST_r04)", dict(in_channels=1, n_classes=4, feature_scale=2, is_deconv=False), True, 2, 64),
        ("bf16-deconv-fs2-64", dict(in_channels=1, n_classes=4, feature_scale=2), True, 2, 64),
        ("f32-fs4-64", dict(in_channels=1, n_classes=4, feature_scale=4), False, 2, 64),
        ("bf16-d5-fs4-64", dict(in_channels=3, n_classes=5, feature_scale=4, depth=5), True, 2, 64),
        # wider sweep (round 5, after the fix): the other structures and larger geometries, both storages
        ("wide-f32-base32-128", dict(in_channels=1, n_classes=4, feature_scale=1), False, 4, 128),
        ("wide-bf16-base32-128", dict(in_channels=1, n_classes=4, feature_scale=1), True, 4, 128),
        ("wide-bf16-d5-base64-96", dict(in_channels=3, n_classes=5, feature_scale=0.5, depth=5), True, 2, 96),
        ("wide-f32-bilinear-nobn-fs2", dict(in_channels=3, n_classes=5, feature_scale=2, is_deconv=False, is_batchnorm=False), False, 2, 64),
        ("wide-bf16-nobn-fs2", dict(in_channels=1, n_classes=4, feature_scale=2, is_batchnorm=False), True, 2, 96),
        ("wide-f32-fs2-160", dict(in_channels=3, n_classes=4, feature_scale=2), False, 2, 160),
        ("wide-bf16-bilinear-base32-96", dict(in_channels=1, n_classes=4, feature_scale=1, is_deconv=False), True, 2, 96),
        # the BENCHMARKED geometries themselves (BASELINE configs[1], [3], [4]): a few runs each is all they take
        ("bench-f32-configs1-256-b32", dict(in_channels=1, n_classes=4, feature_scale=1), False, 32, 256),
        ("bench-bf16-configs3-512-b8", dict(in_channels=1, n_classes=4, feature_scale=1), True, 8, 512),
        ("bench-bf16-configs4-d5-384-b4", dict(in_channels=3, n_classes=5, feature_scale=0.5, depth=5), True, 4, 384),
    ]
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    for name, ctor, bf16, b, size in cases:
        if only is not None and only not in name:
            continue
        torch.manual_seed(81)
        base = UNet_Nested(**ctor).to(dev).train()
        if bf16:
            base.set_activation_dtype(torch.bfloat16)
        base.drop_out.p = 0.0
        g = torch.Generator().manual_seed(5)
        x = torch.randn(b, ctor["in_channels"], size, size, generator=g).to(dev)
        t = torch.rand(b, ctor["n_classes"], size, size, generator=g).to(dev)
        for regime in ("back-to-back", "idle-start", "fresh-plan", "fresh-plan-second-pass"):
            m = copy.deepcopy(base)
            m._debug_keep_saved = True
            ref = None
            bad_runs, first_bad = 0, {}
            for r in range(runs):
                if regime.startswith("fresh-plan"):
                    m = None
                    gc.collect()                     # (a model that kept its saved activations is cyclic garbage: 7-14 GB per run
                    torch.cuda.empty_cache()         # at the benchmarked geometries, which nothing collects under GPU memory pressure)
                    m = copy.deepcopy(base)          # a deep copy starts with an empty PackPlan
                    m._debug_keep_saved = True
                    if regime.endswith("second-pass"):
                        one_step(m, crit, x, t)      # pass 1 records the jobs; the measured pass builds the table
                        with torch.no_grad():        # (BatchNorm buffers moved: put them back)
                            for (k, bsrc), (_, bdst) in zip(base.named_buffers(), m.named_buffers()):
                                bdst.copy_(bsrc)
                if regime == "idle-start":
                    torch.cuda.synchronize()
                    time.sleep(0.02)
                got = one_step(m, crit, x, t)
                if regime != "fresh-plan":
                    with torch.no_grad():
                        for (k, bsrc), (_, bdst) in zip(base.named_buffers(), m.named_buffers()):
                            bdst.copy_(bsrc)
                if ref is None:
                    ref = got
                    continue
                diff = [k for (k, a), (_, c) in zip(ref, got) if not k.startswith("buf/") and not torch.equal(bits(a), bits(c))]
                if diff:
                    bad_runs += 1
                    first_bad[diff[0]] = first_bad.get(diff[0], 0) + 1
                    fwd = [k for k in diff if not k.startswith("grad/")]
                    k0 = diff[0]
                    a0 = dict(ref)[k0].float().flatten()
                    c0 = dict(got)[k0].float().flatten()
                    idx = (a0 != c0).nonzero().flatten()
                    shape = tuple(dict(ref)[k0].shape)
                    print("    run %d: %d tensors differ; forward-order list: %s ...; in %s %s: %d elements differ, first flat indices %s, "
                          "max |diff| %.4g" % (r, len(diff), fwd[:8], k0, shape, idx.numel(), idx[:6].tolist(),
                                                float((a0 - c0).abs().max())), flush=True)
            print("%-36s %-24s runs %d  differing runs %d  first differing tensor: %s" % (name, regime, runs, bad_runs, first_bad or "-"),
                  flush=True)


if __name__ == "__main__":
    main()
