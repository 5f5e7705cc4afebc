"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: imports /root/reference/models/unet.py,
tools/losses/focal_loss.py and tools/misc/helper.py as they are (CPU torch) and
records inputs + outputs.  The fixtures are data (state_dict, inputs, outputs,
loss, gradients, BN running stats, parameters after one optimizer step); no
reference source is stored.  Re-run with:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

The reference does not exist on the GPU box; nothing at test time calls this.
"""
import os
import sys
import warnings

import numpy as np

REF = os.environ.get("UNETPP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _flat(prefix, tensors):
    return {"%s/%s" % (prefix, k): v.detach().cpu().numpy().copy() for k, v in tensors.items()}


def run_case(name, ctor_kwargs, batch, hw, seed, with_opt_steps=False, keypoints=False):
    import torch
    import numpy.matlib  # noqa: F401  (tools/misc/helper.py uses np.matlib without importing it)
    from models.unet import UNet_Nested
    from tools.losses.focal_loss import FocalLoss_BCE_2d
    from tools.misc.helper import create_heatmap

    torch.manual_seed(seed)
    torch.set_num_threads(1)  # fixed reduction order for reproducible fixtures
    model = UNet_Nested(**ctor_kwargs)
    n_cls = model.final_1.out_channels
    cin = ctor_kwargs.get("in_channels", 3)
    h, w = hw
    x = torch.randn(batch, cin, h, w)
    target = torch.rand(batch, n_cls, h, w)
    blob = {}
    blob.update(_flat("state0", model.state_dict()))
    blob["x"] = x.numpy()
    blob["target"] = target.numpy()
    blob["meta/ctor"] = np.array(repr(sorted(ctor_kwargs.items())))

    # eval-mode forward (running stats, dropout off)
    model.eval()
    with torch.no_grad():
        outs = model(x)
    for i, o in enumerate(outs):
        blob["eval_out/%d" % i] = o.numpy()

    # train-mode step with only dropout disabled
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    model.train()
    model.drop_out.eval()
    model.zero_grad()
    outs = model(x)
    loss = 0
    for o in outs:
        loss = loss + crit(o, target)
    loss = 1.0 * loss / len(outs)
    loss.backward()
    for i, o in enumerate(outs):
        blob["train_out/%d" % i] = o.detach().numpy()
    blob["loss"] = loss.detach().numpy()
    blob.update(_flat("grad", {k: p.grad for k, p in model.named_parameters()}))
    blob.update(_flat("state1_buffers", dict(model.named_buffers())))

    if keypoints:
        pts = torch.rand(batch, 7, 2) * torch.tensor([float(w), float(h)])
        blob["kp/points"] = pts.numpy()
        blob["kp/heatmap"] = create_heatmap(pts, h, w)

    if with_opt_steps:
        for tag, make in (("adam", lambda p: torch.optim.Adam(p, lr=1e-3)),
                          ("sgd", lambda p: torch.optim.SGD(p, lr=0.01, momentum=0.9))):
            torch.manual_seed(seed)
            m2 = UNet_Nested(**ctor_kwargs)
            m2.load_state_dict({k[len("state0/"):]: torch.from_numpy(v) for k, v in blob.items()
                                if k.startswith("state0/")})
            m2.train()
            m2.drop_out.eval()
            opt = make(m2.parameters())
            opt.zero_grad()
            outs = m2(x)
            l2 = 0
            for o in outs:
                l2 = l2 + crit(o, target)
            l2 = 1.0 * l2 / len(outs)
            l2.backward()
            opt.step()
            blob.update(_flat("after_%s" % tag, dict(m2.named_parameters())))

    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **blob)
    print("%-28s %8.1f KB  loss=%.6f" % (name, os.path.getsize(path) / 1024, float(loss)))


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    warnings.filterwarnings("ignore")
    # configs[0]-sized: the reference-instantiable 4-level net at base width 8
    run_case("c1_fs4_64x64_b4_seed0", dict(in_channels=1, n_classes=4, feature_scale=4),
             4, (64, 64), 0, with_opt_steps=True, keypoints=True)
    run_case("c1_fs4_64x64_b2_seed1", dict(in_channels=1, n_classes=4, feature_scale=4),
             2, (64, 64), 1)
    # small variants: bilinear up path, no batch-norm, 3->5 channels, non-square
    run_case("fs8_bilinear_32x32_b2", dict(in_channels=1, n_classes=4, feature_scale=8, is_deconv=False),
             2, (32, 32), 0)
    run_case("fs8_nobn_32x32_b2", dict(in_channels=1, n_classes=4, feature_scale=8, is_batchnorm=False),
             2, (32, 32), 0)
    run_case("fs8_rgb5_24x40_b2", dict(in_channels=3, n_classes=5, feature_scale=8),
             2, (24, 40), 0)


if __name__ == "__main__":
    main()
