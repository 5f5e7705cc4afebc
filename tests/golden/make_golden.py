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


def _level5_class():
    """Throw-away subclass of the reference's UNet_Nested that switches its commented-out level-5 lines on
    (models/unet.py:224,230,234,237,239,245 construction; :264-265,271,275,278,280,287 forward), built from the
    reference's own unetConv2 / unetUp / init_weights.  Exists only while the fixture is generated."""
    import torch
    from torch import nn
    from models.unet import UNet_Nested, init_weights, unetConv2, unetUp

    class Level5(UNet_Nested):
        def __init__(self, n_classes=4, **kw):
            super().__init__(n_classes=n_classes, **kw)
            f = [int(x / self.feature_scale) for x in (32, 64, 128, 256, 512)]
            extra = dict(conv40=unetConv2(f[3], f[4], self.is_batchnorm),
                         up_concat31=unetUp(f[4], f[3], self.is_deconv),
                         up_concat22=unetUp(f[3], f[2], self.is_deconv, 3),
                         up_concat13=unetUp(f[2], f[1], self.is_deconv, 4),
                         up_concat04=unetUp(f[1], f[0], self.is_deconv, 5),
                         final_4=nn.Conv2d(f[0], n_classes, 1))
            for name, mod in extra.items():
                setattr(self, name, mod)
                for m in mod.modules():
                    if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                        init_weights(m, init_type="kaiming")

        def forward(self, inputs):
            X_00 = self.conv00(inputs)
            X_10 = self.conv10(self.maxpool(X_00))
            X_20 = self.conv20(self.maxpool(X_10))
            X_30 = self.conv30(self.maxpool(X_20))
            X_40 = self.conv40(self.maxpool(X_30))
            X_01 = self.up_concat01(X_10, X_00)
            X_11 = self.up_concat11(X_20, X_10)
            X_21 = self.up_concat21(X_30, X_20)
            X_31 = self.up_concat31(X_40, X_30)
            X_02 = self.up_concat02(X_11, X_00, X_01)
            X_12 = self.up_concat12(X_21, X_10, X_11)
            X_22 = self.up_concat22(X_31, X_20, X_21)
            X_03 = self.up_concat03(X_12, X_00, X_01, X_02)
            X_13 = self.up_concat13(X_22, X_10, X_11, X_12)
            X_04 = self.up_concat04(X_13, X_00, X_01, X_02, X_03)
            return tuple(torch.sigmoid(getattr(self, "final_%d" % j)(self.drop_out(x)))
                         for j, x in ((1, X_01), (2, X_02), (3, X_03), (4, X_04)))

    return Level5


def _truncated_class(depth):
    """The reference's own modules run over the first `depth` levels of its graph only (depth 2 or 3): the
    sub-network the generalised `depth` argument of the build must equal.  Fixture-time only."""
    import torch
    from models.unet import UNet_Nested

    class Truncated(UNet_Nested):
        def forward(self, inputs):
            X = {(0, 0): self.conv00(inputs)}
            for i in range(1, depth):
                X[(i, 0)] = getattr(self, "conv%d0" % i)(self.maxpool(X[(i - 1, 0)]))
            for j in range(1, depth):
                for i in range(depth - j):
                    X[(i, j)] = getattr(self, "up_concat%d%d" % (i, j))(X[(i + 1, j - 1)], *[X[(i, jj)] for jj in range(j)])
            return tuple(torch.sigmoid(getattr(self, "final_%d" % j)(self.drop_out(X[(0, j)]))) for j in range(1, depth))

    return Truncated


def run_depth_case(name, depth, ctor_kwargs, batch, hw, seed):
    """Depth != 4: level 5 from the reference's commented-out lines, depth 2/3 as the truncation of its graph.
    Records only the parameters/buffers the depth-`depth` network owns, under the reference's key names."""
    import torch
    from tools.losses.focal_loss import FocalLoss_BCE_2d

    torch.manual_seed(seed)
    torch.set_num_threads(1)
    cls = _level5_class() if depth == 5 else _truncated_class(depth)
    model = cls(**ctor_kwargs)
    n_cls = model.final_1.out_channels
    cin = ctor_kwargs.get("in_channels", 3)
    h, w = hw
    x = torch.randn(batch, cin, h, w)
    target = torch.rand(batch, n_cls, h, w)

    def owned(key):  # conv{i}0 for i < depth, up_concat{i}{j} for i + j < depth, final_j for j < depth
        head = key.split(".")[0]
        if head.startswith("conv"):
            return int(head[4]) < depth
        if head.startswith("up_concat"):
            return int(head[9]) + int(head[10]) < depth
        return int(head.split("_")[1]) < depth

    blob = {}
    blob.update(_flat("state0", {k: v for k, v in model.state_dict().items() if owned(k)}))
    blob["x"], blob["target"] = x.numpy(), target.numpy()
    blob["meta/ctor"] = np.array(repr(sorted(dict(ctor_kwargs, depth=depth).items())))
    model.eval()
    with torch.no_grad():
        outs = model(x)
    assert len(outs) == depth - 1
    for i, o in enumerate(outs):
        blob["eval_out/%d" % i] = o.numpy()
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
    blob.update(_flat("grad", {k: p.grad for k, p in model.named_parameters() if owned(k)}))
    blob.update(_flat("state1_buffers", {k: b for k, b in model.named_buffers() if owned(k)}))
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **blob)
    print("%-28s %8.1f KB  loss=%.6f" % (name, os.path.getsize(path) / 1024, float(loss)))


def run_plain_unet_case(name, widths, n_classes, n_channels, batch, hw, seed, seeded):
    """The reference's classic UNet (models/unet.py:8-117).  widths=None: the reference class itself (13.4 M
    parameters -> state from tests.helpers.seeded_state, gradients subsampled); else a fixture-time subclass that
    wires the reference's own inconv/down/up/outconv blocks with narrower widths (full state and gradients)."""
    import torch
    from models.unet import UNet, down, inconv, outconv, up
    from tools.losses.focal_loss import FocalLoss_BCE_2d
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from tests.helpers import GRAD_STRIDE, seeded_state

    torch.manual_seed(seed)
    torch.set_num_threads(1)
    if widths is None:
        model = UNet(n_classes=n_classes, n_channels=n_channels)
    else:
        class Narrow(UNet):
            def __init__(self):
                torch.nn.Module.__init__(self)
                w0, w1, w2, w3, w4 = widths
                self.inc = inconv(n_channels, w0)
                self.down1, self.down2, self.down3, self.down4 = down(w0, w1), down(w1, w2), down(w2, w3), down(w3, w4)
                self.up1, self.up2, self.up3, self.up4 = up(w4 + w3, w2), up(w2 + w2, w1), up(w1 + w1, w0), up(w0 + w0, w0)
                self.outc = outconv(w0, n_classes)
        model = Narrow()
    if seeded:
        model.load_state_dict(seeded_state(model, seed))
    h, w = hw
    x = torch.randn(batch, n_channels, h, w)
    target = torch.rand(batch, n_classes, h, w)
    blob = {"x": x.numpy(), "target": target.numpy(),
            "meta/ctor": np.array(repr(sorted(dict(n_classes=n_classes, n_channels=n_channels,
                                                   widths=tuple(widths or (64, 128, 256, 512, 512))).items()))),
            "meta/seeded_state": np.array(seed if seeded else -1)}
    if not seeded:
        blob.update(_flat("state0", model.state_dict()))
    model.eval()
    with torch.no_grad():
        blob["eval_out/0"] = model(x).numpy()
    crit = FocalLoss_BCE_2d(gamma=3, size_average=False)
    model.train()
    model.zero_grad()
    out = model(x)
    loss = crit(out, target)  # trainer/trainer.py:133: a non-tuple output goes to the criterion as it is
    loss.backward()
    blob["train_out/0"] = out.detach().numpy()
    blob["loss"] = loss.detach().numpy()
    for k, p in model.named_parameters():
        g = p.grad.detach()
        if seeded and g.numel() > 4096:
            blob["grad_sub/" + k] = g.reshape(-1)[::GRAD_STRIDE].numpy().copy()
            blob["grad_norm/" + k] = np.array(float(g.double().norm()))
        else:
            blob["grad/" + k] = g.numpy().copy()
    blob.update(_flat("state1_buffers", dict(model.named_buffers())))
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **blob)
    print("%-28s %8.1f KB  loss=%.6f" % (name, os.path.getsize(path) / 1024, float(loss)))


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    warnings.filterwarnings("ignore")
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only in (None, "depth"):
        # depth generalisation: level 5 = the reference's commented-out lines; 2/3 = truncations
        run_depth_case("d5_fs8_rgb5_32x32_b2", 5, dict(in_channels=3, n_classes=5, feature_scale=8), 2, (32, 32), 0)
        # (batch 2 at 32x48: the deepest BatchNorm still sees 2*2*3 = 12 values per channel; with 2 values its
        # normalisation is +-1 whatever the input and the gradients are ill-conditioned in fp32)
        run_depth_case("d5_fs8_bilinear_32x48_b2", 5, dict(in_channels=1, n_classes=4, feature_scale=8, is_deconv=False),
                       2, (32, 48), 1)
        run_depth_case("d3_fs8_32x48_b2", 3, dict(in_channels=1, n_classes=4, feature_scale=8), 2, (32, 48), 0)
        run_depth_case("d2_fs4_64x64_b4", 2, dict(in_channels=1, n_classes=4, feature_scale=4), 4, (64, 64), 0)
        if only == "depth":
            return
    if only in (None, "plain"):
        run_plain_unet_case("unet_w8_rgb5_32x48_b2", (8, 16, 32, 64, 64), 5, 3, 2, (32, 48), 0, seeded=False)
        run_plain_unet_case("unet_ref_rgb5_64x64_b1", None, 5, 3, 1, (64, 64), 3, seeded=True)
        if only == "plain":
            return
    if only in (None, "plain_odd"):
        # sizes that are NOT multiples of 16: MaxPool2d floors (40x56 -> 20x28 -> 10x14 -> 5x7 -> 2x3) and `up` zero-pads
        # the upsampled tensor to its skip partner (models/unet.py:63-70: 4x6 -> 5x7, pad right / bottom)
        run_plain_unet_case("unet_w8_rgb5_40x56_b2", (8, 16, 32, 64, 64), 5, 3, 2, (40, 56), 4, seeded=False)
        if only == "plain_odd":
            return
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
