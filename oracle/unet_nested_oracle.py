"""CPU oracle: a plain-PyTorch restatement of the reference's nested U-Net.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every class cites the lines of
``/root/reference/models/unet.py`` it restates.  The module tree is laid out so
that ``state_dict()`` carries exactly the reference's key names and shapes for
depth 4; depth 2/3/5 follow the pattern of the reference's commented-out
level-5 lines (models/unet.py:224,230,234,237,239,245,264-265,271,275,278,280,287).

The arithmetic itself lives in torch.nn (ATen / oneDNN on CPU) -- the same
third-party code the reference delegates to.
"""
from __future__ import annotations

import torch
from torch import nn

_BASE_WIDTHS = (32, 64, 128, 256, 512)  # models/unet.py:215


def _kaiming_like_reference(module: nn.Module) -> None:
    """models/unet.py:165-174 -- kaiming-normal (fan_in) for anything whose class
    name contains 'Conv' or 'Linear'; BatchNorm gamma ~ N(1, 0.02), beta = 0."""
    name = type(module).__name__
    if "Conv" in name or "Linear" in name:
        nn.init.kaiming_normal_(module.weight.data, a=0, mode="fan_in")
    elif "BatchNorm" in name:
        nn.init.normal_(module.weight.data, 1.0, 0.02)
        nn.init.constant_(module.bias.data, 0.0)


class ConvPairOracle(nn.Module):
    """models/unet.py:121-156 (unetConv2): n x [Conv3x3 p1 s1 (+BN) + ReLU]."""

    def __init__(self, cin: int, cout: int, with_bn: bool, n: int = 2):
        super().__init__()
        self.n = n
        for idx in range(1, n + 1):
            layers = [nn.Conv2d(cin, cout, 3, 1, 1)]
            if with_bn:
                layers.append(nn.BatchNorm2d(cout))
            layers.append(nn.ReLU(inplace=True))
            self.add_module("conv%d" % idx, nn.Sequential(*layers))
            cin = cout
        for child in self.children():  # models/unet.py:146-148
            child.apply(_kaiming_like_reference)

    def forward(self, x):
        for idx in range(1, self.n + 1):
            x = getattr(self, "conv%d" % idx)(x)
        return x


class UpJoinOracle(nn.Module):
    """models/unet.py:182-202 (unetUp): up(high) first, then every same-level
    node in column order, channel-concatenated, then a BN-less conv pair."""

    def __init__(self, c_high: int, c_out: int, use_deconv: bool, n_concat: int = 2):
        super().__init__()
        self.conv = ConvPairOracle(c_high + (n_concat - 2) * c_out, c_out, False)
        if use_deconv:
            self.up = nn.ConvTranspose2d(c_high, c_out, kernel_size=2, stride=2, padding=0)
        else:
            self.up = nn.Sequential(nn.UpsamplingBilinear2d(scale_factor=2),
                                    nn.Conv2d(c_high, c_out, 1))
        for child in self.children():  # models/unet.py:194-196
            if isinstance(child, ConvPairOracle):
                continue
            child.apply(_kaiming_like_reference)

    def forward(self, high, *lows):
        joined = self.up(high)
        for low in lows:
            joined = torch.cat([joined, low], 1)
        return self.conv(joined)


class UNetNestedOracle(nn.Module):
    """models/unet.py:204-300 (UNet_Nested), generalised over ``depth``.

    depth = number of resolution levels (reference as shipped: 4).  Node
    X[i][j] exists for i + j <= depth - 1; heads final_1 .. final_{depth-1}.
    """

    def __init__(self, in_channels=3, n_classes=4, feature_scale=2, is_deconv=True,
                 is_batchnorm=True, is_ds=True, depth=4):
        super().__init__()
        if not 2 <= depth <= len(_BASE_WIDTHS):
            raise ValueError("depth must be in 2..5")
        self.in_channels = in_channels
        self.feature_scale = feature_scale
        self.is_deconv = is_deconv
        self.is_batchnorm = is_batchnorm
        self.is_ds = is_ds
        self.depth = depth
        f = [int(w / feature_scale) for w in _BASE_WIDTHS]  # models/unet.py:216

        self.maxpool = nn.MaxPool2d(kernel_size=2)  # :219
        cin = in_channels
        for i in range(depth):  # :220-224
            setattr(self, "conv%d0" % i, ConvPairOracle(cin, f[i], is_batchnorm))
            cin = f[i]
        for j in range(1, depth):  # :227-239, column by column
            for i in range(depth - j):
                setattr(self, "up_concat%d%d" % (i, j),
                        UpJoinOracle(f[i + 1], f[i], is_deconv, j + 1))
        for j in range(1, depth):  # :242-245
            setattr(self, "final_%d" % j, nn.Conv2d(f[0], n_classes, 1))
        for m in self.modules():  # :248-252 second init pass (Conv2d / BatchNorm2d only)
            if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                m.apply(_kaiming_like_reference)
        self.drop_out = nn.Dropout(p=0.4)  # :254

    def forward(self, inputs):
        d = self.depth
        X = [[None] * d for _ in range(d)]
        X[0][0] = self.conv00(inputs)  # :257
        for i in range(1, d):  # :258-265
            X[i][0] = getattr(self, "conv%d0" % i)(self.maxpool(X[i - 1][0]))
        for j in range(1, d):  # :268-280
            for i in range(d - j):
                up = getattr(self, "up_concat%d%d" % (i, j))
                X[i][j] = up(X[i + 1][j - 1], *X[i][:j])
        outs = []
        for j in range(1, d):  # :283-287
            head = getattr(self, "final_%d" % j)
            outs.append(torch.sigmoid(head(self.drop_out(X[0][j]))))
        return tuple(outs)  # :300
