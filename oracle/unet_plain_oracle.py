"""CPU oracle: a plain-PyTorch restatement of the reference's classic U-Net.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates ``/root/reference/models/unet.py:8-117``
(double_conv, inconv, down, up, outconv, UNet) with exactly the reference's ``state_dict`` key
names and shapes.  ``widths`` generalises the hard-coded (64, 128, 256, 512, 512) of
models/unet.py:95-99 so that small fixtures can pin the block semantics; the default is the reference.

The arithmetic lives in torch.nn (ATen / oneDNN on CPU) -- the same third-party code the reference
delegates to.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

REFERENCE_WIDTHS = (64, 128, 256, 512, 512)  # models/unet.py:95-99


class DoubleConvOracle(nn.Module):
    """models/unet.py:8-25: (conv3x3 p1 => BN => ReLU) * 2 inside ONE nn.Sequential (indices 0,1,3,4)."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
            nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv(x)


class InConvOracle(nn.Module):
    """models/unet.py:28-35"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = DoubleConvOracle(cin, cout)

    def forward(self, x):
        return self.conv(x)


class DownOracle(nn.Module):
    """models/unet.py:38-48: MaxPool2d(2) then double_conv (Sequential indices 0, 1)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.mpconv = nn.Sequential(nn.MaxPool2d(2), DoubleConvOracle(cin, cout))

    def forward(self, x):
        return self.mpconv(x)


class UpOracle(nn.Module):
    """models/unet.py:51-82 with bilinear=True (the only form UNet builds): x1 is upsampled x2 (align_corners=True),
    zero-padded to x2's size (diff//2 before, the rest after), concatenated AFTER x2, then double_conv."""

    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv = DoubleConvOracle(cin, cout)

    def forward(self, x1, x2):
        x1 = self.up(x1)
        dy = x2.size(2) - x1.size(2)
        dx = x2.size(3) - x1.size(3)
        x1 = F.pad(x1, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        return self.conv(torch.cat([x2, x1], dim=1))


class OutConvOracle(nn.Module):
    """models/unet.py:85-92"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        return self.conv(x)


class UNetOracle(nn.Module):
    """models/unet.py:94-117.  Default initialisation is torch's own (the reference never re-initialises UNet)."""

    def __init__(self, n_classes=5, n_channels=3, widths=REFERENCE_WIDTHS):
        super().__init__()
        w0, w1, w2, w3, w4 = widths
        self.widths = tuple(widths)
        self.inc = InConvOracle(n_channels, w0)
        self.down1 = DownOracle(w0, w1)
        self.down2 = DownOracle(w1, w2)
        self.down3 = DownOracle(w2, w3)
        self.down4 = DownOracle(w3, w4)
        self.up1 = UpOracle(w4 + w3, w2)
        self.up2 = UpOracle(w2 + w2, w1)
        self.up3 = UpOracle(w1 + w1, w0)
        self.up4 = UpOracle(w0 + w0, w0)
        self.outc = OutConvOracle(w0, n_classes)

    def forward(self, x):
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        x = self.up1(x5, x4)
        x = self.up2(x, x3)
        x = self.up3(x, x2)
        x = self.up4(x, x1)
        return torch.sigmoid(self.outc(x))
