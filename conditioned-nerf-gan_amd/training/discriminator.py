"""Progressive-growing CoordConv residual discriminator in plain PyTorch (counterpart of
discriminators/discriminators.py:39-199 in the reference, which cannot even be imported here: it pulls in tkinter).
Images (B,3,R,R) with R a power of two in [2,512] enter at the block matching their resolution; during a fade-in the
half-resolution image is blended in after the first block with weight 1-alpha.  Names follow the reference
(layers.{i}.network.{0,2}.conv, layers.{i}.proj, fromRGB.{i}.model.0, final_layer): 12,412,465 parameters."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

WIDTHS = (16, 32, 64, 128, 256, 400, 400, 400, 400)      # channels at 512, 256, ..., 2 pixels


class CoordConv(nn.Module):
    """Conv2d over the input plus two channels holding the pixel coordinates in [-1, 1]."""

    def __init__(self, c_in, c_out, **conv_kw):
        super().__init__()
        self.conv = nn.Conv2d(c_in + 2, c_out, **conv_kw)

    def forward(self, x):
        b, _, h, w = x.shape
        ys = torch.linspace(-1, 1, h, device=x.device, dtype=x.dtype).view(1, 1, h, 1).expand(b, 1, h, w)
        xs = torch.linspace(-1, 1, w, device=x.device, dtype=x.dtype).view(1, 1, 1, w).expand(b, 1, h, w)
        # channel order of the reference: first the coordinate that varies along dim 2 after its transposes (rows), then columns
        return self.conv(torch.cat([x, ys, xs], dim=1))


class ResidualCoordConvBlock(nn.Module):
    def __init__(self, c_in, c_out, downsample=True):
        super().__init__()
        self.network = nn.Sequential(CoordConv(c_in, c_out, kernel_size=3, padding=1), nn.LeakyReLU(0.2, inplace=True),
                                     CoordConv(c_out, c_out, kernel_size=3, padding=1), nn.LeakyReLU(0.2, inplace=True))
        self.proj = nn.Conv2d(c_in, c_out, 1) if c_in != c_out else None
        self.downsample = downsample

    def forward(self, x):
        y = self.network(x)
        if self.downsample:
            y, x = F.avg_pool2d(y, 2), F.avg_pool2d(x, 2)
        if self.proj is not None:
            x = self.proj(x)
        return (y + x) / math.sqrt(2)


class _FromRGB(nn.Module):
    def __init__(self, c_out):
        super().__init__()
        self.model = nn.Sequential(nn.Conv2d(3, c_out, 1), nn.LeakyReLU(0.2))

    def forward(self, x):
        return self.model(x)


class ProgressiveDiscriminator(nn.Module):
    def __init__(self, **_ignored):
        super().__init__()
        self.epoch = 0
        self.step = 0
        self.layers = nn.ModuleList(ResidualCoordConvBlock(WIDTHS[i], WIDTHS[i + 1]) for i in range(8))
        self.fromRGB = nn.ModuleList(_FromRGB(w) for w in WIDTHS)
        self.final_layer = nn.Conv2d(400, 1, 2)

    def forward(self, img, alpha, **_ignored):
        start = 9 - int(math.log2(img.shape[-1]))              # 512 px -> block 0, 2 px -> block 8 (no block)
        x = self.fromRGB[start](img)
        for i, layer in enumerate(self.layers[start:]):
            if i == 1:
                x = alpha * x + (1 - alpha) * self.fromRGB[start + 1](F.interpolate(img, scale_factor=0.5, mode="nearest"))
            x = layer(x)
        return self.final_layer(x).reshape(x.shape[0], 1)
