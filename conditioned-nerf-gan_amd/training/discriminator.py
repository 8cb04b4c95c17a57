"""Progressive-growing CoordConv residual discriminators in plain PyTorch.

ProgressiveDiscriminator: counterpart of discriminators/discriminators.py:39-199 in the reference, which cannot be imported
here (it pulls in tkinter, which this image lacks).  CCSDiscriminator: counterpart of discriminators/sgdiscriminators.py:234-306
(the "sgdiscriminator" BASELINE.json names) -- the same assembly (fromRGB adapters per resolution, residual CoordConv blocks,
fade-in of the half-resolution image after the first block, 2x2 final convolution) with strided blocks; that file IS importable,
so this class is pinned against the reference itself (tests/golden/aux_ccs_discriminator.npz: parameters under a fixed seed,
outputs at 32/64/128 px for alpha 0 / 0.5 / 1), and with it the fade-in / entry-resolution logic both classes share.
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


def _kaiming_leaky(module):
    """kaiming_leaky_init of the reference (sgdiscriminators.py:25-28), applied to the Conv2d / Linear leaves of `module`."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            torch.nn.init.kaiming_normal_(m.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")


class ResidualCCBlock(nn.Module):
    """CoordConv 3x3 -> LeakyReLU -> CoordConv 3x3 stride 2 -> LeakyReLU, plus a strided 1x1 projection of the input; /sqrt(2)
    (sgdiscriminators.py:234-254)."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.network = nn.Sequential(CoordConv(c_in, c_out, kernel_size=3, padding=1), nn.LeakyReLU(0.2, inplace=True),
                                     CoordConv(c_out, c_out, kernel_size=3, stride=2, padding=1), nn.LeakyReLU(0.2, inplace=True))
        _kaiming_leaky(self.network)
        self.proj = nn.Conv2d(c_in, c_out, 1, stride=2)

    def forward(self, x):
        return (self.network(x) + self.proj(x)) / math.sqrt(2)


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


class CCSDiscriminator(nn.Module):
    """sgdiscriminators.py:256-306: images of 4 .. 256 px, seven strided residual CoordConv blocks; returns (prediction, None,
    None) like the reference.  `pose_layer` is a parameter the reference constructs and never uses (kept for its state dict)."""
    WIDTHS = (32, 64, 128, 256, 400, 400, 400, 400)            # channels at 256, 128, ..., 2 pixels

    def __init__(self, **_ignored):
        super().__init__()
        self.epoch = 0
        self.step = 0
        w = self.WIDTHS
        self.layers = nn.ModuleList(ResidualCCBlock(w[i], w[i + 1]) for i in range(7))
        self.fromRGB = nn.ModuleList(_FromRGB(c) for c in w)
        self.final_layer = nn.Conv2d(400, 1, 2)
        self.pose_layer = nn.Linear(2, 400)

    def forward(self, img, alpha, options=None, **kwargs):
        start = 8 - int(math.log2(img.shape[-1]))              # 256 px -> block 0, 2 px -> block 7 (no block)
        x = self.fromRGB[start](img)
        if kwargs.get("instance_noise", 0) > 0:
            x = x + torch.randn_like(x) * kwargs["instance_noise"]
        for i, layer in enumerate(self.layers[start:]):
            if i == 1 and alpha < 1:
                x = alpha * x + (1 - alpha) * self.fromRGB[start + 1](F.interpolate(img, scale_factor=0.5, mode="nearest"))
            x = layer(x)
        return self.final_layer(x).reshape(x.shape[0], 1), None, None
