"""3D U-Net voxel encoder in plain PyTorch (counterpart of generators/unet3d.py:487-827 in the reference; it stays stock
PyTorch-ROCm per the north star).  Voxels (B,4,V,V,V) -> feature volume (B,out_channels,V,V,V) and, with
return_global=True, the spatial mean of the deepest level (B, f_maps * 2**(num_levels-1)) that drives the FiLM mapping.

Module names follow the reference (encoders.{i}.basic_module.SingleConv{1,2}.{groupnorm,conv}, decoders.{i}..., final_conv)
so `encoder_state_dict` of its checkpoints loads; 4,083,592 parameters at f_maps=32, num_levels=4.
Layer order 'gcr': GroupNorm (on the conv's input channels) -> Conv3d 3x3x3 without bias -> ReLU."""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def _gcr(c_in, c_out, num_groups=8):
    groups = num_groups if c_in >= num_groups else 1
    return nn.Sequential(OrderedDict([("groupnorm", nn.GroupNorm(groups, c_in)),
                                      ("conv", nn.Conv3d(c_in, c_out, 3, padding=1, bias=False)),
                                      ("ReLU", nn.ReLU(inplace=True))]))


def _double_conv(c_in, c_out, encoder, num_groups=8):
    """Encoder side widens in two steps (c_in -> max(c_out/2, c_in) -> c_out), decoder side c_in -> c_out -> c_out."""
    mid = max(c_out // 2, c_in) if encoder else c_out
    return nn.Sequential(OrderedDict([("SingleConv1", _gcr(c_in, mid, num_groups)), ("SingleConv2", _gcr(mid, c_out, num_groups))]))


class _Down(nn.Module):
    def __init__(self, c_in, c_out, pool):
        super().__init__()
        self.pooling = nn.MaxPool3d(2) if pool else None
        self.basic_module = _double_conv(c_in, c_out, encoder=True)

    def forward(self, x):
        return self.basic_module(x if self.pooling is None else self.pooling(x))


class _Up(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.basic_module = _double_conv(c_in, c_out, encoder=False)

    def forward(self, skip, x):
        x = F.interpolate(x, size=skip.shape[2:], mode="nearest")
        return self.basic_module(torch.cat((skip, x), dim=1))


class UNet3D(nn.Module):
    def __init__(self, in_channels=4, out_channels=32, f_maps=32, num_levels=4, return_global=True,
                 feature_volume_channels_last=True, **_ignored):
        super().__init__()
        self.feature_volume_channels_last = feature_volume_channels_last
        widths = [f_maps * 2 ** k for k in range(num_levels)] if isinstance(f_maps, int) else list(f_maps)
        self.encoders = nn.ModuleList(_Down(in_channels if i == 0 else widths[i - 1], w, pool=i > 0) for i, w in enumerate(widths))
        rev = widths[::-1]
        self.decoders = nn.ModuleList(_Up(rev[i] + rev[i + 1], rev[i + 1]) for i in range(len(rev) - 1))
        self.final_conv = nn.Conv3d(widths[0], out_channels, 1)
        self.return_global = return_global

    def forward(self, x):
        skips = []
        for enc in self.encoders:
            x = enc(x)
            skips.insert(0, x)
        glob = x.mean(dim=(2, 3, 4)) if self.return_global else None
        for dec, skip in zip(self.decoders, skips[1:]):
            x = dec(skip, x)
        x = self._final(x)
        return (x, glob) if self.return_global else x

    def _final(self, x):
        """The final 1x1x1 convolution (unet3d.py:593), written channel-last: a 1x1x1 convolution is the GEMM
        (B, V^3, f_maps) x (f_maps, out) -- evaluated in that orientation (rocBLAS reads the channel-first activations through
        a transposed operand, no copy) its output IS the (B,V,V,V,C) layout the render kernels gather from, one 128-byte line
        per trilinear corner.  The tensor handed on is the usual (B,C,V,V,V) view of it (torch.channels_last_3d strides), which
        ops.channel_last() takes as it is: the hand-off to the render path costs no transpose kernel and no copy, forward or
        backward.  Same parameters (`final_conv.weight/bias`), same values up to summation order."""
        if not self.feature_volume_channels_last:
            return self.final_conv(x)
        B, C = x.shape[:2]
        w = self.final_conv.weight.reshape(self.final_conv.out_channels, C)
        y = torch.baddbmm(self.final_conv.bias.view(1, 1, -1), x.flatten(2).transpose(1, 2), w.t().unsqueeze(0).expand(B, -1, -1))
        return y.view(B, *x.shape[2:], -1).permute(0, 4, 1, 2, 3)
