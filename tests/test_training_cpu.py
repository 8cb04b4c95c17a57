"""CPU checks of the plain-PyTorch halves of the GAN step: the voxel encoder against the reference's own UNet3D
(golden fixture, state-dict compatible), parameter counts of encoder / discriminator, discriminator
entry resolutions and fade-in."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
import os


def test_parameter_counts_match_the_reference():
    from cnerf_amd.training import UNet3D, ProgressiveDiscriminator
    assert sum(p.numel() for p in UNet3D().parameters()) == 4_083_592           # UNet3D f_maps 32, 4 levels
    assert sum(p.numel() for p in ProgressiveDiscriminator().parameters()) == 12_412_465


def test_encoder_matches_reference_unet3d():
    """Reference generators/unet3d.py::UNet3D(4, 16, f_maps=8, num_levels=3, return_global=True) on a 1x4x8^3 voxel grid:
    its state dict loads into the restatement (strict) and both outputs agree."""
    from cnerf_amd.training import UNet3D
    d = np.load(os.path.join(GOLDEN_DIR, "aux_unet3d_small.npz"))
    net = UNet3D(in_channels=4, out_channels=16, f_maps=8, num_levels=3, return_global=True)
    sd = {k[len("param/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("param/")}
    net.load_state_dict(sd, strict=True)
    net.eval()
    with torch.no_grad():
        fv, glob = net(torch.from_numpy(d["voxel"]))
    assert np.abs(fv.numpy() - d["feature_volume"]).max() < 1e-5
    assert np.abs(glob.numpy() - d["global_feature"]).max() < 1e-5


def test_discriminator_building_blocks_match_reference():
    """CoordConv and the fromRGB adapter against the reference's own classes (golden written by importing
    discriminators/sgdiscriminators.py, whose CoordConv / AddCoords / AdapterBlock text is the one discriminators.py uses):
    same parameter names, same coordinate channels on a non-square input."""
    from cnerf_amd.training.discriminator import CoordConv, _FromRGB
    d = np.load(os.path.join(GOLDEN_DIR, "aux_coordconv.npz"))
    cc = CoordConv(5, 7, kernel_size=3, padding=1)
    cc.load_state_dict({k[len("coordconv/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("coordconv/")}, strict=True)
    ad = _FromRGB(6)
    ad.load_state_dict({k[len("adapter/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("adapter/")}, strict=True)
    with torch.no_grad():
        y, a = cc(torch.from_numpy(d["x"])), ad(torch.from_numpy(d["img"]))
    assert np.abs(y.numpy() - d["y"]).max() < 1e-6
    assert np.abs(a.numpy() - d["adapter_out"]).max() < 1e-6


@pytest.mark.parametrize("res", [32, 64, 128])
def test_discriminator_resolutions_and_fade(res):
    from cnerf_amd.training import ProgressiveDiscriminator
    torch.manual_seed(0)
    d = ProgressiveDiscriminator()
    x = torch.randn(2, 3, res, res)
    full, half = d(x, 1.0), d(x, 0.0)
    assert full.shape == (2, 1) and torch.isfinite(full).all()
    assert not torch.allclose(full, half)            # alpha blends the half-resolution path in
    mid = d(x, 0.5)
    assert torch.isfinite(mid).all()
    # state-dict names the reference's checkpoints use
    keys = d.state_dict().keys()
    assert "layers.0.network.0.conv.weight" in keys and "fromRGB.8.model.0.bias" in keys and "final_layer.weight" in keys
