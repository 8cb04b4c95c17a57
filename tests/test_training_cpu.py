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


def test_on_disk_formats_round_trip(tmp_path):
    """voxel.npz / cameras.npz readers (datasets.py:104-148) and the checkpoint dictionary (utils.py:463-501, key names of the
    reference) on synthetic files: layout conventions and a save / load round trip through weights_only loading."""
    import types
    from cnerf_amd.training import load_voxel_npz, load_cam2world, save_checkpoint, load_checkpoint, UNet3D, ProgressiveDiscriminator
    from cnerf_amd.generators import ImplicitGenerator3d
    rs = np.random.RandomState(0)
    vox = rs.rand(8, 8, 8, 4).astype(np.float32)                     # (X, Y, Z, [occ, r, g, b])
    np.savez(tmp_path / "voxel.npz", voxel=vox)
    t = load_voxel_npz(tmp_path / "voxel.npz", resolution=8)
    assert t.shape == (4, 8, 8, 8) and t.dtype == torch.float32
    assert float(t[1, 2, 3, 5]) == float(vox[5, 3, 2, 1])           # (c, z, y, x) <- (x, y, z, c)
    with pytest.raises(ValueError):
        load_voxel_npz(tmp_path / "voxel.npz", resolution=64)
    mats = {f"world_mat_inv_{i}": rs.rand(4, 4) for i in range(3)}
    np.savez(tmp_path / "cameras.npz", **mats)
    m = load_cam2world(tmp_path / "cameras.npz", 2)
    assert m.dtype == torch.float32 and np.allclose(m.numpy(), mats["world_mat_inv_2"].astype(np.float32))

    def make():
        torch.manual_seed(0)
        tr = types.SimpleNamespace()
        tr.device = torch.device("cpu")
        tr.metadata = {"enable_discriminator": True}
        tr.generator = ImplicitGenerator3d("SHORTSIREN_FG", 16, 32, 4, 64)
        tr.encoder = UNet3D(in_channels=4, out_channels=32, f_maps=8, num_levels=2, return_global=True)
        tr.discriminator = ProgressiveDiscriminator()
        for name, mod in (("G", tr.generator), ("E", tr.encoder), ("D", tr.discriminator)):
            setattr(tr, "optimizer_" + name, torch.optim.Adam(mod.parameters(), lr=1e-3, betas=(0.0, 0.9)))
        tr.losses = {"g": [0.5], "d": [1.25], "photo": [0.1]}
        return tr

    a = make()
    a.generator.step = 41
    with torch.no_grad():
        a.generator.siren.final_layer.bias += 1.0
    path = save_checkpoint(a, tmp_path / "checkpoints")
    assert path.endswith("41.tar")
    raw = torch.load(path, weights_only=True)
    # exactly what the reference's Trainer.save_models writes with photo_loss and the discriminator on, depth_loss off
    # (utils.py:473-501) -- its load_models reads scaler_state_dict and the four *_val / *_test histories unconditionally
    ref_keys = {"step", "generator_state_dict", "optimizer_G_state_dict", "scaler_state_dict", "encoder_state_dict",
                "optimizer_E_state_dict", "photometry_losses", "photometry_losses_val", "depth_losses_val",
                "photometry_losses_test", "depth_losses_test", "discriminator_state_dict", "optimizer_D_state_dict",
                "generator_losses", "discriminator_losses"}
    assert set(raw) == ref_keys
    torch.amp.GradScaler("cpu", enabled=True).load_state_dict(raw["scaler_state_dict"])     # the call utils.py:336 makes
    assert "siren.network.0.layer.weight" in raw["generator_state_dict"]
    b = make()
    load_checkpoint(b, path)
    assert b.generator.step == 41
    assert torch.equal(b.generator.siren.final_layer.bias, a.generator.siren.final_layer.bias)
    assert b.losses == {"g": [0.5], "d": [1.25], "photo": [0.1]} and b.eval_losses["depth_losses_test"] == []
    # a dictionary with the reference's key set (as its trainer would have written it, AMP scaler state included) loads too
    ref_like = dict(raw)
    ref_like["scaler_state_dict"] = {"scale": 1024.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 7}
    ref_like["photometry_losses_val"] = [0.3, 0.2]
    torch.save(ref_like, tmp_path / "ref_like.tar")
    c = make()
    load_checkpoint(c, tmp_path / "ref_like.tar")
    assert c.eval_losses["photometry_losses_val"] == [0.3, 0.2] and c.generator.step == 41


def test_ccs_discriminator_matches_reference():
    """CCSDiscriminator against discriminators/sgdiscriminators.py of the reference (tests/golden/make_golden.py ccs): the same
    torch seed gives the same 12.9 M parameters (construction order and initialisers are the reference's), and the outputs
    at 32 / 64 / 128 px agree for alpha 0, 0.5 and 1 -- entry block per resolution, strided residual blocks, fade-in of the
    half-resolution image after the first block (skipped at alpha = 1), 2x2 final convolution."""
    from cnerf_amd.training import CCSDiscriminator
    d = np.load(os.path.join(GOLDEN_DIR, "aux_ccs_discriminator.npz"))
    torch.manual_seed(0)
    net = CCSDiscriminator()
    net.eval()
    sd = net.state_dict()
    stats = {k[len("stat/"):]: d[k] for k in d.files if k.startswith("stat/")}
    assert set(sd) == set(stats)
    assert sum(p.numel() for p in net.parameters()) == 12_879_217
    for k, v in sd.items():
        f = v.double().flatten()
        mine = np.array([f.sum().item(), f.abs().sum().item(), *f[:3].tolist(), *([0.0] * max(0, 3 - f.numel()))][:5])
        assert np.allclose(mine, stats[k], rtol=1e-12, atol=1e-12), k
    for res in (32, 64, 128):
        x = torch.from_numpy(d[f"img_{res}"])
        for alpha in (0.0, 0.5, 1.0):
            with torch.no_grad():
                pred, a, b = net(x, alpha)
            assert a is None and b is None
            assert np.abs(pred.numpy() - d[f"pred_{res}_{alpha}"]).max() < 1e-5, (res, alpha)
