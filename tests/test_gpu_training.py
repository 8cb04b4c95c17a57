"""GPU smoke of the GAN step harness: D step with R1, G step with split accumulation through the HIP
render path and its backward, plain and under DDP with a one-rank RCCL process group."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(ddp):
    import cnerf_amd
    from cnerf_amd.training import GanTrainer, default_metadata
    from cnerf_amd.training.gan_step import synthetic_sample
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    md = default_metadata(img_size=32, num_steps=12, batch_size=4, batch_split=2, hidden_dim=64)
    md["unet"].update(f_maps=8, num_levels=3)
    md["generator"]["z_dim"] = 32                      # deepest encoder level: 8 * 2**2 channels
    tr = GanTrainer(md, dev, ddp=ddp)
    before = {k: v.detach().clone() for k, v in tr.generator.state_dict().items()}
    enc_before = tr.encoder.final_conv.weight.detach().clone()
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        tr.step(synthetic_sample(4, 32, 16, dev, g))
    assert all(map(lambda x: x == x and abs(x) < 1e4, tr.losses["d"] + tr.losses["g"] + tr.losses["photo"]))
    changed = [k for k, v in tr.generator.state_dict().items() if not torch.equal(v, before[k])]
    assert len(changed) == len(before), "every generator parameter receives a gradient through the HIP backward"
    assert not torch.equal(tr.encoder.final_conv.weight, enc_before), "the encoder is trained through d(feature volume)"
    assert tr.generator.step == 2 and 0 < tr.alpha <= 1 and md["nerf_noise"] < 1.0


def test_gan_step_single_process():
    _run(ddp=False)


@pytest.mark.parametrize("precisions", [("fp32", "fp32"), ("fp16x3", "fp16")])
def test_gan_step_at_the_size_of_baseline_config_3(precisions):
    """BASELINE config 3 as it is written: one full GAN step (UNet3D on 64^3 voxel grids -> 32 x 64^3 feature volume + global feature,
    SHORTSIREN_FG hidden 256 rendered at 128x128 rays x (64 + 64) samples, ProgressiveDiscriminator with R1, Adam) -- batch 2, the
    reference's chunk size at this stage (configs/thousand/special.py:24-30) -- through the HIP forward and backward, in the exact
    arithmetic and in the fast one: finite losses, every generator parameter and the encoder updated, kept activations released.
    (Convolutions in MIOpen's immediate mode: the kernel search that train.py turns on takes minutes.)"""
    import cnerf_amd
    from cnerf_amd.training import GanTrainer, default_metadata
    from cnerf_amd.training.gan_step import synthetic_sample
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    md = default_metadata(img_size=128, num_steps=64, batch_size=2, batch_split=1, hidden_dim=256)
    md["render_precision"], md["backward_precision"] = precisions
    tr = GanTrainer(md, dev)
    before = {k: v.detach().clone() for k, v in tr.generator.state_dict().items()}
    enc_before = tr.encoder.final_conv.weight.detach().clone()
    torch.cuda.reset_peak_memory_stats()
    tr.step(synthetic_sample(2, 128, 64, dev, torch.Generator().manual_seed(3)))
    torch.cuda.synchronize()
    assert all(x == x and abs(x) < 1e4 for x in tr.losses["d"] + tr.losses["g"] + tr.losses["photo"])
    assert all(not torch.equal(v, before[k]) for k, v in tr.generator.state_dict().items())
    assert not torch.equal(tr.encoder.final_conv.weight, enc_before)
    assert 0 < tr.last["g_grad_norm"] < 1e6 and 0 < tr.last["e_grad_norm"] < 1e6     # finite, non-zero gradients reached both networks
    assert torch.cuda.max_memory_allocated() < 120 << 30


@pytest.mark.parametrize("precisions", [("fp32", "fp32"), ("fp16x3", "fp16")])
def test_gan_step_with_the_per_point_film_generator(precisions):
    """`--siren-type TALLSIREN` (one of the SIREN classes the reference's shipped configs still resolve): the encoder hands
    over the bare feature volume, the mapping MLP runs per point inside the kernels.  One GAN step at 32x32 rays x (12 + 12) samples,
    hidden 64, in the exact arithmetic and in the fast one (field_pw16 / chain_pw16): finite losses, every generator parameter --
    the mapping network included -- and the encoder updated; the two arithmetics' generator gradient norms agree to 2 %."""
    import cnerf_amd
    from cnerf_amd.training import GanTrainer, default_metadata
    from cnerf_amd.training.gan_step import synthetic_sample
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    md = default_metadata(img_size=32, num_steps=12, batch_size=2, batch_split=1, siren_type="TALLSIREN", hidden_dim=64)
    assert md["generator"]["input_dim"] == 3 and md["generator"]["z_dim"] == 32 and md["unet"]["return_global"] is False
    md["render_precision"], md["backward_precision"] = precisions
    tr = GanTrainer(md, dev)
    before = {k: v.detach().clone() for k, v in tr.generator.state_dict().items()}
    enc_before = tr.encoder.final_conv.weight.detach().clone()
    tr.step(synthetic_sample(2, 32, 64, dev, torch.Generator().manual_seed(3)))
    torch.cuda.synchronize()
    assert all(x == x and abs(x) < 1e4 for x in tr.losses["d"] + tr.losses["g"] + tr.losses["photo"])
    assert all(not torch.equal(v, before[k]) for k, v in tr.generator.state_dict().items())
    assert not torch.equal(tr.encoder.final_conv.weight, enc_before)
    assert 0 < tr.last["g_grad_norm"] < 1e6 and 0 < tr.last["e_grad_norm"] < 1e6
    NORMS[precisions] = tr.last["g_grad_norm"]
    if len(NORMS) == 2:
        a, b = NORMS[("fp32", "fp32")], NORMS[("fp16x3", "fp16")]
        assert abs(a - b) <= 2e-2 * a, (a, b)


NORMS = {}


def test_no_grad_render_runs_the_plain_forward(monkeypatch):
    """The D step renders under torch.no_grad() with the generator in training mode and its parameters requiring grad: nothing will
    back-propagate, so the forward must not keep activations (the activation-storing kernel is 1.5 x the plain one and its buffers are
    GiBs).  Regression: inside autograd.Function.forward `ctx.needs_input_grad` is True for parameters whatever the grad mode."""
    import cnerf_amd
    from cnerf_amd import ops
    from cnerf_amd.generators import ImplicitGenerator3d
    dev = torch.device("cuda:0")
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 32, 32, 4, 64).to(dev)
    gen.set_device(dev)
    gen.siren.precision, gen.siren.backward_precision = "fp16x3", "fp16"
    gen.train()
    calls = []
    real = ops.resident_act16
    monkeypatch.setattr(ops, "resident_act16", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    z = (torch.randn(1, 32, 8, 8, 8, device=dev), torch.randn(1, 32, device=dev))
    cam = torch.eye(4, device=dev).unsqueeze(0)
    kw = dict(clamp_mode="relu", nerf_noise=0.0, white_back=True)
    with torch.no_grad():
        gen(z, cam, 8, 30.0, 0.5, 1.5, 6, True, **kw)
    assert calls == []
    px, _ = gen(z, cam, 8, 30.0, 0.5, 1.5, 6, True, **kw)
    assert calls == [1] and px.requires_grad


def test_gan_step_ddp_one_rank_rccl():
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        _run(ddp=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_gan_step_matches_reference_modules(precision):
    """One D step and one G step through the HIP render path (forward and backward) against tests/golden/aux_gan_step.npz: the
    same step run on the CPU with the reference's own generator, UNet3D and CCSDiscriminator modules, fp32 (step logic of
    utils.py:621-842 spelled out in tests/golden/make_golden.py::build_gan_step -- that file cannot be imported).  Same
    parameters (generator / encoder from the fixture, discriminator rebuilt under its seed and checked), same voxels, images,
    cameras and random draws; the renders run free (nothing forced).  Pinned: D loss, R1 penalty, G loss, photometric loss,
    the pre-clip gradient norms of all three networks, the post-step parameters of the generator."""
    import numpy as np
    from conftest import GOLDEN_DIR
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
    from cnerf_amd.training import GanTrainer, default_metadata, UNet3D, CCSDiscriminator
    d = np.load(os.path.join(GOLDEN_DIR, "aux_gan_step.npz"))
    dev = torch.device("cuda:0")
    T = lambda k: torch.from_numpy(np.asarray(d[k]))
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 32, 32, 4, 64)
    gen.load_state_dict({k[len("gen/"):]: T(k) for k in d.files if k.startswith("gen/")}, strict=True)
    enc = UNet3D(in_channels=4, out_channels=32, f_maps=8, num_levels=3, return_global=True)
    enc.load_state_dict({k[len("enc/"):]: T(k) for k in d.files if k.startswith("enc/")}, strict=True)
    torch.manual_seed(123)
    disc = CCSDiscriminator()
    for k, v in disc.state_dict().items():
        f = v.double().flatten()
        mine = np.array([f.sum().item(), f.abs().sum().item(), *f[:3].tolist(), *([0.0] * max(0, 3 - f.numel()))][:5])
        assert np.allclose(mine, d["dstat/" + k], rtol=1e-12, atol=1e-12), k
    md = default_metadata(img_size=16, num_steps=8, batch_size=2, batch_split=1, hidden_dim=64)
    md["discriminator"] = "CCSDiscriminator"
    tr = GanTrainer(md, dev, modules={"generator": gen, "encoder": enc, "discriminator": disc})
    tr.generator.set_device(dev)
    tr.generator.siren.precision = precision
    tr.generator.train(); tr.encoder.train(); tr.discriminator.train()
    tr.generator.step = 1000
    tr.set_alpha()
    assert abs(tr.alpha - float(d["alpha"])) < 1e-7 and abs(md["nerf_noise"] - float(d["nerf_noise"])) < 1e-6
    tr.render_rng = lambda chunk, phase: {k: T(f"{phase}/{k}").to(dev) for k in ("u_strat", "u_fine", "eps_coarse", "eps_final")}
    sample = {"voxel": T("voxel"), "img": T("img"), "cam2world": T("cam2world")}
    np.random.seed(7)                                     # the D step samples its cameras from NumPy (bit-exact helper)
    tr.train_discriminator(sample)
    tr.train_generator(sample)
    got = dict(tr.last)
    print(precision, {k: (got[k], float(d[k])) for k in ("d_loss", "r1_penalty", "d_grad_norm", "fake_mean", "g_loss", "photo_loss",
                                                          "g_grad_norm", "e_grad_norm")})
    rel = lambda k: abs(got[k] - float(d[k])) / max(abs(float(d[k])), 1e-6)
    # measured on an MI355X (both precisions): d_loss 2e-6, r1 2e-7, g_loss 1e-5, photo 2.2e-4 (free-running renders: the
    # resampled depths differ at the 1e-5 level), gradient norms 4e-5 .. 3.1e-3
    for k in ("d_loss", "r1_penalty", "g_loss"):
        assert rel(k) < 1e-4, (k, got[k], float(d[k]))
    assert rel("photo_loss") < 1e-3
    assert abs(got["fake_mean"] - float(d["fake_mean"])) < 1e-4
    for k in ("d_grad_norm", "g_grad_norm", "e_grad_norm"):
        assert rel(k) < 1e-2, (k, got[k], float(d[k]))
    # Adam's first step moves every parameter by ~lr * sign(gradient): sums after the step pin the signs of the gradients
    for k, v in tr.generator.state_dict().items():
        ref = float(d["gen_after_sum/" + k])
        assert abs(v.double().sum().item() - ref) < 2e-3 * max(1.0, abs(ref)) + 5e-5 * v.numel() * 0.02, k


def test_encoder_hands_its_volume_over_channel_last():
    """the encoder's final 1x1x1 convolution writes the feature volume channel-last, the render kernels take it
    without a transpose or a copy (forward), and d(loss)/d(volume) goes back in the same layout (backward).  Same image and
    same encoder gradients as with the channel-first output + transpose kernels."""
    import cnerf_amd
    from cnerf_amd import ops
    from cnerf_amd.generators import ImplicitGenerator3d
    from cnerf_amd.training import UNet3D
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    enc = UNet3D(in_channels=4, out_channels=32, f_maps=8, num_levels=2, return_global=True).to(dev)
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 16, 32, 4, 64).to(dev)
    gen.set_device(dev)
    gen.train()
    vox = torch.rand(2, 4, 8, 8, 8, device=dev)
    cam = torch.eye(4, device=dev).unsqueeze(0).repeat(2, 1, 1).contiguous()
    cam[:, 2, 3] = -1.0
    rng = {"u_strat": torch.rand(2, 36, 8, device=dev), "u_fine": torch.rand(2, 36, 8, device=dev)}
    outs = []
    for cl in (True, False):
        enc.feature_volume_channels_last = cl
        enc.zero_grad()
        fv, glob = enc(vox)
        assert (ops.channel_last(fv).data_ptr() == fv.data_ptr()) == cl          # zero-copy exactly when written channel-last
        px, dp = gen((fv, glob), cam, 6, 49.13, 0.25, 1.95, 8, True, clamp_mode="relu", nerf_noise=0.0, _rng=rng)
        (px.square().mean() + dp.mean()).backward()
        outs.append((px.detach().clone(), enc.final_conv.weight.grad.clone(), enc.encoders[0].basic_module.SingleConv1.conv.weight.grad.clone()))
    (pa, ga, ea), (pb, gb, eb) = outs
    assert (pa - pb).abs().max().item() < 1e-5
    # (two fp32 evaluation orders -- baddbmm vs Conv3d -- downstream of a volume gradient accumulated with float atomics, whose
    # order varies from run to run: agreement is at the 1e-4 level of the tensor's scale, not bit-wise)
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()
    assert rel(ga, gb) < 1e-4 and (ga - gb).abs().max().item() < 1e-3 * gb.abs().max().item()
    assert rel(ea, eb) < 1e-3 and (ea - eb).abs().max().item() < 1e-2 * eb.abs().max().item()
