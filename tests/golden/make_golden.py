#!/usr/bin/env python3
"""Generate golden vectors for the render path by IMPORTING the reference.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  Output: small .npz files next to this script; they hold DATA only
(inputs, the reference's intermediates and outputs) -- no reference source.

Captured per fixture:
  inputs   : feature_volume (B,C,V,V,V channel-first, as the encoder emits),
             global_feature, cam2worlds, every parameter of generator.siren
             (state-dict names), the RNG tensors in draw order
             (u_strat, eps_coarse, u_fine, eps_final)
  interm.  : world points / jittered z of the coarse pass, looked-up features,
             coarse rgb_sigma, coarse weights, cdf, inds, fine z, fine rgb_sigma,
             sort indices, final weights
  outputs  : pixels, depth_map
  backward : grads of  pixels.square().mean() + depth.mean()  w.r.t. every
             parameter, feature_volume, global_feature

Usage:  python tests/golden/make_golden.py [name ...]
"""
import os
import sys
import math

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

FOV = 49.134342641202636
RAY_START, RAY_END = 0.25, 1.95

# name -> spec.  "full": store all intermediates + grads; otherwise a slim set.
FIXTURES = {
    # cfg-1 shape (32x32x12, B=2), primary variant, full hidden width
    "short_fg_32x12": dict(variant="SHORTSIREN_FG", B=2, R=32, S=12, V=16, C=32, H=256, Z=256,
                           noise=0.0, clamp="relu", white_back=True, last_back=False, seed=0, full=False,
                           grads=False),
    # cfg-2 numerics gate (64x64x24, B=1)
    "short_fg_64x24": dict(variant="SHORTSIREN_FG", B=1, R=64, S=24, V=24, C=32, H=256, Z=256,
                           noise=0.0, clamp="relu", white_back=True, last_back=False, seed=1, full=False,
                           grads=False),
    # small-width variants with every toggle, intermediates and grads
    "short_fg_small": dict(variant="SHORTSIREN_FG", B=2, R=16, S=12, V=12, C=32, H=64, Z=48,
                           noise=0.5, clamp="relu", white_back=True, last_back=False, seed=2, full=True,
                           grads=True),
    "tall_fg_small": dict(variant="TALLSIREN_FG", B=1, R=16, S=8, V=10, C=32, H=64, Z=48,
                          noise=0.0, clamp="softplus", white_back=False, last_back=True, seed=3, full=True,
                          grads=True),
    "double_fg_small": dict(variant="DOUBLESIREN_FG", B=2, R=12, S=16, V=8, C=32, H=64, Z=32,
                            noise=0.5, clamp="softplus", white_back=True, last_back=True, seed=4, full=True,
                            grads=True),
    "single_dg_small": dict(variant="SingleSIREN_dg", B=1, R=16, S=12, V=12, C=32, H=64, Z=48,
                            noise=0.0, clamp="relu", white_back=False, last_back=False, seed=5, full=True,
                            grads=True),
    "short_f_small": dict(variant="SHORTSIREN_F", B=1, R=16, S=12, V=12, C=32, H=64, Z=32,
                          noise=0.0, clamp="relu", white_back=True, last_back=False, seed=6, full=True,
                          grads=True),
    "short_fres_small": dict(variant="SHORTSIREN_FRes", B=1, R=16, S=12, V=12, C=32, H=64, Z=32,
                             noise=0.0, clamp="relu", white_back=True, last_back=False, seed=7, full=True,
                             grads=True),
    "tall_dres_small": dict(variant="TALLSIREN_dRes", B=1, R=16, S=12, V=12, C=32, H=64, Z=32,
                            noise=0.0, clamp="relu", white_back=True, last_back=False, seed=8, full=True,
                            grads=True),
    # non-hierarchical single pass
    "short_fg_nohier": dict(variant="SHORTSIREN_FG", B=1, R=16, S=12, V=12, C=32, H=64, Z=48,
                            noise=0.0, clamp="relu", white_back=True, last_back=False, seed=9, full=True,
                            grads=True, hierarchical=False),
    # S > 32 (several MFMA point tiles per ray, multi-chunk scans)
    "short_fg_s40": dict(variant="SHORTSIREN_FG", B=1, R=8, S=40, V=12, C=32, H=64, Z=48,
                         noise=0.0, clamp="relu", white_back=True, last_back=False, seed=10, full=True,
                         grads=True),
    # feature || xyz input (K0 = 35)
    "tall_dgx_small": dict(variant="TALLSIREN_dgx", B=1, R=16, S=8, V=10, C=32, H=64, Z=48,
                           noise=0.0, clamp="relu", white_back=True, last_back=False, seed=11, full=True,
                           grads=True, input_dim=35),
    # feature pyramid: three volumes of different resolution / width concatenated (K0 = 128)
    "short_pyrmd_small": dict(variant="SHORTSIREN_FG_Pyrmd", B=2, R=12, S=12, V=12, C=32, H=64, Z=48,
                              noise=0.0, clamp="relu", white_back=True, last_back=False, seed=12, full=True,
                              grads=True, input_dim=128, pyramid=[(32, 12), (64, 6), (32, 3)]),
    # per-point FiLM from the looked-up feature, xyz input (pi-GAN style TALLSIREN)
    "tallsiren_small": dict(variant="TALLSIREN", B=1, R=12, S=8, V=10, C=32, H=64, Z=32,
                            noise=0.0, clamp="relu", white_back=True, last_back=False, seed=13, full=True,
                            grads=True),
    # the deepest residual chain: sine + 4 residual blocks + sine (siren.py:411-488)
    "tall_dreslong_small": dict(variant="TALLSIREN_dResLong", B=1, R=12, S=12, V=10, C=32, H=64, Z=32,
                                noise=0.0, clamp="relu", white_back=True, last_back=False, seed=14, full=True,
                                grads=True),
    # A WELL-CONDITIONED scene for free-running parity (no teacher forcing): smooth feature volume (a 3^3 grid upsampled
    # trilinearly), mild head, softplus density -> every importance bin carries mass, the field varies slowly along a ray, and
    # the reference's own fp32 result sits ~1e-6 from exact arithmetic (the other fixtures: 1e-3 .. 1e-1, see
    # tests/test_gpu_parity.py::test_free_running_noise_floor).
    "short_fg_smooth": dict(variant="SHORTSIREN_FG", B=2, R=16, S=12, V=16, C=32, H=64, Z=48,
                            noise=0.0, clamp="softplus", white_back=True, last_back=False, seed=22, full=True,
                            grads=False, head=(3.0, 5.0, 0.5), smooth_from=3, amp=0.3),
    # training mode with dropout behind every sine (siren.py:158-159,175-176,197-198): the keep decisions F.dropout drew are
    # recorded (drop_coarse / drop_fine) like the other random draws
    "short_fg_drop_small": dict(variant="SHORTSIREN_FG", B=2, R=12, S=10, V=10, C=32, H=64, Z=48,
                                noise=0.5, clamp="relu", white_back=True, last_back=False, seed=23, full=True,
                                grads=True, drop_out=0.25, head=(6.0, 20.0, 0.25)),   # (kept activations are scaled by 4/3 per layer)
    "tallsiren_drop_small": dict(variant="TALLSIREN", B=1, R=10, S=8, V=8, C=32, H=64, Z=32,
                                 noise=0.0, clamp="softplus", white_back=True, last_back=False, seed=24, full=True,
                                 grads=True, drop_out=0.1),
    "short_f_drop_small": dict(variant="SHORTSIREN_F", B=1, R=10, S=12, V=8, C=32, H=128, Z=32,
                               noise=0.0, clamp="relu", white_back=False, last_back=True, seed=25, full=True,
                               grads=True, drop_out=0.5),
}

# variants whose `z` is the bare feature volume (no global feature)
NO_GLOBAL = {"SHORTSIREN_F", "SHORTSIREN_FRes", "TALLSIREN_dRes", "TALLSIREN_dResLong", "TALLSIREN"}


def build(name):
    spec = FIXTURES[name]
    sys.path.insert(0, REF)
    from generators import generators as ref_gen          # noqa: E402
    from generators import volumetric_rendering as ref_vr  # noqa: E402

    torch.manual_seed(spec["seed"])
    np.random.seed(spec["seed"])
    B, R, S, V, C, H, Z = (spec[k] for k in "BRSVCHZ")
    hier = spec.get("hierarchical", True)

    variant = spec["variant"]
    # FG family: FiLMLayer(input_dim, hidden) eats the looked-up feature -> input_dim = C,
    # z_dim = width of the global feature.  Plain-sine families set input_dim = z_dim themselves.
    p_drop = spec.get("drop_out", 0)
    if variant == "TALLSIREN":      # input = xyz; z_dim = width of the looked-up feature feeding the per-point mapping net
        gen = ref_gen.ImplicitGenerator3d(variant, z_dim=C, input_dim=3, output_dim=4, hidden_dim=H, drop_out=p_drop)
    elif variant in NO_GLOBAL:
        gen = ref_gen.ImplicitGenerator3d(variant, z_dim=C, input_dim=C, output_dim=4, hidden_dim=H, drop_out=p_drop)
    else:
        gen = ref_gen.ImplicitGenerator3d(variant, z_dim=Z, input_dim=spec.get("input_dim", C), output_dim=4, hidden_dim=H,
                                          drop_out=p_drop)
    gen.set_device(torch.device("cpu"))
    gen.train(bool(p_drop))         # dropout is active in training mode only
    # Default init gives near-zero densities (an all-background image pins nothing): scale the head so
    # that sigma spans both signs at O(1..10) and colours leave the sigmoid's linear range.  The scaled
    # values are stored in the fixture like every other parameter.
    head = spec.get("head", (6.0, 40.0, 0.25))
    with torch.no_grad():
        gen.siren.final_layer.weight[:3] *= head[0]
        gen.siren.final_layer.weight[3] *= head[1]
        gen.siren.final_layer.bias[3] += head[2]

    if "smooth_from" in spec:
        low = torch.randn(B, C, spec["smooth_from"], spec["smooth_from"], spec["smooth_from"]) * spec["amp"]
        fvol = F.interpolate(low, size=(V, V, V), mode="trilinear", align_corners=True).requires_grad_(True)
    else:
        fvol = (torch.randn(B, C, V, V, V) * 0.5).requires_grad_(True)
    pyr = None
    if "pyramid" in spec:
        pyr = [fvol] + [(torch.randn(B, c, v, v, v) * 0.5).requires_grad_(True) for c, v in spec["pyramid"][1:]]
    glob = torch.randn(B, Z).requires_grad_(True)
    origins = ref_vr.sample_camera_positions(torch.device("cpu"), "y", cam_r_start=0.7, cam_r_end=1.5, n=B)
    cam2world = ref_vr.create_cam2world_matrix(origins, "y", device=torch.device("cpu")).float()

    rec = {"rand": [], "randn": [], "feat": [], "siren_in": [], "siren_out": [], "weights": [],
           "z_in": [], "cdf": [], "inds": [], "fine_z": [], "sort_idx": [], "drop": []}

    o_rand, o_randn, o_gs, o_ss, o_sort = torch.rand, torch.randn, F.grid_sample, torch.searchsorted, torch.sort
    o_fi, o_sp = ref_gen.fancy_integration, ref_gen.sample_pdf
    o_drop = F.dropout

    def w_drop(x, p=0.5, training=True, inplace=False):
        # nn.Dropout.forward -> F.dropout: run the reference's call and read the keep decisions off its output
        # (y = x * keep / (1 - p); sin(.) == 0 exactly does not occur in these fixtures, asserted below)
        y = o_drop(x, p, training, inplace)
        if training and p > 0:
            assert (x != 0).all()
            rec["drop"].append((y != 0).detach().clone())
        return y

    def w_rand(*a, **k):
        t = o_rand(*a, **k); rec["rand"].append(t.detach().clone()); return t

    def w_randn(*a, **k):
        t = o_randn(*a, **k); rec["randn"].append(t.detach().clone()); return t

    def w_gs(*a, **k):
        t = o_gs(*a, **k); rec["feat"].append(t.detach().clone()); return t

    def w_ss(cdf, u, **k):
        t = o_ss(cdf, u, **k); rec["cdf"].append(cdf.detach().clone()); rec["inds"].append(t.clone()); return t

    def w_sort(*a, **k):
        t = o_sort(*a, **k); rec["sort_idx"].append(t[1].clone()); return t

    def w_fi(rgb_sigma, z_vals, **k):
        out = o_fi(rgb_sigma, z_vals, **k)
        rec["weights"].append(out[2].detach().clone()); rec["z_in"].append(z_vals.detach().clone())
        return out

    def w_sp(*a, **k):
        t = o_sp(*a, **k); rec["fine_z"].append(t.detach().clone()); return t

    def siren_hook(_m, inp, outp):
        rec["siren_in"].append(inp[0].detach().clone())
        rec["siren_out"].append(outp.detach().clone())

    hook = gen.siren.register_forward_hook(siren_hook)
    torch.rand, torch.randn, F.grid_sample, torch.searchsorted, torch.sort = w_rand, w_randn, w_gs, w_ss, w_sort
    ref_gen.fancy_integration, ref_gen.sample_pdf = w_fi, w_sp
    F.dropout = w_drop
    try:
        z = fvol if variant in NO_GLOBAL else ((pyr if pyr is not None else fvol), glob)
        kw = dict(clamp_mode=spec["clamp"], nerf_noise=spec["noise"], white_back=spec["white_back"],
                  last_back=spec["last_back"],
                  # the caller splats its whole metadata dict: extra keys must be ignored
                  batch_size=B, generator={"siren_type": variant})
        pixels, depth = gen(z, cam2world, R, FOV, RAY_START, RAY_END, S, hier, **kw)
    finally:
        torch.rand, torch.randn, F.grid_sample, torch.searchsorted, torch.sort = o_rand, o_randn, o_gs, o_ss, o_sort
        ref_gen.fancy_integration, ref_gen.sample_pdf = o_fi, o_sp
        F.dropout = o_drop
        hook.remove()

    out = {}
    meta = dict(spec)
    meta.update(fov=FOV, ray_start=RAY_START, ray_end=RAY_END, torch=torch.__version__, name=name,
                hierarchical=hier, has_global=variant not in NO_GLOBAL)
    out["meta_json"] = np.frombuffer(__import__("json").dumps(meta).encode(), dtype=np.uint8)
    out["feature_volume"] = fvol.detach().numpy()
    if pyr is not None:
        for i, v in enumerate(pyr[1:], 1):
            out[f"feature_volume_l{i}"] = v.detach().numpy()
    if variant not in NO_GLOBAL:
        out["global_feature"] = glob.detach().numpy()
    out["cam2worlds"] = cam2world.numpy()
    for k, v in gen.state_dict().items():
        out["param/" + k] = v.numpy()

    # RNG tensors in draw order: rand(B,R2,S,1) randn(B,R2,S,1) rand(BR2,S) randn(B,R2,2S,1)
    out["u_strat"] = rec["rand"][0].numpy().reshape(B, R * R, S)
    if hier:
        assert len(rec["rand"]) == 2 and len(rec["randn"]) == 2
        out["u_fine"] = rec["rand"][1].numpy().reshape(B, R * R, S)
        if spec["noise"] != 0:
            out["eps_coarse"] = rec["randn"][0].numpy().reshape(B, R * R, S)
            out["eps_final"] = rec["randn"][1].numpy().reshape(B, R * R, 2 * S)
    else:
        assert len(rec["rand"]) == 1 and len(rec["randn"]) == 1
        if spec["noise"] != 0:
            out["eps_final"] = rec["randn"][0].numpy().reshape(B, R * R, S)

    if p_drop:
        n_drop = len(rec["drop"]) // (2 if hier else 1)      # one F.dropout call per FiLM / sine layer and forward call
        out["drop_coarse"] = np.stack([m.numpy().reshape(B, R * R * S, H) for m in rec["drop"][:n_drop]]).astype(np.uint8)
        if hier:
            out["drop_fine"] = np.stack([m.numpy().reshape(B, R * R * S, H) for m in rec["drop"][n_drop:]]).astype(np.uint8)
    out["pixels"] = pixels.detach().numpy()
    out["depth"] = depth.detach().numpy()
    out["coarse_rgb_sigma"] = rec["siren_out"][0].numpy().reshape(B, R * R, S, 4)
    out["coarse_z"] = rec["z_in"][0].numpy().reshape(B, R * R, S)
    if hier:
        out["fine_rgb_sigma"] = rec["siren_out"][1].numpy().reshape(B, R * R, S, 4)
        out["fine_z"] = rec["fine_z"][0].numpy().reshape(B, R * R, S)
        out["inds"] = rec["inds"][0].numpy().reshape(B, R * R, S).astype(np.int16)
        out["sort_idx"] = rec["sort_idx"][0].numpy().reshape(B, R * R, 2 * S).astype(np.int16)
        out["coarse_weights"] = rec["weights"][0].numpy().reshape(B, R * R, S)
    if spec["full"]:
        out["coarse_points"] = rec["siren_in"][0].numpy().reshape(B, R * R, S, 3)
        nlev = len(pyr) if pyr is not None else 1     # grid_sample runs once per pyramid level and pass
        out["coarse_feat"] = np.concatenate([rec["feat"][i].numpy().reshape(B, -1, R * R * S).transpose(0, 2, 1)
                                             for i in range(nlev)], -1).copy()
        out["final_weights"] = rec["weights"][-1].numpy().reshape(B, R * R, -1)
        if hier:
            out["cdf"] = rec["cdf"][0].numpy().reshape(B, R * R, S - 1)
            out["fine_points"] = rec["siren_in"][1].numpy().reshape(B, R * R, S, 3)

    if spec["grads"]:
        loss = pixels.square().mean() + depth.mean()
        params = dict(gen.named_parameters())
        extra = pyr[1:] if pyr is not None else []
        leaves = list(params.values()) + [fvol] + ([] if variant in NO_GLOBAL else [glob]) + extra
        grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        for (k, _), g in zip(params.items(), grads):
            out["grad/" + k] = (g if g is not None else torch.zeros_like(params[k])).numpy()
        out["grad_feature_volume"] = grads[len(params)].numpy()
        nxt = len(params) + 1
        if variant not in NO_GLOBAL:
            out["grad_global_feature"] = grads[nxt].numpy()
            nxt += 1
        for i, g in enumerate(grads[nxt:], 1):
            out[f"grad_feature_volume_l{i}"] = g.numpy()
        out["loss"] = np.float32(loss.item())

    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)/1e6:.2f} MB  pixels[{pixels.min():.3f},{pixels.max():.3f}] "
          f"depth[{depth.min():.3f},{depth.max():.3f}]")


def build_unet3d():
    """aux_unet3d_small.npz: the reference's UNet3D (generators/unet3d.py) on a tiny voxel grid, for the plain-PyTorch
    encoder restatement of the training harness (conditioned-nerf-gan_amd/training/encoder.py)."""
    sys.path.insert(0, REF)
    from generators import unet3d
    torch.manual_seed(0)
    net = unet3d.UNet3D(in_channels=4, out_channels=16, f_maps=8, num_levels=3, final_sigmoid=False, is_segmentation=False,
                        return_global=True)
    net.eval()
    vox = torch.rand(1, 4, 8, 8, 8)
    with torch.no_grad():
        fv, g = net(vox)
    out = {"voxel": vox.numpy(), "feature_volume": fv.numpy(), "global_feature": g.numpy()}
    for k, v in net.state_dict().items():
        out["param/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "aux_unet3d_small.npz"), **out)


def build_coordconv():
    """aux_coordconv.npz: CoordConv and AdapterBlock as discriminators/sgdiscriminators.py defines them (the same source text
    as in discriminators/discriminators.py, which cannot be imported here: it pulls in tkinter).  Pins the coordinate
    channels (order, orientation, range) and the parameter names of the discriminator restatement's building blocks; the
    assembly of ProgressiveDiscriminator itself (block list, fade-in) stays restated from source only."""
    sys.path.insert(0, REF)
    from discriminators import sgdiscriminators as sgd
    torch.manual_seed(0)
    cc = sgd.CoordConv(5, 7, kernel_size=3, padding=1)
    ad = sgd.AdapterBlock(6)
    x = torch.randn(2, 5, 6, 9)                  # non-square on purpose: rows and columns must not be swapped
    img = torch.randn(2, 3, 4, 4)
    with torch.no_grad():
        y, a = cc(x), ad(img)
    out = {"x": x.numpy(), "y": y.numpy(), "img": img.numpy(), "adapter_out": a.numpy()}
    for k, v in cc.state_dict().items():
        out["coordconv/" + k] = v.numpy()
    for k, v in ad.state_dict().items():
        out["adapter/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "aux_coordconv.npz"), **out)


def _param_stats(module):
    """Per parameter: (sum, sum of absolute values, first three values) in float64 -- enough to tell that two modules built
    under the same seed hold the same 12.9 M numbers without storing them."""
    out = {}
    for k, v in module.state_dict().items():
        d = v.detach().double().flatten()
        out[k] = np.array([d.sum().item(), d.abs().sum().item(), *d[:3].tolist(), *([0.0] * max(0, 3 - d.numel()))][:5])
    return out


def build_ccs():
    """aux_ccs_discriminator.npz: discriminators/sgdiscriminators.py::CCSDiscriminator (importable; the progressive CoordConv
    discriminator BASELINE.json calls "sgdiscriminator") under torch.manual_seed(0): parameter statistics and its outputs at
    32 / 64 / 128 px for alpha 0, 0.5, 1 -- pins entry resolution, block list and fade-in of the restatement."""
    sys.path.insert(0, REF)
    from discriminators import sgdiscriminators as sgd
    torch.manual_seed(0)
    d = sgd.CCSDiscriminator()
    d.eval()
    out = {}
    for k, v in _param_stats(d).items():
        out["stat/" + k] = v
    g = torch.Generator().manual_seed(1)
    for res in (32, 64, 128):
        x = torch.randn(1, 3, res, res, generator=g)
        out[f"img_{res}"] = x.numpy()
        for alpha in (0.0, 0.5, 1.0):
            with torch.no_grad():
                out[f"pred_{res}_{alpha}"] = d(x, alpha)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "aux_ccs_discriminator.npz"), **out)
    print("aux_ccs_discriminator:", os.path.getsize(os.path.join(HERE, "aux_ccs_discriminator.npz")) / 1e6, "MB")


def build_gan_step():
    """aux_gan_step.npz: one D step and one G step of the GAN loop on the CPU with the reference's OWN modules -- generator
    (generators.ImplicitGenerator3d SHORTSIREN_FG, hidden 64), voxel encoder (unet3d.UNet3D f_maps 8, 3 levels) and discriminator
    (sgdiscriminators.CCSDiscriminator) -- fp32, 2 images of 16x16 rays x (8 + 8) samples.  The step logic (losses softplus(+-pred),
    R1 penalty on the real images, gradient clipping, Adam(0, 0.9)) follows utils.py:621-842, which cannot be imported
    (SURVEY.md F4), so it is spelled out here; everything it calls is the reference.  Stored: inputs, the generator's and the
    encoder's parameters, the random draws of both renders, loss terms, pre-clip gradient norms, post-step parameter sums."""
    sys.path.insert(0, REF)
    from generators import generators as ref_gen, unet3d
    from generators import volumetric_rendering as ref_vr
    from discriminators import sgdiscriminators as sgd
    R, S, B, V, H = 16, 8, 2, 8, 64
    torch.manual_seed(0)
    gen = ref_gen.ImplicitGenerator3d("SHORTSIREN_FG", z_dim=32, input_dim=32, output_dim=4, hidden_dim=H)
    gen.set_device(torch.device("cpu"))
    with torch.no_grad():                       # a head that yields visible structure (default init renders the background)
        gen.siren.final_layer.weight[3] *= 20.0
        gen.siren.final_layer.bias[3] += 0.5
    enc = unet3d.UNet3D(in_channels=4, out_channels=32, f_maps=8, num_levels=3, final_sigmoid=False, is_segmentation=False,
                        return_global=True)
    torch.manual_seed(123)
    disc = sgd.CCSDiscriminator()
    gen.train(); enc.train(); disc.train()
    gen.step = 1000
    alpha, nerf_noise = min(1.0, 1000 / 2000), max(0.0, 1.0 - 1000 / 5000.0)          # utils.py:610-618
    md = dict(img_size=R, num_steps=S, fov=FOV, ray_start=RAY_START, ray_end=RAY_END, white_back=True, last_back=False,
              clamp_mode="relu", hierarchical_sample=True, nerf_noise=nerf_noise)
    g = torch.Generator().manual_seed(5)
    occ = (torch.rand(B, 1, V, V, V, generator=g) > 0.7).float()
    vox = torch.cat([occ, torch.rand(B, 3, V, V, V, generator=g) * occ], 1)
    imgs = torch.rand(B, 3, R, R, generator=g) * 2 - 1
    np.random.seed(6)
    cams_g = ref_vr.create_cam2world_matrix(ref_vr.sample_camera_positions(torch.device("cpu"), "y", 0.7, 1.5, B), "y", device=torch.device("cpu")).float()
    out = {"voxel": vox.numpy(), "img": imgs.numpy(), "cam2world": cams_g.numpy(), "alpha": np.float32(alpha), "nerf_noise": np.float32(nerf_noise)}
    for k, v in gen.state_dict().items():
        out["gen/" + k] = v.numpy().copy()
    for k, v in enc.state_dict().items():
        out["enc/" + k] = v.numpy().copy()
    for k, v in _param_stats(disc).items():
        out["dstat/" + k] = v
    opt = lambda m, lr: torch.optim.Adam(m.parameters(), lr=lr, betas=(0.0, 0.9), weight_decay=0)
    opt_g, opt_e, opt_d = opt(gen, 5e-5), opt(enc, 5e-5), opt(disc, 2e-4)

    o_rand, o_randn = torch.rand, torch.randn
    draws = []

    def capture():
        rec = {"rand": [], "randn": []}
        torch.rand = lambda *a, **k: rec["rand"].append(o_rand(*a, **k)) or rec["rand"][-1]
        torch.randn = lambda *a, **k: rec["randn"].append(o_randn(*a, **k)) or rec["randn"][-1]
        return rec

    def release(rec, tag):
        torch.rand, torch.randn = o_rand, o_randn
        P = R * R
        out[tag + "/u_strat"] = rec["rand"][0].numpy().reshape(B, P, S)
        out[tag + "/u_fine"] = rec["rand"][1].numpy().reshape(B, P, S)
        out[tag + "/eps_coarse"] = rec["randn"][0].numpy().reshape(B, P, S)
        out[tag + "/eps_final"] = rec["randn"][1].numpy().reshape(B, P, 2 * S)

    # ---- discriminator step (utils.py:743-842) ----
    np.random.seed(7)
    cams_d = ref_vr.create_cam2world_matrix(ref_vr.sample_camera_positions(torch.device("cpu"), "y", 0.7, 1.5, B), "y", device=torch.device("cpu")).float()
    out["cam2world_d"] = cams_d.numpy()
    with torch.no_grad():
        rec = capture()
        try:
            fake, _ = gen(enc(vox), cams_d, **md)
        finally:
            release(rec, "d")
    real = imgs.clone().requires_grad_(True)
    r_preds = disc(real, alpha)[0]
    (grad_real,) = torch.autograd.grad(r_preds.sum(), real, create_graph=True)
    penalty = 0.5 * 10 * (grad_real.view(B, -1).norm(2, dim=1) ** 2).mean()
    g_preds = disc(fake, alpha)[0]
    d_loss = F.softplus(g_preds).mean() + F.softplus(-r_preds).mean() + penalty
    opt_d.zero_grad()
    d_loss.backward()
    d_norm = torch.nn.utils.clip_grad_norm_(disc.parameters(), 1)
    opt_d.step()
    out.update(d_loss=np.float64(d_loss.item()), r1_penalty=np.float64(penalty.item()), d_grad_norm=np.float64(float(d_norm)),
               fake_mean=np.float64(fake.mean().item()), fake=fake.numpy())
    # ---- generator step (utils.py:621-741), batch_split 1 ----
    rec = capture()
    try:
        gen_imgs, _ = gen(enc(vox), cams_g, **md)
    finally:
        release(rec, "g")
    loss_g = F.softplus(-disc(gen_imgs, alpha)[0]).mean()
    photo = ((imgs - gen_imgs) ** 2).mean()
    (loss_g + photo).backward()
    g_norm = torch.nn.utils.clip_grad_norm_(gen.parameters(), 1)
    opt_g.step()
    e_norm = torch.nn.utils.clip_grad_norm_(enc.parameters(), 1)
    opt_e.step()
    out.update(g_loss=np.float64(loss_g.item()), photo_loss=np.float64(photo.item()), g_grad_norm=np.float64(float(g_norm)),
               e_grad_norm=np.float64(float(e_norm)), gen_imgs=gen_imgs.detach().numpy())
    for k, v in gen.state_dict().items():
        out["gen_after_sum/" + k] = np.float64(v.double().sum().item())
    out["enc_after_sum/final_conv.weight"] = np.float64(enc.final_conv.weight.double().sum().item())
    np.savez_compressed(os.path.join(HERE, "aux_gan_step.npz"), **out)
    print("aux_gan_step:", os.path.getsize(os.path.join(HERE, "aux_gan_step.npz")) / 1e6, "MB;",
          {k: float(out[k]) for k in ("d_loss", "r1_penalty", "d_grad_norm", "g_loss", "photo_loss", "g_grad_norm", "e_grad_norm", "fake_mean")})


if __name__ == "__main__":
    names = sys.argv[1:] or list(FIXTURES)
    for n in names:
        if n == "unet3d":
            build_unet3d()
        elif n == "coordconv":
            build_coordconv()
        elif n == "ccs":
            build_ccs()
        elif n == "gan_step":
            build_gan_step()
        else:
            build(n)
