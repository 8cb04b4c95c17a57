#!/usr/bin/env python3
"""HIP gradients vs autograd through the CPU oracle (fp32, same samples: the oracle's fine depths are forced).
usage: grad_vs_oracle.py variant B R S V H noise clamp [white] [last]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import scaled_err
from oracle import render_oracle as O
import cnerf_amd
from cnerf_amd.generators import ImplicitGenerator3d
from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
def run(variant, B, R, S, V, H, noise, clamp, white=True, last=False, seed=0):
    torch.manual_seed(seed); np.random.seed(seed)
    Z = 32
    has_glob = O.FIELD_SPECS[variant].has_global
    gen = ImplicitGenerator3d(variant, Z if has_glob else 32, 32, 4, H)
    with torch.no_grad():
        gen.siren.final_layer.weight[:3] *= 6; gen.siren.final_layer.weight[3] *= 40; gen.siren.final_layer.bias[3] += 0.25
    fvol = (torch.randn(B, 32, V, V, V) * 0.5).requires_grad_(True); glob = torch.randn(B, Z).requires_grad_(True) if has_glob else None
    cam = create_cam2world_matrix(sample_camera_positions("cpu", "y", 0.7, 1.5, B), "y")
    P = R * R
    rng = {"u_strat": torch.rand(B, P, S), "eps_coarse": torch.randn(B, P, S), "u_fine": torch.rand(B, P, S), "eps_final": torch.randn(B, P, 2 * S)}
    params = {k: v.detach().clone().requires_grad_(True) for k, v in gen.siren.state_dict().items()}
    ref = O.render(variant, params, fvol, glob, cam, R, 49.13, 0.25, 1.95, S, True, clamp, noise, white, last,
                   rng["u_strat"], rng["eps_coarse"], rng["u_fine"], rng["eps_final"])
    loss = ref.pixels.square().mean() + ref.depth.mean()
    leaves = [fvol] + ([glob] if has_glob else []) + list(params.values())
    gref = torch.autograd.grad(loss, leaves)
    dev = torch.device("cuda:0")
    gen.to(dev); gen.set_device(dev)
    fv = fvol.detach().to(dev).requires_grad_(True); gl = glob.detach().to(dev).requires_grad_(True) if has_glob else None
    r = {k: v.to(dev) for k, v in rng.items()}; r["fine_z"] = ref.aux["fine_z"].detach().to(dev)
    px, dp = gen((fv, gl) if has_glob else fv, cam.to(dev), R, 49.13, 0.25, 1.95, S, True, clamp_mode=clamp, nerf_noise=noise, white_back=white, last_back=last, _rng=r)
    l2 = px.square().mean() + dp.mean(); l2.backward()
    out = {"loss": abs(l2.item() - loss.item()), "fvol": scaled_err(fv.grad.cpu().numpy(), gref[0].numpy())}
    i = 1
    if has_glob:
        out["glob"] = scaled_err(gl.grad.cpu().numpy(), gref[1].numpy()); i = 2
    worst = 0
    for (k, _), gg in zip(params.items(), gref[i:]):
        p = dict(gen.siren.named_parameters())[k]
        e = scaled_err(p.grad.cpu().numpy(), gg.numpy()); worst = max(worst, e)
    out["params_worst"] = worst
    print(f"{variant} B={B} R={R} S={S} V={V} H={H} noise={noise} {clamp} white={white} last={last}:", {k: f"{v:.2e}" for k, v in out.items()}, flush=True)
if __name__ == "__main__":
    if len(sys.argv) > 1:
        a = sys.argv[1:]
        run(a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), float(a[6]), a[7])
    else:
        run("SHORTSIREN_FG", 1, 16, 12, 12, 64, 0.0, "relu")
        run("SHORTSIREN_FG", 2, 16, 12, 12, 64, 0.0, "relu")
        run("SHORTSIREN_FG", 1, 16, 12, 12, 64, 0.5, "relu")
        run("SHORTSIREN_FG", 2, 16, 12, 12, 64, 0.5, "relu")
        run("SHORTSIREN_FG", 2, 16, 12, 12, 64, 0.5, "softplus")
        run("DOUBLESIREN_FG", 2, 16, 12, 12, 64, 0.5, "relu")
        run("SHORTSIREN_FG", 2, 16, 12, 12, 64, 0.5, "relu", seed=1)
        run("SHORTSIREN_FG", 2, 16, 12, 12, 64, 0.5, "relu", seed=2)
