#!/usr/bin/env python3
"""Per-tensor gradient errors of the HIP backward against the golden fixtures (teacher-forced fine depths)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import Golden, scaled_err
from test_gpu_parity import make_generator, G
dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    g = Golden(name); m = g.meta
    gen = make_generator(g, dev); gen.train()
    fvol = G(g["feature_volume"], dev).requires_grad_(True)
    glob = G(g["global_feature"], dev).requires_grad_(True) if m["has_global"] else None
    z = (fvol, glob) if m["has_global"] else fvol
    rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
    if m["hierarchical"]: rng["fine_z"] = G(g["fine_z"], dev)
    aux = {}
    pixels, depth = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"],
                        clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=rng, _aux=aux)
    loss = pixels.square().mean() + depth.mean(); loss.backward()
    print(name, "loss", loss.item(), float(g["loss"]), "pix err", scaled_err(pixels.detach().cpu().numpy(), g["pixels"]))
    print("   fvol", scaled_err(fvol.grad.cpu().numpy(), g["grad_feature_volume"]), "max ref", np.abs(g["grad_feature_volume"]).max())
    if glob is not None: print("   glob", scaled_err(glob.grad.cpu().numpy(), g["grad_global_feature"]))
    ref = {k[len("grad/"):]: g[k] for k in g.d.files if k.startswith("grad/")}
    for k, p in gen.named_parameters():
        print("   ", k, scaled_err(p.grad.cpu().numpy(), ref[k]))
    # where is the fvol error located?
    d = np.abs(fvol.grad.cpu().numpy() - g["grad_feature_volume"])
    idx = np.unravel_index(np.argmax(d), d.shape); print("   worst fvol idx", idx, "mine", fvol.grad.cpu().numpy()[idx], "ref", g["grad_feature_volume"][idx])
    print("   frac of voxels with rel err > 1e-2:", float((d > 1e-2 * np.abs(g["grad_feature_volume"]).max()).mean()))
