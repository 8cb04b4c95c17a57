#!/usr/bin/env python3
"""Per-tensor gradient errors of the HIP backward against the golden fixtures (the reference's CPU autograd, fine depths
teacher-forced), for the exact fp32 backward and the half-precision one.  Markdown on stdout.
Columns: scaled_err = max|a-b| / max(|b|, rms(b)) (the tests' metric) and rel-L2 = ||a-b|| / ||b||; `floor` = the
reference's own fp32-vs-float64 scaled_err on that tensor (tests/test_gpu_parity.py::reference_grad_noise_floor)."""
import os, sys, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import Golden, scaled_err
from test_gpu_parity import make_generator, make_z, G, GRAD_FIXTURES, HALF_BACKWARD_FIXTURES, reference_grad_noise_floor
dev = torch.device("cuda:0")
names = sys.argv[1:] or GRAD_FIXTURES
l2 = lambda a, b: float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))
print("| fixture | tensor | floor | fp32 backward scaled_err | rel-L2 | fp16 backward scaled_err | rel-L2 |")
print("|---|---|---|---|---|---|---|")
for name in names:
    g = Golden(name); m = g.meta
    floor = reference_grad_noise_floor(g)
    res = {}
    for bp in ("fp32", "fp16"):
        if bp == "fp16" and name not in HALF_BACKWARD_FIXTURES:
            continue
        gen = make_generator(g, dev); gen.train()
        gen.siren.precision = "fp32" if bp == "fp32" else "fp16x3"
        gen.siren.backward_precision = bp
        z, vleaves, glob = make_z(g, dev, requires_grad=True)
        rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
        if m["hierarchical"]: rng["fine_z"] = G(g["fine_z"], dev)
        px, dp = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"], clamp_mode=m["clamp"],
                     nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=rng)
        (px.square().mean() + dp.mean()).backward()
        got = {}
        for li, leaf in enumerate(vleaves):
            sfx = f"_l{li}" if li else ""
            got["feature_volume" + sfx] = (leaf.grad.cpu().numpy(), g["grad_feature_volume" + sfx])
        if glob is not None:
            got["global_feature"] = (glob.grad.cpu().numpy(), g["grad_global_feature"])
        for k, p in gen.named_parameters():
            got[k] = (p.grad.cpu().numpy(), g["grad/" + k])
        res[bp] = {k: (scaled_err(a, b), l2(a, b)) for k, (a, b) in got.items()}
    for k in res["fp32"]:
        h = res.get("fp16", {}).get(k)
        print(f"| {name} | {k} | {floor[k]:.1e} | {res['fp32'][k][0]:.1e} | {res['fp32'][k][1]:.1e} | " + (f"{h[0]:.1e} | {h[1]:.1e} |" if h else "- | - |"))
