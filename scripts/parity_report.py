#!/usr/bin/env python3
"""Parity table: HIP render path vs the golden vectors of the reference, per fixture and precision.
Metric: max|a-b| / max(|b|, rms(b)).  Fine depths are teacher-forced.  Markdown on stdout."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import Golden, GOLDEN_NAMES, scaled_err
from test_gpu_parity import make_generator, make_z, G, SPLIT_FIXTURES
dev = torch.device("cuda:0")
print("| fixture | variant | shape BxRxRxS | precision | points | z | feat | rgb coarse | sigma coarse | rgb fine | sigma fine | sort_idx | inds equal (free-running) | pixels | depth |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name in GOLDEN_NAMES:
    g = Golden(name); m = g.meta
    for prec in (["fp32", "fp16x3"] if name in SPLIT_FIXTURES else ["fp32"]):
        gen = make_generator(g, dev); gen.siren.precision = prec
        z, _, _ = make_z(g, dev)
        rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
        free = {}
        if m["hierarchical"]:
            with torch.no_grad():
                gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], True, clamp_mode=m["clamp"],
                    nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=rng, _aux=free)
            rng = dict(rng); rng["fine_z"] = G(g["fine_z"], dev)
        aux = {}
        with torch.no_grad():
            px, dp = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"],
                         clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=rng, _aux=aux)
        a = {k: v.cpu().numpy() for k, v in aux.items()}
        e = lambda x, y: f"{scaled_err(x, y):.1e}"
        bit = lambda x, y: "bit-exact" if np.array_equal(x, y) else f"{(x != y).mean():.1e} differ"
        row = [name, m["variant"], f"{m['B']}x{m['R']}x{m['R']}x{m['S']}", prec,
               bit(a["coarse_points"], g["coarse_points"]) if "coarse_points" in g else "-", bit(a["coarse_z"], g["coarse_z"]), "-",
               e(a["coarse_rgb_sigma"][..., :3], g["coarse_rgb_sigma"][..., :3]), e(a["coarse_rgb_sigma"][..., 3], g["coarse_rgb_sigma"][..., 3])]
        if m["hierarchical"]:
            row += [e(a["fine_rgb_sigma"][..., :3], g["fine_rgb_sigma"][..., :3]), e(a["fine_rgb_sigma"][..., 3], g["fine_rgb_sigma"][..., 3]),
                    bit(a["sort_idx"], g["sort_idx"].astype(np.int32)), f"{(free['inds'].cpu().numpy() == g['inds']).mean()*100:.3f} %"]
        else:
            row += ["-", "-", "-", "-"]
        row += [e(px.cpu().numpy(), g["pixels"]), e(dp.cpu().numpy(), g["depth"])]
        print("| " + " | ".join(row) + " |", flush=True)
