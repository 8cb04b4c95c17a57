#!/usr/bin/env python3
"""Parity tables: HIP render path vs the golden vectors of the reference, per fixture and precision.  Markdown on stdout.

Table 1 (teacher-forced: the reference's fine depths injected): per tensor the test metric
`scaled_err` = max|a-b| / max(|b|, rms(b)) and, next to it, the fraction of elements inside the gate SURVEY.md 8(d) wrote
down, |a-b| <= 1e-4 * max(|b|, 1e-3) ("8d pass").
Table 2 (free-running: nothing forced): pixel / depth error statistics of HIP-vs-reference beside the reference's own
fp32-vs-float64 statistics on the same fixture (tests/test_gpu_parity.py::free_running_floor)."""
import os, sys, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import Golden, GOLDEN_NAMES, scaled_err
from test_gpu_parity import make_generator, make_z, G, SPLIT_FIXTURES, survey_metric_pass, err_stats, free_running_floor
dev = torch.device("cuda:0")
free_rows = []
print("| fixture | variant | shape BxRxRxS | precision | points | z | rgb coarse | sigma coarse | 8d pass rgb-sigma coarse | rgb fine | sigma fine | sort_idx | inds equal (free-running) | pixels | depth | 8d pass pixels |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name in GOLDEN_NAMES:
    g = Golden(name); m = g.meta
    floor = free_running_floor(g) if m["hierarchical"] else None
    for prec in (["fp32", "fp16x3"] if name in SPLIT_FIXTURES else ["fp32"]):
        gen = make_generator(g, dev); gen.siren.precision = prec
        z, _, _ = make_z(g, dev)
        rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
        free = {}
        if m["hierarchical"]:
            with torch.no_grad():
                fpx, fdp = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], True, clamp_mode=m["clamp"],
                               nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=rng, _aux=free)
            free_rows.append((name, prec, err_stats(fpx.cpu().numpy(), g["pixels"]), err_stats(fdp.cpu().numpy(), g["depth"]), floor))
            rng = dict(rng); rng["fine_z"] = G(g["fine_z"], dev)
        aux = {}
        with torch.no_grad():
            px, dp = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"],
                         clamp_mode=m["clamp"], nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=rng, _aux=aux)
        a = {k: v.cpu().numpy() for k, v in aux.items()}
        e = lambda x, y: f"{scaled_err(x, y):.1e}"
        bit = lambda x, y: "bit-exact" if np.array_equal(x, y) else f"{(x != y).mean():.1e} differ"
        row = [name, m["variant"], f"{m['B']}x{m['R']}x{m['R']}x{m['S']}", prec,
               bit(a["coarse_points"], g["coarse_points"]) if "coarse_points" in g else "-", bit(a["coarse_z"], g["coarse_z"]),
               e(a["coarse_rgb_sigma"][..., :3], g["coarse_rgb_sigma"][..., :3]), e(a["coarse_rgb_sigma"][..., 3], g["coarse_rgb_sigma"][..., 3]),
               f"{survey_metric_pass(a['coarse_rgb_sigma'], g['coarse_rgb_sigma']):.4f}"]
        if m["hierarchical"]:
            row += [e(a["fine_rgb_sigma"][..., :3], g["fine_rgb_sigma"][..., :3]), e(a["fine_rgb_sigma"][..., 3], g["fine_rgb_sigma"][..., 3]),
                    bit(a["sort_idx"], g["sort_idx"].astype(np.int32)), f"{(free['inds'].cpu().numpy() == g['inds']).mean()*100:.3f} %"]
        else:
            row += ["-", "-", "-", "-"]
        row += [e(px.cpu().numpy(), g["pixels"]), e(dp.cpu().numpy(), g["depth"]), f"{survey_metric_pass(px.cpu().numpy(), g['pixels']):.4f}"]
        print("| " + " | ".join(row) + " |", flush=True)

print("\n## Free-running renders (nothing forced): HIP-vs-reference beside the reference's own fp32-vs-float64 distance\n")
print("| fixture | precision | pixels HIP-vs-ref mean / p99 / p99.9 / max | pixels ref fp32-vs-fp64 mean / p99 / p99.9 / max | depth HIP-vs-ref mean / p99 / p99.9 / max | depth ref fp32-vs-fp64 mean / p99 / p99.9 / max |")
print("|---|---|---|---|---|---|")
f3 = lambda s: f"{s['mean']:.1e} / {s['p99']:.1e} / {s['p999']:.1e} / {s['max']:.1e}"
for name, prec, hpx, hdp, (fpx, fdp) in free_rows:
    print(f"| {name} | {prec} | {f3(hpx)} | {f3(fpx)} | {f3(hdp)} | {f3(fdp)} |")
