#!/usr/bin/env python3
"""Diagnostic: the chunk buffers of the fp16 backward (TB16) against those of the fp32 backward on one golden fixture."""
import os, sys, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import Golden
from test_gpu_parity import make_generator, make_z, G
import cnerf_amd
from cnerf_amd import ops
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "double_fg_small"
g = Golden(name); m = g.meta
cap = {}
grads = {}
for bp in ("fp32", "fp16"):
    gen = make_generator(g, dev); gen.train()
    gen.siren.precision = "fp16x3"; gen.siren.backward_precision = bp
    z, vleaves, glob = make_z(g, dev, requires_grad=True)
    rng = {k: G(g.get(k), dev) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
    if m["hierarchical"]: rng["fine_z"] = G(g["fine_z"], dev)
    ops.DEBUG_CAPTURE = cap
    px, dp = gen(z, G(g["cam2worlds"], dev), m["R"], m["fov"], m["ray_start"], m["ray_end"], m["S"], m["hierarchical"], clamp_mode=m["clamp"],
                 nerf_noise=m["noise"], white_back=m["white_back"], last_back=m["last_back"], _rng=rng)
    (px.square().mean() + dp.mean()).backward()
    ops.DEBUG_CAPTURE = None
    grads[bp] = {"fvol": vleaves[0].grad.clone(), **{k: p.grad.clone() for k, p in gen.named_parameters()}}
def untb(t, npi_total_rows):            # (..., T, CT, 32, 32) -> (..., T*32, CT*32)
    *lead, T, CT, _, _ = t.shape
    return t.permute(*range(len(lead)), len(lead), len(lead) + 2, len(lead) + 1, len(lead) + 3).reshape(*lead, T * 32, CT * 32).float()
rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
for pss in (0, 1):
    if ("b16", pss) not in cap: continue
    a, b = cap[("b16", pss)], cap[("f32", pss)]
    tpi, cnt = a["tpi"], a["cnt"]
    n = b["feat"].shape[0]; npi = n // cnt
    rows = torch.cat([torch.arange(npi) + i * tpi * 32 for i in range(cnt)]).to(dev)     # TB16 row of every real point
    print(f"pass {pss}: npi {npi} tpi {tpi} cnt {cnt}; scales {a['scales'].tolist()}; gmax {a['gmax'].view(torch.float32).tolist()}")
    print("  feat", rel(untb(a["feat"], n)[rows][:, :b["feat"].shape[1]], b["feat"]))
    for l in range(b["h"].shape[0]):
        print(f"  slab {l}: sin {rel(untb(a['h'][l], n)[rows], b['h'][l]):.2e} (cos: fragment-major, private to the chain) "
              f"g {rel(untb(a['g'][l], n)[rows] * a['scales'][2 * l + 1], b['g'][l]):.2e}  |g|max {b['g'][l].abs().max().item():.3e}")
    nsl = b["h"].shape[0]
    print("  go", rel(untb(a["go"], n)[rows][:, :4] * a["scales"][2 * nsl + 1], b["go"]), " |go|max", b["go"].abs().max().item())
    pad = torch.ones(a["g"].shape[1] * 32, dtype=torch.bool, device=dev); pad[rows] = False
    print("  padded G rows max |.|:", untb(a["g"][0], n)[pad].abs().max().item() if pad.any() else "none")
for k in grads["fp32"]:
    print(f"grad {k}: fp16-vs-fp32 {rel(grads['fp16'][k], grads['fp32'][k]):.2e}")
# --- point-level look at the last slab of pass 0
a, b = cap[("b16", 0)], cap[("f32", 0)]
nsl = b["h"].shape[0]; n = b["feat"].shape[0]
Wh = gen.siren.final_layer.weight.detach()                  # (4, H)
go = b["go"]                                                 # (n, 4) fp32 path
gh = go @ Wh                                                 # (n, H) expected g_h of the last slab
exp_g = gh * b["c"][nsl - 1]
got = untb(a["g"][nsl - 1], n)[rows] * a["scales"][2 * (nsl - 1) + 1]
print("last slab: fp32-path g vs host W_head^T go * cos:", rel(b["g"][nsl - 1], exp_g), " fp16-path vs host:", rel(got, exp_g))
pt = int(b["g"][nsl - 1].abs().sum(1).argmax())
print("point", pt, "ref g[:12]", [f"{v:.3e}" for v in b["g"][nsl - 1][pt, :12].tolist()])
print("point", pt, "got g[:12]", [f"{v:.3e}" for v in got[pt, :12].tolist()])
print("point", pt, "g_h  [:12]", [f"{v:.3e}" for v in gh[pt, :12].tolist()])
print("point", pt, "got/cos[:12]", [f"{v:.3e}" for v in (got[pt, :12] / b['c'][nsl - 1][pt, :12]).tolist()])
# is got a permutation of ref within the point's channels?
r_sorted, g_sorted = b["g"][nsl - 1][pt].abs().sort()[0], got[pt].abs().sort()[0]
print("sorted |.| agree (channel permutation?):", rel(g_sorted, r_sorted))
gh_sorted = (got[pt] / b['c'][nsl - 1][pt]).abs().sort()[0]
print("|got/cos| sorted vs |g_h| sorted:", rel(gh_sorted, gh[pt].abs().sort()[0]))
