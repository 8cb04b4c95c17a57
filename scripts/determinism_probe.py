#!/usr/bin/env python3
"""Two forwards of the same inputs and draws must agree bit for bit, stage by stage; on a difference, say where it starts (stage,
how many elements, which rays).  Diagnostic for tests/test_gpu_parity.py::test_full_size_properties.
    python scripts/determinism_probe.py [R] [S] [precision] [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnerf_amd
from cnerf_amd.generators import ImplicitGenerator3d
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
prec = sys.argv[3] if len(sys.argv) > 3 else "fp16x3"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev); gen.set_device(dev)
gen.siren.precision = prec
with torch.no_grad():
    gen.siren.final_layer.weight[3] *= 40
fvol, glob = torch.randn(1, 32, 64, 64, 64, device=dev), torch.randn(1, 256, device=dev)
cam = torch.eye(4, device=dev).unsqueeze(0).clone(); cam[0, 2, 3] = -1.0
rng = {"u_strat": torch.rand(1, R * R, S, device=dev), "u_fine": torch.rand(1, R * R, S, device=dev)}
if os.environ.get("PROBE_PRELUDE"):          # the test runs the fp32 kernel at the same size in the same process first
    gen.siren.precision = os.environ["PROBE_PRELUDE"]
    with torch.no_grad():
        gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0, white_back=True, _rng=rng, _aux={})
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev); gen.set_device(dev)      # (a fresh network, as in the next test)
    gen.siren.precision = prec
    with torch.no_grad():
        gen.siren.final_layer.weight[3] *= 40
ref = None
ORDER = ("coarse_points", "coarse_z", "coarse_rgb_sigma", "coarse_weights", "cdf", "inds", "fine_z", "fine_points", "fine_rgb_sigma", "sort_idx", "final_weights")
for it in range(reps):
    aux = {}
    with torch.no_grad():
        px, dp = gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0, white_back=True, _rng=rng, _aux=aux)
    cur = {k: v.clone() for k, v in aux.items()}; cur["pixels"], cur["depth"] = px.clone(), dp.clone()
    if ref is None:
        ref = cur
        continue
    bad = False
    for k in ORDER + ("pixels", "depth"):
        a, b = ref[k], cur[k]
        ne = (a != b) & ~(torch.isnan(a) & torch.isnan(b)) if a.dtype.is_floating_point else (a != b)
        if ne.any():
            idx = ne.nonzero()
            print(f"run {it}: stage {k}: {int(ne.sum())} of {ne.numel()} elements differ; first indices {idx[:6].tolist()}; values {a[ne][:4].tolist()} vs {b[ne][:4].tolist()}", flush=True)
            bad = True
            if k in ("coarse_rgb_sigma", "fine_rgb_sigma"):      # which 32-point tiles, which tile groups, whose turn in which block
                pts = ne.reshape(-1, 4).any(-1).nonzero().flatten()
                tiles = torch.unique(pts // 32)
                tpi = R * R * S // 32
                G = (tpi + 3) // 4
                g = tiles // 4
                cls = g * 8 // G
                idx = g - (G * cls) // 8
                nblk = 256
                print(f"   {tiles.numel()} tiles in {torch.unique(g).numel()} groups; points per differing tile min/max "
                      f"{int(torch.bincount((pts // 32 - tiles.min()).long()).clamp(min=0)[torch.bincount((pts // 32 - tiles.min()).long()) > 0].min())}/"
                      f"{int(torch.bincount((pts // 32 - tiles.min()).long()).max())}")
                print("   classes:", torch.bincount(cls.long(), minlength=8).tolist())
                print("   iteration of the block (idx // 32):", torch.unique(idx // (nblk // 8), return_counts=True))
                print("   block in class (idx % 32):", torch.bincount((idx % (nblk // 8)).long(), minlength=32).tolist())
                print("   wave (tile % 4):", torch.bincount((tiles % 4).long(), minlength=4).tolist())
    if bad or it % 50 == 0 or it == reps - 1:
        print(f"run {it}: {'DIFFERS' if bad else 'identical'}", flush=True)
