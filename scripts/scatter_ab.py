#!/usr/bin/env python3
"""Feature-volume gradient of the half-precision backward with the chain's own scatter (CNERF_SCATTER=chain) against the sorted patch
scatter: same addends, different summation order.  Usage: scatter_ab.py run OUT.pt [R S V B] | scatter_ab.py cmp A.pt B.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        x, y = a[k].double(), b[k].double()
        print(f"{k}: rel-L2 {((x - y).norm() / y.norm().clamp_min(1e-300)).item():.3e}  max|d|/max|ref| {((x - y).abs().max() / y.abs().max().clamp_min(1e-300)).item():.3e}  norm {y.norm().item():.4e}")
    sys.exit(0)
import cnerf_amd
from cnerf_amd.generators import ImplicitGenerator3d
out = sys.argv[2]
R, S, V, B = (int(v) for v in sys.argv[3:7]) if len(sys.argv) > 6 else (128, 64, 64, 2)
dev = torch.device("cuda:0"); torch.manual_seed(0)
gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev); gen.set_device(dev)
gen.siren.precision, gen.siren.backward_precision = "fp16x3", "fp16"
fvol = torch.randn(B, 32, V, V, V, device=dev, requires_grad=True); glob = torch.randn(B, 256, device=dev, requires_grad=True)
from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
import numpy as np
np.random.seed(1)
origin = sample_camera_positions(dev, "y", 0.9, 1.1, n=B)          # oblique views: the patch windows are not axis-aligned
cam = create_cam2world_matrix(origin, "y", device=dev)
torch.manual_seed(5)
px, dp = gen((fvol, glob), cam, R, 49.134342641202636, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=1.0, white_back=True)
(px.square().mean() + dp.mean()).backward()
torch.save({"fvol": fvol.grad.cpu(), "glob": glob.grad.cpu(), "w0": gen.siren.network[0].layer.weight.grad.cpu()}, out)
print("saved", out, fvol.grad.norm().item())
