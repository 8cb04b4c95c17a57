#!/usr/bin/env python3
"""Fixed forward + backward workload for rocprofv3: B images at 128x128x(64+64), SHORTSIREN_FG hidden 256.
Usage: rocprofv3 --kernel-trace --stats ... -- python3 scripts/profile_backward.py [B] [precision] [backward_precision] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnerf_amd
from cnerf_amd.generators import ImplicitGenerator3d
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0"); torch.manual_seed(0)
VARIANT = os.environ.get("CNERF_VARIANT", "SHORTSIREN_FG")      # TALLSIREN: z = the bare feature volume, input = xyz
from cnerf_amd.generators.siren import FIELD_SPECS
gen = (ImplicitGenerator3d("TALLSIREN", 32, 3, 4, 256) if VARIANT == "TALLSIREN" else      # networks without a global feature: z_dim = C
       ImplicitGenerator3d(VARIANT, 256 if FIELD_SPECS[VARIANT].has_global else 32, 32, 4, 256)).to(dev); gen.set_device(dev)
gen.siren.precision = sys.argv[2] if len(sys.argv) > 2 else "fp16x3"
gen.siren.backward_precision = sys.argv[3] if len(sys.argv) > 3 else "fp16"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
fvol = torch.randn(B, 32, 64, 64, 64, device=dev, requires_grad=True); glob = torch.randn(B, 256, device=dev, requires_grad=True)
cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1); cam[:, 2, 3] = -1.0
for _ in range(steps):
    px, dp = gen((fvol, glob) if gen.siren.spec.has_global else fvol, cam, 128, 49.134342641202636, 0.25, 1.95, 64, True, clamp_mode="relu", nerf_noise=1.0, white_back=True)
    (px.square().mean() + dp.mean()).backward()
torch.cuda.synchronize()
print("ok", fvol.grad.norm().item())
