#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 runs: a few ImplicitGenerator3d.forward calls at 128x128x64 (B images),
nothing else on the GPU.  Usage: rocprofv3 ... -- python3 scripts/profile_workload.py [B] [calls]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnerf_amd
from cnerf_amd.generators import ImplicitGenerator3d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev)
gen.set_device(dev)
gen.siren.precision = os.environ.get("CNERF_PRECISION", "fp32")
fvol, glob = torch.randn(B, 32, 64, 64, 64, device=dev), torch.randn(B, 256, device=dev)
cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1)
cam[:, 2, 3] = -1.0
with torch.no_grad():
    for _ in range(calls):
        px, dp = gen((fvol, glob), cam, 128, 49.134342641202636, 0.25, 1.95, 64, True, clamp_mode="relu", nerf_noise=0.0,
                     white_back=True)
torch.cuda.synchronize()
print("ok", float(px.mean()), float(dp.mean()))
