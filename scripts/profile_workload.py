#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 runs: a few ImplicitGenerator3d.forward calls at 128x128x64 (B images), then the
unfused sample + composite pass of SURVEY.md 8(d) on the same samples (cnerf_gather_features over every coarse and fine
point, cnerf_composite over the merged samples), nothing else on the GPU.
Usage: rocprofv3 ... -- python3 scripts/profile_workload.py [B] [calls]      (CNERF_PRECISION=fp32|fp16x3,
CNERF_WORKLOAD=field: only the forward calls, no intermediates written -- what bench.py times;  unfused: one forward that
writes its sample points, then the unfused gather + composite launches)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnerf_amd
from cnerf_amd import ops
from cnerf_amd.generators import ImplicitGenerator3d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
R, S = 128, 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev)
gen.set_device(dev)
gen.siren.precision = os.environ.get("CNERF_PRECISION", "fp32")
fvol, glob = torch.randn(B, 32, 64, 64, 64, device=dev), torch.randn(B, 256, device=dev)
cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1)
cam[:, 2, 3] = -1.0
mode = os.environ.get("CNERF_WORKLOAD", "field")
aux = {}
with torch.no_grad():
    for _ in range(calls if mode == "field" else 1):
        px, dp = gen((fvol, glob), cam, R, 49.134342641202636, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0,
                     white_back=True, **({} if mode == "field" else {"_aux": aux}))
    if mode == "field":
        torch.cuda.synchronize()
        print("ok", float(px.mean()), float(dp.mean()), "points per field launch", B * R * R * S)
        sys.exit(0)
    fcl = ops.channel_last(fvol)
    pts = [aux[k].reshape(B, -1, 3).contiguous() for k in ("coarse_points", "fine_points")]   # one lookup launch per field pass
    allz = torch.cat([aux["fine_z"], aux["coarse_z"]], -1)
    allrs = torch.cat([aux["fine_rgb_sigma"], aux["coarse_rgb_sigma"]], -2)
    idx = aux["sort_idx"].long()
    zs = torch.gather(allz, -1, idx).reshape(B * R * R, 2 * S).contiguous()
    rss = torch.gather(allrs, -2, idx.unsqueeze(-1).expand(-1, -1, -1, 4)).reshape(B * R * R, 2 * S, 4).contiguous()
    for _ in range(calls):
        for p in pts:
            ops.gather_features(gen.siren, fcl, p)
        ops.composite(rss, zs, None, 0.0, "relu", True, False)
torch.cuda.synchronize()
print("ok", float(px.mean()), float(dp.mean()), "points per field launch", B * R * R * S, "gather points per launch", pts[0].shape[0] * pts[0].shape[1])
