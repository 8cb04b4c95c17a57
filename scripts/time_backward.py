#!/usr/bin/env python3
"""Timing of forward+backward of the render path at 128x128x64 (informational; the headline metric is forward rays/s)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnerf_amd
from cnerf_amd.generators import ImplicitGenerator3d
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0"); torch.manual_seed(0)
VARIANT = os.environ.get("CNERF_VARIANT", "SHORTSIREN_FG")      # TALLSIREN: z = the bare feature volume, input = xyz
from cnerf_amd.generators.siren import FIELD_SPECS
gen = (ImplicitGenerator3d("TALLSIREN", 32, 3, 4, 256) if VARIANT == "TALLSIREN" else      # networks without a global feature: z_dim = C
       ImplicitGenerator3d(VARIANT, 256 if FIELD_SPECS[VARIANT].has_global else 32, 32, 4, 256)).to(dev); gen.set_device(dev)
gen.siren.precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
gen.siren.backward_precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
fvol = torch.randn(B, 32, 64, 64, 64, device=dev, requires_grad=True); glob = torch.randn(B, 256, device=dev, requires_grad=True)
cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1); cam[:, 2, 3] = -1.0
def step(bwd):
    px, dp = gen((fvol, glob) if gen.siren.spec.has_global else fvol, cam, 128, 49.134342641202636, 0.25, 1.95, 64, True, clamp_mode="relu", nerf_noise=1.0, white_back=True)
    if bwd:
        (px.square().mean() + dp.mean()).backward()
for bwd in (False, True):
    for _ in range(2): step(bwd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 3
    for _ in range(n): step(bwd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{gen.siren.precision} bwd {gen.siren.backward_precision} B={B} {'fwd+bwd' if bwd else 'fwd    '}: {dt*1e3:8.1f} ms/step  {B*128*128/dt/1e6:.3f} M rays/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
print("grad norms", fvol.grad.norm().item(), glob.grad.norm().item() if glob.grad is not None else None, gen.siren.network[0].layer.weight.grad.norm().item())
