#!/usr/bin/env python3
"""hipGraph capture of one ImplicitGenerator3d.forward (torch.cuda.graph stream capture: every launch of the render path
goes to torch's current stream through the C ABI, so it is capturable as is).  Small renders are launch-bound: prints
eager vs replay time per call for BASELINE configs 1 and 2 shapes and checks that replay reproduces the eager image."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnerf_amd
from cnerf_amd.generators import ImplicitGenerator3d
dev = torch.device("cuda:0"); torch.manual_seed(0)
gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev); gen.set_device(dev); gen.eval()
gen.siren.precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
for (B, R, S, V) in [(2, 32, 12, 64), (1, 64, 24, 64), (1, 128, 64, 64)]:
    fvol, glob = torch.randn(B, 32, V, V, V, device=dev), torch.randn(B, 256, device=dev)
    cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1).contiguous(); cam[:, 2, 3] = -1.0
    rng = {"u_strat": torch.rand(B, R * R, S, device=dev), "u_fine": torch.rand(B, R * R, S, device=dev)}
    def call():
        with torch.no_grad():
            return gen((fvol, glob), cam, R, 49.13, 0.25, 1.95, S, True, clamp_mode="relu", nerf_noise=0.0, white_back=True, _rng=rng)
    for _ in range(3): ref = call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = call()
    g.replay(); torch.cuda.synchronize()
    same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    n = 50
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): call()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n
    print(f"{gen.siren.precision} B={B} {R}x{R}x{S}: eager {te*1e3:.3f} ms  graph replay {tg*1e3:.3f} ms  ({te/tg:.2f}x)  identical image: {same}", flush=True)
