"""A/B of the two forms of the unfused lookup (csrc/ray_kernels.hip): ms per 16.8 M lookups of the bench step (8 images, 128x128
rays, 64 coarse + 64 fine samples), and that they agree bit for bit.    python scripts/ab_gather.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    import bench
    import cnerf_amd
    from cnerf_amd import ops
    from cnerf_amd.generators import ImplicitGenerator3d
    dev = torch.device("cuda:0")
    B, R, S = 8, 128, 64
    torch.manual_seed(0)
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev)
    gen.set_device(dev)
    gen.siren.precision = "fp16x3"
    fvol, glob, cam = bench.synthetic_inputs(B, 64, 256, dev, 0)
    aux = {}
    with torch.no_grad():
        gen((fvol, glob), cam, R, bench.FOV, bench.RAY_START, bench.RAY_END, S, True, clamp_mode="relu", nerf_noise=0.0, white_back=True, _aux=aux)
        fcl = ops.channel_last(fvol)
        pts = [aux[k].reshape(B, -1, 3).contiguous() for k in ("coarse_points", "fine_points")]
        del aux
        ref = [ops.gather_features(gen.siren, fcl, p) for p in pts]          # no hint: point by point in ray order (gather_kernel)
        out = [ops.gather_features(gen.siren, fcl, p, R, S) for p in pts]    # hint: patches, distinct lines once (gather_box_kernel)
        same = all(torch.equal(a, b) for a, b in zip(ref, out))
        del ref, out
        for name, hint in (("gather_kernel (no hint)", ()), ("gather_box_kernel (R, S hint)", (R, S))):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for i in range(12):
                if i == 2:
                    ev[0].record()
                for p in pts:
                    ops.gather_features(gen.siren, fcl, p, *hint)
            ev[1].record()
            torch.cuda.synchronize()
            print(f"{name}: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms per 16.8 M lookups", flush=True)
    print("bit-identical:", same, flush=True)


if __name__ == "__main__":
    child()
