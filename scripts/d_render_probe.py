#!/usr/bin/env python3
"""Where the D step's no-grad render spends its time beyond the two field launches (GAN step at batch 8, 128x128x64, fp16x3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cnerf_amd
from cnerf_amd.training import GanTrainer, default_metadata
from cnerf_amd.training.gan_step import synthetic_sample
from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(0)
md = default_metadata(128, 64, 8, 1, "SHORTSIREN_FG", 256)
md["render_precision"], md["backward_precision"] = "fp16x3", "fp16"
tr = GanTrainer(md, dev, ddp=False)
sample = synthetic_sample(8, 128, 64, dev, torch.Generator().manual_seed(1))
for _ in range(2): tr.step(sample)
def timed(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
vox = sample["voxel"].to(dev)
with torch.no_grad():
    z = tr.encoder(vox)
    cams = create_cam2world_matrix(sample_camera_positions(dev, "y", md["cam_r_start"], md["cam_r_end"], 8), "y", dev)
    print("generator(z, cams, **md) no_grad, nerf_noise %.2f: %.2f ms" % (md["nerf_noise"], timed(lambda: tr.generator(z, cams, **md))))
    md0 = dict(md, nerf_noise=0.0)
    print("same with nerf_noise 0: %.2f ms" % timed(lambda: tr.generator(z, cams, **md0)))
    tr.generator.rng_mode = "philox"
    print("same, in-kernel Philox draws (nerf_noise %.2f): %.2f ms" % (md["nerf_noise"], timed(lambda: tr.generator(z, cams, **md))))
    tr.generator.rng_mode = "torch"
    from cnerf_amd.generators.generators import draw_rng
    try:
        print("draw_rng alone: %.2f ms" % timed(lambda: draw_rng(8, 128, 64, True, md["nerf_noise"], dev)))
    except Exception as e:
        print("draw_rng probe skipped:", e)
    for p in tr.generator.parameters():       # a parameter update between calls: the packing is redone
        pass
    def with_update():
        with torch.no_grad():
            for p in tr.generator.parameters(): p.add_(0.0)
        tr.generator(z, cams, **md)
    print("with a parameter update before every call (re-pack): %.2f ms" % timed(with_update))
