#!/usr/bin/env python3
"""Runs MIOpen's convolution-kernel search (find mode) once for the convolution shapes of the GAN step at the bench's sizes --
UNet3D on 64^3 voxel grids, CCSDiscriminator / ProgressiveDiscriminator on 128x128 images, batch 8 and 2, forward, backward-data,
backward-weights, R1 double backward -- and leaves the results in MIOPEN_USER_DB_PATH (default gpurun_out/r3/miopen_db), so that
they can be committed under conditioned-nerf-gan_amd/training/miopen_db/ and a fresh box starts from them instead of searching
for 1.5-8 minutes (train.py, bench.py: training.miopen_db.use_shipped_db).  The encoder and the discriminator are stock
PyTorch-ROCm modules (out of the hot path's scope); this only picks their MIOpen solvers.
    python scripts/make_miopen_db.py [batches ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "gpurun_out", "r3", "miopen_db"))
os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)

import numpy as np
import torch
import cnerf_amd
from cnerf_amd.training.miopen_db import use_shipped_db
use_shipped_db(os.environ["MIOPEN_USER_DB_PATH"])        # start from what is already known: only new shapes are searched
from cnerf_amd.training import GanTrainer, default_metadata
from cnerf_amd.training.gan_step import synthetic_sample

dev = torch.device("cuda:0")
for disc in os.environ.get("CNERF_DB_DISCRIMINATORS", "CCSDiscriminator,ProgressiveDiscriminator").split(","):
    for batch in [int(a) for a in sys.argv[1:]] or [8, 2]:
        torch.manual_seed(0)
        np.random.seed(0)
        md = default_metadata(128, 64, batch, 1, "SHORTSIREN_FG", 256)
        md.update(discriminator=disc, render_precision="fp16x3", backward_precision="fp16", miopen_find=True)
        tr = GanTrainer(md, dev)
        sample = synthetic_sample(batch, 128, 64, dev, torch.Generator().manual_seed(1))
        t0 = time.perf_counter()
        tr.warm_convolutions(sample)
        torch.cuda.synchronize()
        print(f"{disc} batch {batch}: warm_convolutions {time.perf_counter() - t0:.1f} s", flush=True)
        for i in range(3):
            t0 = time.perf_counter()
            tr.step(sample)
            torch.cuda.synchronize()
            print(f"{disc} batch {batch}: step {i} {time.perf_counter() - t0:.3f} s", flush=True)
        del tr, sample
        torch.cuda.empty_cache()
print("files:", sorted(os.listdir(os.environ["MIOPEN_USER_DB_PATH"])))
