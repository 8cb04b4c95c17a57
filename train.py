#!/usr/bin/env python3
"""GAN training loop around the MI355X render path on synthetic ShapeNetCar-shaped data (counterpart of the reference's
train.py:58-143 + utils.Trainer; the reference's own `train.py -o test -p 1` needs its dataset and CUDA and does not even
import: SURVEY.md F4).

    python train.py --steps 2 --img-size 32 --num-steps 12 --batch 2 --batch-split 1 --hidden 64      # plumbing check
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py    # 8 GPUs, DDP / RCCL

One process per GPU; every rank draws its own images (seed = rank, like train.py:71 of the reference); gradients are
averaged by DDP over RCCL once per optimizer step."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--img-size", type=int, default=128)
    ap.add_argument("--num-steps", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--batch-split", type=int, default=1,
                    help="gradient-accumulation chunks per step; the reference's configs use 4-6 to fit 48 GB GPUs, 288 GB of HBM take the "
                         "whole batch at once (0.218 s/step against 0.267 with 4 chunks at batch 8)")
    ap.add_argument("--voxel-res", type=int, default=64)
    ap.add_argument("--siren-type", default="SHORTSIREN_FG")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--precision", default="fp16x3", choices=["fp32", "fp16x3"],
                    help="arithmetic of the forward render: fp16x3 = fp32-accurate split products (same 1e-4 parity gate as fp32, 2.5x faster)")
    ap.add_argument("--backward-precision", default="fp16", choices=["fp32", "fp16"],
                    help="gradient GEMMs of the render backward: fp16 operands with fp32 sums (gradients within 1e-3 relative L2 of fp32 "
                         "autograd; the reference itself trains under fp16 autocast), or the exact fp32 MFMA chain (3.5x slower).  NOTE: the "
                         "defaults of this script (fp16x3 forward, fp16 backward) are the fast training arithmetic, not the fp32 parity path")
    ap.add_argument("--encoder-autocast", default=None, choices=["bf16", "fp16"],
                    help="run the 3D U-Net's convolutions under torch.autocast (the reference trains under fp16 autocast)")
    ap.add_argument("--teacher", action="store_true",
                    help="a learnable synthetic task: the target images are renders of the batch's voxel grids by a frozen, differently "
                         "initialised encoder + generator (default: random images, which nothing can fit)")
    ap.add_argument("--no-discriminator", action="store_true", help="photometric loss only (no D step, no adversarial term)")
    ap.add_argument("--lr-scale", type=float, default=1.0, help="multiplies the three learning rates of the default config")
    ap.add_argument("--no-miopen-find", action="store_true",
                    help="keep MIOpen's immediate-mode kernel choice for the Conv3d/Conv2d layers (fast start, 3x slower steps)")
    ap.add_argument("--checkpoint-dir", default=None, help="write <step>.tar (the reference's checkpoint keys) after the last step")
    ap.add_argument("--resume", default=None, help="a <step>.tar written by this harness or by the reference's trainer")
    ap.add_argument("-p", "--print-freq", type=int, default=1)
    args = ap.parse_args()

    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # generous watchdog: no rank waits for another's start-up work inside a collective any more (NodeLatch below), this is
        # the second line of defence for a slow first step
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(hours=1))
    import __graft_entry__ as ge
    import cnerf_amd                                   # (the package loads libcnerf_hip.so lazily: importable before the build)
    from cnerf_amd.training.latch import rank0_first
    rank0_first(rank, "build", ge.build)               # rank 0 compiles, the others poll a marker file (no pending collective)
    from cnerf_amd.training import GanTrainer, default_metadata
    from cnerf_amd.training.gan_step import synthetic_sample

    torch.manual_seed(rank)
    np.random.seed(rank)
    md = default_metadata(args.img_size, args.num_steps, args.batch, args.batch_split, args.siren_type, args.hidden)
    md["render_precision"] = args.precision
    md["backward_precision"] = args.backward_precision
    md["encoder_autocast"] = args.encoder_autocast
    md["enable_discriminator"] = not args.no_discriminator
    for k in ("gen_lr", "disc_lr", "enc_lr"):
        md[k] *= args.lr_scale
    md["miopen_find"] = not args.no_miopen_find
    md["encoder_channels_last"] = bool(int(os.environ.get("CNERF_ENCODER_CHANNELS_LAST", "0")))
    trainer = GanTrainer(md, dev, ddp=world > 1)
    if args.resume:
        from cnerf_amd.training import load_checkpoint
        ck = load_checkpoint(trainer, args.resume)
        if rank == 0:
            print(f"resumed from {args.resume} at step {ck['step']}", flush=True)
    gen = torch.Generator().manual_seed(1000 + rank)
    teacher = None
    if args.teacher:
        from cnerf_amd.generators import ImplicitGenerator3d
        from cnerf_amd.training.encoder import UNet3D
        torch.manual_seed(4321)                       # the same teacher on every rank
        t_enc = UNet3D(**md["unet"]).to(dev).eval()
        t_gen = ImplicitGenerator3d(**md["generator"]).to(dev).eval()
        t_gen.set_device(dev)
        t_gen.siren.precision = args.precision
        with torch.no_grad():                         # default-init heads render empty space: give the teacher colours and densities
            t_gen.siren.final_layer.weight[:3] *= 6.0
            t_gen.siren.final_layer.weight[3] *= 40.0
            t_gen.siren.final_layer.bias[3] += 0.25
        torch.manual_seed(rank)

        def teacher(sample):
            with torch.no_grad():
                imgs = []
                for c in trainer._chunks(sample["voxel"].shape[0]):
                    imgs.append(t_gen(t_enc(sample["voxel"][c]), sample["cam2world"][c], **{**md, "nerf_noise": 0.0})[0])
                sample["img"] = torch.cat(imgs, 0)
            return sample
    import threading
    first_done = threading.Event()

    def heartbeat():                       # kernel selection / compilation by MIOpen takes minutes with the search on: keep talking
        t0 = time.perf_counter()
        while not first_done.wait(60.0):
            print(f"... still selecting convolution kernels / running the first step ({time.perf_counter() - t0:.0f} s, MIOpen find mode)", flush=True)

    if rank == 0:
        threading.Thread(target=heartbeat, daemon=True).start()
    if md["miopen_find"]:
        # MIOpen's kernel search runs once per node: rank 0 first, the others after it (they find its results in the shared user
        # database instead of searching for ~7 minutes each, all at once)
        # ... and starts from the find results shipped with the repo for the default sizes (training/miopen_db.py): nothing to search then
        from cnerf_amd.training.miopen_db import use_shipped_db
        rank0_first(rank, "miopen_db", use_shipped_db)
        use_shipped_db(merge=False)
        warm = synthetic_sample(args.batch, args.img_size, args.voxel_res, dev, torch.Generator().manual_seed(1))
        def search():
            t0 = time.perf_counter()
            trainer.warm_convolutions(warm)
            torch.cuda.synchronize()
            print(f"convolution kernels selected in {time.perf_counter() - t0:.1f} s (MIOpen find mode, results in {os.environ['MIOPEN_USER_DB_PATH']})", flush=True)

        # rank 0 searches (minutes); the others poll a marker file meanwhile -- NOT a barrier: a rank waiting in an RCCL collective
        # for longer than the process group's timeout aborts the job -- then read rank 0's results from the database
        rank0_first(rank, "miopen_find", search)
        if world > 1:
            if rank != 0:
                trainer.warm_convolutions(warm)
                torch.cuda.synchronize()
            dist.barrier()
    for step in range(args.steps):
        sample = synthetic_sample(args.batch, args.img_size, args.voxel_res, dev, gen)
        if teacher is not None:
            sample = teacher(sample)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.step(sample)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        first_done.set()
        if rank == 0 and step % args.print_freq == 0:
            d_loss = trainer.losses["d"][-1] if trainer.losses["d"] else float("nan")        # (no D step without a discriminator)
            print(f"step {step}: D {d_loss:.4f}  G {trainer.losses['g'][-1]:.4f}  photo {trainer.losses['photo'][-1]:.4f}  "
                  f"alpha {trainer.alpha:.3f}  nerf_noise {md['nerf_noise']:.3f}  sec/step {dt:.3f}  "
                  f"({world * args.batch * args.img_size ** 2 * 2 / dt / 1e6:.2f} M rays/s rendered, D + G passes)", flush=True)
            if trainer.last.get("render_bwd_clamped_blocks"):
                print(f"  warning: the fp16 render backward clamped gradients in {trainer.last['render_bwd_clamped_blocks']} (tile, matrix) blocks "
                      f"(outliers beyond 32x the sampled maximum); --backward-precision fp32 is exact", flush=True)
    if args.checkpoint_dir and rank == 0:
        from cnerf_amd.training import save_checkpoint
        trainer.generator.step -= 1                       # the step that was just completed, as the reference names its files
        print("saved", save_checkpoint(trainer, args.checkpoint_dir), flush=True)
        trainer.generator.step += 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
