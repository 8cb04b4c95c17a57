#!/usr/bin/env python3
"""Headline benchmark of the render path: rays/s of ImplicitGenerator3d.forward at 128x128x64 spp on synthetic
ShapeNetCar-shaped inputs (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one generator forward over this rank's B images (the four random draws, FiLM mapping, channel-last copy of the
feature volume, weight packing, coarse pass, resampling, fine pass, merge + composite).  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line with the whole-job rays/s, the roofline of the dominant kernel
(timed with HIP events recorded inside the timed region) and, at N=1, the CPU oracle timed on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FOV, RAY_START, RAY_END = 49.134342641202636, 0.25, 1.95     # configs/thousand/special.py:35-41 of the reference
PEAK_F32_MFMA_TFLOPS = 157.3                                  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBPS = 8000.0


def macs_per_point(C, H, n_layers):
    return C * H + (n_layers - 1) * H * H + H * 4             # SHORTSIREN_FG: 205,824


class HipEvents:
    """hipEvent_t pairs through ctypes (the C ABI records them around the field kernel on the launch stream)."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]

    def create(self, n):
        out = []
        for _ in range(n):
            e = ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
            out.append(e.value)
        return out

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        assert self.hip.hipEventSynchronize(b) == 0
        assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
        return ms.value


def timed_region(step, steps, warmup, dist=None, sync=lambda: None, device="cpu"):
    """W untimed warm-up steps, then exactly K steps bracketed by device sync + barrier on both sides; returns the
    MAX over ranks of the elapsed seconds.  `dist` is torch.distributed once a process group exists (RCCL on the GPU
    box, gloo in the CPU test), None for a single process."""
    for _ in range(warmup):
        step(None)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed


def whole_job_rays_per_s(world, images_per_gpu, img_size, steps, elapsed):
    """Weak scaling: every rank renders its own images; the job's throughput is all ranks' rays over the slowest rank's time."""
    return world * images_per_gpu * img_size * img_size * steps / elapsed


def synthetic_inputs(B, V, Z, dev, seed):
    """ShapeNetCar-shaped synthetic inputs: unet3d-like feature volume, global feature, cameras."""
    from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
    g = torch.Generator().manual_seed(seed)
    fvol = torch.randn(B, 32, V, V, V, generator=g)
    glob = torch.randn(B, Z, generator=g)
    np.random.seed(seed)
    cam = create_cam2world_matrix(sample_camera_positions("cpu", "y", 0.7, 1.5, B), "y")
    return fvol.to(dev), glob.to(dev), cam.to(dev)


def sample_composite_pass(args, gen, fvol, glob, cam, meta, evs):
    """The unfused sample + composite pass of SURVEY.md 8(d): trilinear lookup that materialises the (N,32) features
    (cnerf_gather_features) for every coarse and fine sample of one step, plus the compositing of the merged samples
    (cnerf_composite).  Algorithmic bytes: 1152 B per field evaluation, 20 B per composited sample, 16 B per ray."""
    import cnerf_amd
    ops = cnerf_amd.ops
    B, R, S = args.batch, args.img_size, args.num_steps
    aux = {}
    with torch.no_grad():
        gen(zin(gen, fvol, glob), cam, R, FOV, RAY_START, RAY_END, S, _aux=aux, **meta)
        fcl = ops.channel_last(fvol)
        pts = torch.cat([aux["coarse_points"].reshape(B, -1, 3), aux["fine_points"].reshape(B, -1, 3)], 1).contiguous()
        allz = torch.cat([aux["fine_z"], aux["coarse_z"]], -1)
        allrs = torch.cat([aux["fine_rgb_sigma"], aux["coarse_rgb_sigma"]], -2)
        idx = aux["sort_idx"].long()
        zs = torch.gather(allz, -1, idx).reshape(B * R * R, 2 * S).contiguous()
        rss = torch.gather(allrs, -2, idx.unsqueeze(-1).expand(-1, -1, -1, 4)).reshape(B * R * R, 2 * S, 4).contiguous()
        del aux
        reps = 10
        ev = evs.create(4)
        for i in range(reps + 2):
            if i == 2:
                evs.hip.hipEventRecord(ev[0], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            ops.gather_features(gen.siren, fcl, pts)
        evs.hip.hipEventRecord(ev[1], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        for i in range(reps + 2):
            if i == 2:
                evs.hip.hipEventRecord(ev[2], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            ops.composite(rss, zs, None, 0.0, "relu", True, False)
        evs.hip.hipEventRecord(ev[3], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    t_g = evs.elapsed_ms(ev[0], ev[1]) / reps
    t_c = evs.elapsed_ms(ev[2], ev[3]) / reps
    evals, rays = pts.shape[0] * pts.shape[1], B * R * R
    bytes_g, bytes_c = evals * 1152.0, evals * 20.0 + rays * 16.0
    gbps = (bytes_g + bytes_c) / ((t_g + t_c) * 1e-3) / 1e9
    return {"kernels": "gather_kernel (cnerf_gather_features) + composite_kernel (cnerf_composite), unfused",
            "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
            "traffic": None, "gather_ms": t_g, "composite_ms": t_c, "algorithmic_bytes": bytes_g + bytes_c,
            "note": "effective bandwidth: neighbouring samples share corner lines in L2 / Infinity Cache (SURVEY.md 8d)"}


def zin(gen, fvol, glob):
    """The generator's `z`: (feature volume, global feature) for the globally conditioned families, the volume alone otherwise."""
    return (fvol, glob) if gen.siren.spec.has_global else fvol


def fast_path(args, gen, fvol, glob, cam, meta, evs):
    """Secondary measurement (not `value`): the same step with precision = "fp16x3" -- every fp32 product evaluated as three
    fp16 MFMAs (fp32 accumulate) on two-way fp16 splits of both operands.  Same parity gates as the fp32 path
    (tests/test_gpu_parity.py::test_split_precision: rgb / sigma within 1e-4 of the reference, measured at its fp32
    noise floor); reported separately so that the headline number stays plain fp32 MFMA arithmetic."""
    B, R, S = args.batch, args.img_size, args.num_steps
    gen.siren.precision = "fp16x3"
    steps = max(3, args.steps // 2)
    events = evs.create(4 * steps)
    try:
        with torch.no_grad():
            for _ in range(2):
                gen(zin(gen, fvol, glob), cam, R, FOV, RAY_START, RAY_END, S, **meta)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                gen(zin(gen, fvol, glob), cam, R, FOV, RAY_START, RAY_END, S, _field_events=events[4 * i:4 * i + 4], **meta)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
    finally:
        gen.siren.precision = "fp32"
    ms = float(np.mean([evs.elapsed_ms(events[4 * i + k], events[4 * i + k + 1]) for i in range(steps) for k in (0, 2)]))
    flops = 2.0 * macs_per_point(32, args.hidden, len(gen.siren.spec.layers)) * B * R * R * S
    return {"value": B * R * R / dt, "unit": "rays/s", "ms_per_step": dt * 1e3, "dtype": "fp16x3 (fp32-equivalent split, fp32 accumulate)",
            "kernel": "field_h3_kernel<8>", "avg_launch_ms": ms,
            "algorithmic_tflops": flops / (ms * 1e-3) / 1e12,
            "fp16_mfma_tflops": 3 * flops / (ms * 1e-3) / 1e12, "fp16_mfma_peak": 2500.0,
            "frac_of_fp16_mfma_peak": 3 * flops / (ms * 1e-3) / 1e12 / 2500.0}


def cpu_baseline(args, gen_cpu):
    """The CPU oracle (same ATen op sequence as the reference's CPU path) on one image of the same workload."""
    from oracle import render_oracle as O
    R, S = args.img_size, args.num_steps
    g = torch.Generator().manual_seed(1234)
    fvol = torch.randn(1, 32, args.volume, args.volume, args.volume, generator=g)
    glob = torch.randn(1, args.z_dim, generator=g)
    cam = torch.eye(4).unsqueeze(0).clone()
    cam[0, 2, 3] = -1.0
    params = {k: v.detach() for k, v in gen_cpu.siren.state_dict().items()}
    # a 1-GPU box grants a 16-core share of the host (more threads than that only oversubscribe it)
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    times = []
    with torch.no_grad():
        for i in range(1 + args.cpu_reps):
            u1, u2 = torch.rand(1, R * R, S, generator=g), torch.rand(1, R * R, S, generator=g)
            t0 = time.perf_counter()
            O.render(args.variant, params, fvol, glob, cam, R, FOV, RAY_START, RAY_END, S, True, "relu", 0.0, True, False,
                     u1, None, u2, None)
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
    t = float(np.median(times))
    return {"value": R * R / t, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 image {R}x{R}x{S} hierarchical fp32 no_grad, median of {args.cpu_reps} after 1 warm-up "
                      f"({t:.2f} s each), oracle/render_oracle.py (ATen op sequence of the reference CPU path)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--img-size", type=int, default=128)
    ap.add_argument("--num-steps", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (BASELINE config 4: 8/GPU)")
    ap.add_argument("--volume", type=int, default=64)
    ap.add_argument("--variant", default="SHORTSIREN_FG")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--z-dim", type=int, default=256)
    ap.add_argument("--input-dim", type=int, default=32, help="generator input_dim (3 for TALLSIREN, whose z_dim is the feature width 32)")
    ap.add_argument("--noise", type=float, default=0.0)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16x3"],
                    help="arithmetic of the MLP products: exact fp32 MFMA, or the fp32-accurate fp16x3 split")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-path", action="store_true", help="skip the secondary fp16x3 measurement")
    ap.add_argument("--cpu-reps", type=int, default=2)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)     # RCCL; only the barrier and the max-over-ranks use it

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d

    torch.manual_seed(0)                                      # reference default init under seed 0
    gen_cpu = ImplicitGenerator3d(args.variant, args.z_dim, args.input_dim, 4, args.hidden)
    import copy
    gen = copy.deepcopy(gen_cpu).to(dev)
    gen.set_device(dev)
    gen.siren.precision = args.precision
    gen.eval()
    B, R, S = args.batch, args.img_size, args.num_steps
    fvol, glob, cam = synthetic_inputs(B, args.volume, args.z_dim, dev, seed=rank)
    meta = dict(clamp_mode="relu", nerf_noise=args.noise, white_back=True, hierarchical_sample=True)
    meta_nohier = dict(meta)

    evs = HipEvents()
    n_ev = 4 * args.steps
    events = evs.create(n_ev)

    def step(i=None):
        ev = events[4 * i:4 * i + 4] if i is not None else None
        cnerf_amd.ops._pack_cache.clear()                     # training changes the weights every step: re-pack
        with torch.no_grad():
            return gen(zin(gen, fvol, glob), cam, R, FOV, RAY_START, RAY_END, S, _field_events=ev, **meta)

    last = {}

    def step_i(i):
        last["out"] = step(i)

    elapsed = timed_region(step_i, args.steps, args.warmup, dist if world > 1 else None, torch.cuda.synchronize, dev)
    out = last["out"]
    assert torch.isfinite(out[0]).all()

    # dominant kernel: field_tile_kernel, two launches per step (coarse, fine), same work each
    kern_ms = []
    for i in range(args.steps):
        e = events[4 * i:4 * i + 4]
        kern_ms += [evs.elapsed_ms(e[0], e[1]), evs.elapsed_ms(e[2], e[3])]
    avg_ms = float(np.mean(kern_ms))
    n_layers = len(gen.siren.spec.layers)
    flops_per_launch = 2.0 * macs_per_point(32, args.hidden, n_layers) * B * R * R * S
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
    split = args.precision == "fp16x3"
    if split:      # priced per issued flop: three fp16 MFMAs per fp32 product, against the dense fp16 MFMA peak
        roof = {"kernel": "field_h3_kernel<8> (sample + trilinear lookup + FiLM-SIREN MLP, fp16x3 split, fp32 accumulate)",
                "bound": "mfma", "achieved": 3 * achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": 3 * achieved / 2500.0,
                "traffic": None, "algorithmic_tflops": achieved}
    else:
        roof = {"kernel": "field_tile_kernel<8> (sample + trilinear lookup + FiLM-SIREN MLP, fp32 MFMA)",
                "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None}
    # HBM bytes cannot be counted from inside this process.  `traffic` is the rocprofv3 PMC measurement of the same kernels on
    # the same workload shape at batch 2 (scripts/pmc_traffic.sh -> profiles/r01_field_kernel_profile.md: FETCH_SIZE 31,290 KB
    # x 2 for gfx950's wide reads + WRITE_SIZE 36,860 KB per launch of 2,097,152 points, identical for both field kernels =
    # 48.5 B per point: the channel-last volumes read once, rgb_sigma and z written once) scaled to this launch's points.
    PMC_BYTES_PER_POINT = (2 * 31290.0 + 36860.0) * 1024.0 / 2097152.0
    roof["traffic"] = PMC_BYTES_PER_POINT * B * R * R * S if (R, S, args.volume) == (128, 64, 64) else None
    roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE at batch 2 (profiles/r01_field_kernel_profile.md), scaled by points per launch"
    roof.update({"avg_launch_ms": avg_ms, "launches": len(kern_ms), "flops_per_launch": flops_per_launch,

                 "share_of_step": 2 * avg_ms / (elapsed / args.steps * 1e3)})

    if rank == 0:
        rays = world * B * R * R * args.steps
        res = {
            "metric": "rays/sec at 128x128x64spp ShapeNetCar",
            "value": whole_job_rays_per_s(world, B, R, args.steps, elapsed), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp16x3 (fp32-equivalent split, fp32 accumulate)" if split else "f32", "data": "synthetic",
            "config": {"workload": f"ImplicitGenerator3d.forward {args.variant} hidden {args.hidden}, {R}x{R} rays x {S} "
                                   f"coarse + {S} fine samples, feature volume 32x{args.volume}^3, batch {B}/GPU, "
                                   f"hierarchical, white_back, relu, nerf_noise {args.noise}",
                       "img_size": R, "num_steps": S, "images_per_gpu": B, "mpts_per_s": rays / elapsed * 2 * S / 1e6,
                       "parallelism": f"image-batch data parallel x{world}, no data-path collective"},
            "roofline": roof,
        }
        if world == 1:
            res["roofline_sample_composite"] = sample_composite_pass(args, gen, fvol, glob, cam, meta, evs)
        if world == 1 and args.precision == "fp32" and not args.no_fast_path:
            res["fp16x3_split_path"] = fast_path(args, gen, fvol, glob, cam, meta, evs)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, gen_cpu)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
