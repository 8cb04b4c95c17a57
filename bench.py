#!/usr/bin/env python3
"""Headline benchmark of the render path: rays/s of ImplicitGenerator3d.forward at 128x128x64 spp on synthetic
ShapeNetCar-shaped inputs (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU: either the driver starts the ranks itself (`python -m torch.distributed.run ... bench.py
--gpus N`, WORLD_SIZE in the environment) or this process does -- with WORLD_SIZE unset it only launches N children
through torch.distributed.run BEFORE anything touches the GPU, relays rank 0's JSON line and returns their exit code.

A step = one generator forward over this rank's B images (the four random draws, FiLM mapping, channel-last copy of the
feature volume, weight packing, coarse pass, resampling, fine pass, merge + composite).  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line with the whole-job rays/s, the roofline of the dominant kernel
(timed with HIP events recorded inside the timed region) and, at N=1, the CPU oracle timed on image 0 of the SAME inputs
and draws -- whose output is also the checker of the timed run (`check`): the run fails if the image the GPU produced in
its last timed step does not match it.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FOV, RAY_START, RAY_END = 49.134342641202636, 0.25, 1.95     # configs/thousand/special.py:35-41 of the reference
PEAK_F32_MFMA_TFLOPS = 157.3                                  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBPS = 8000.0
CHECK_SEED = 4242                                             # torch.cuda seed of the draws of the last timed step
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel (scripts/pmc_traffic.sh)


def n_matrices(net):
    from cnerf_amd import ops
    return ops.n_matrices(net)


def macs_per_point(C, H, n_layers, net=None):
    """n_layers = weight matrices before the head (a residual block counts two).  Per-point FiLM networks (TALLSIREN): the mapping
    MLP runs per point -- Wm1 (C x 256), 2 n_layers H rows of Wm2 (256 wide) -- and layer 0 reads the position (3 inputs)."""
    if net is not None and net.spec.layers[0] == "pfilm":
        return C * 256 + 2 * n_layers * H * 256 + 3 * H + (n_layers - 1) * H * H + H * 4      # TALLSIREN, H 256: 1,517,312
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

    def record(self, ev):
        self.hip.hipEventRecord(ev, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

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


def zin(gen, fvol, glob):
    """The generator's `z`: (feature volume, global feature) for the globally conditioned families, the volume alone otherwise."""
    return (fvol, glob) if gen.siren.spec.has_global else fvol


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks as children (never exec, never after the GPU is initialised)
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_children(n, argv):
    """Runs `python -m torch.distributed.run --nproc-per-node n bench.py <argv>` as a child process, forwards its output
    (rank 0's JSON line on stdout) and returns its exit code.  This process has not touched the GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for line in proc.stdout:
        lines.append(line)
    rc = proc.wait()
    json_lines = [l for l in lines if l.lstrip().startswith("{") and '"metric"' in l]
    for l in lines:
        if l not in json_lines:
            sys.stderr.write(l)
    if json_lines:
        sys.stdout.write(json_lines[-1])
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks printed no result line\n")
        rc = 1
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# secondary measurements (N = 1)
# ---------------------------------------------------------------------------------------------------------------------
def pmc_bytes(kernel_prefix):
    """HBM bytes per unit of work of one kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written
    by scripts/pmc_traffic.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled as the guide prescribes
    for gfx950's wide reads).  None when the file or the kernel is missing."""
    try:
        with open(PMC_FILE) as f:
            tab = json.load(f)
    except (OSError, ValueError):
        return None
    for k, v in tab.get("kernels", {}).items():
        if k.startswith(kernel_prefix):
            return v
    return None


def sample_composite_pass(args, gen, fvol, glob, cam, meta, evs):
    """The unfused sample + composite pass of SURVEY.md 8(d): trilinear lookup that materialises the (N,32) features
    (cnerf_gather_features) for every coarse and fine sample of one step, plus the compositing of the merged samples
    (cnerf_composite).  Two rates are reported: `frac` = HBM bytes measured by rocprofv3 PMC counters / time / 8 TB/s (the
    roofline fraction), and `l2_side` = the algorithmic bytes of SURVEY 8(d) (1152 B per field evaluation, 20 B per
    composited sample, 16 B per ray -- mostly corner lines served by L2 / Infinity Cache) / time."""
    import cnerf_amd
    ops = cnerf_amd.ops
    B, R, S = args.batch, args.img_size, args.num_steps
    aux = {}
    with torch.no_grad():
        gen(zin(gen, fvol, glob), cam, R, FOV, RAY_START, RAY_END, S, _aux=aux, **meta)
        fcl = ops.channel_last(fvol)
        pts = [aux["coarse_points"].reshape(B, -1, 3).contiguous(), aux["fine_points"].reshape(B, -1, 3).contiguous()]
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
                evs.record(ev[0])
            for p in pts:               # the lookups of the coarse and of the fine pass, as the fused kernels make them
                ops.gather_features(gen.siren, fcl, p)
        evs.record(ev[1])
        for i in range(reps + 2):
            if i == 2:
                evs.record(ev[2])
            ops.composite(rss, zs, None, 0.0, "relu", True, False)
        evs.record(ev[3])
        torch.cuda.synchronize()
    t_g = evs.elapsed_ms(ev[0], ev[1]) / reps
    t_c = evs.elapsed_ms(ev[2], ev[3]) / reps
    evals, rays = sum(p.shape[0] * p.shape[1] for p in pts), B * R * R
    bytes_g, bytes_c = evals * 1152.0, evals * 20.0 + rays * 16.0
    l2_gbps = (bytes_g + bytes_c) / ((t_g + t_c) * 1e-3) / 1e9
    out = {"kernels": "gather_kernel (cnerf_gather_features) + composite_kernel (cnerf_composite), unfused",
           "bound": "hbm", "peak": PEAK_HBM_GBPS, "unit": "GB/s", "gather_ms": t_g, "composite_ms": t_c,
           "algorithmic_bytes": bytes_g + bytes_c,
           "l2_side": {"achieved": l2_gbps, "peak": 34500.0, "unit": "GB/s", "frac": l2_gbps / 34500.0,
                       "note": "algorithmic bytes / time: neighbouring samples share corner lines, so most of them are "
                               "served by the XCD L2s (34.5 TB/s aggregate) and the Infinity Cache, not by HBM"}}
    pg, pc = pmc_bytes("cnerf::gather_kernel"), pmc_bytes("cnerf::composite_kernel")
    if pg and pc and (R, S, args.volume) == (128, 64, 64):
        traffic = pg["bytes_per_point"] * evals + pc["bytes_per_point"] * evals
        out.update({"traffic": traffic, "achieved": traffic / ((t_g + t_c) * 1e-3) / 1e9,
                    "traffic_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes ({os.path.relpath(PMC_FILE, ROOT)}), per point x points"})
        out["frac"] = out["achieved"] / PEAK_HBM_GBPS
    else:
        out.update({"traffic": None, "achieved": None, "frac": None,
                    "traffic_source": "no PMC measurement for this shape: HBM fraction not reported"})
    return out


def fast_path(args, gen, fvol, glob, cam, meta, evs, precision="fp16x3"):
    """Secondary measurement (not `value`): the same step with precision = "fp16x3" -- every fp32 product evaluated as three
    fp16 MFMAs (fp32 accumulate) on two-way fp16 splits of both operands.  Same parity gates as the fp32 path
    (tests/test_gpu_parity.py::test_split_precision: rgb / sigma within 1e-4 of the reference, measured at its fp32
    noise floor); reported separately so that the headline number stays plain fp32 MFMA arithmetic."""
    B, R, S = args.batch, args.img_size, args.num_steps
    gen.siren.precision = precision
    mfmas = 3 if precision == "fp16x3" else 1             # fp16 MFMAs issued per fp32 product
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
    flops = 2.0 * macs_per_point(32, args.hidden, n_matrices(gen.siren), gen.siren) * B * R * R * S
    pfilm = gen.siren.spec.layers[0] == "pfilm"
    return {"value": B * R * R / dt, "unit": "rays/s", "ms_per_step": dt * 1e3,
            "dtype": "fp16x3 (fp32-equivalent split, fp32 accumulate)" if mfmas == 3 else "fp16 products, fp32 accumulate (tolerance 2e-2, not the 1e-4 gate)",
            "kernel": ("h3::pw::field_pw16_kernel<8>" if mfmas == 3 else "h1::pw::field_pw16_kernel<8> (single pass)") if pfilm else
                      ("field_h3_kernel<8>" if mfmas == 3 else "h1::field_h3_kernel<8> (single pass)"), "avg_launch_ms": ms,
            "algorithmic_tflops": flops / (ms * 1e-3) / 1e12,
            "fp16_mfma_tflops": mfmas * flops / (ms * 1e-3) / 1e12, "fp16_mfma_peak": PEAK_F16_MFMA_TFLOPS,
            "frac_of_fp16_mfma_peak": mfmas * flops / (ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS}


def train_step_timing(args, gen, fvol, glob, cam, evs):
    """Secondary measurement: forward + backward of the render path (what the generator step of the GAN loop adds to the
    forward), nerf_noise 1.0 as at training step 0, gradients w.r.t. field parameters, FiLM mapping, feature volume and
    global feature.  Two settings: everything exact fp32 (fp32 MFMA forward, chain and weight gradients), and the training
    setting of the harness -- fp16x3 forward (fp32-accurate) whose activations are kept as fp16 tile blocks, gradient GEMMs
    on the fp16 MFMA with fp32 sums (tests/test_gpu_parity.py::test_backward_half_precision).  Priced against 3x the
    forward's algorithmic FLOPs (forward + two gradient GEMMs per layer)."""
    B, R, S = args.batch, args.img_size, args.num_steps
    meta = dict(clamp_mode="relu", nerf_noise=1.0, white_back=True, hierarchical_sample=True)
    flops = 3 * 2 * 2.0 * macs_per_point(32, args.hidden, n_matrices(gen.siren), gen.siren) * B * R * R * S
    out = {"images": B, "forward_ms_fp32": None}
    keep = (gen.siren.precision, getattr(gen.siren, "backward_precision", "fp32"))
    gen.train()
    try:
        for name, prec, bprec in (("fp32", "fp32", "fp32"), ("fp16x3_forward_fp16_backward", "fp16x3", "fp16")):
            gen.siren.precision, gen.siren.backward_precision = prec, bprec
            fv = fvol.detach().clone().requires_grad_(True)
            gl = glob.detach().clone().requires_grad_(True)
            torch.cuda.reset_peak_memory_stats()
            times, fwd = [], []
            for i in range(5):       # (the first two settle the caching allocator: it takes its second multi-GiB block in iteration 1)
                for p in gen.parameters():
                    p.grad = None
                fv.grad = gl.grad = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                px, dp = gen(zin(gen, fv, gl), cam, R, FOV, RAY_START, RAY_END, S, **meta)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                (px.square().mean() + dp.mean()).backward()
                torch.cuda.synchronize()
                if i > 1:
                    times.append(time.perf_counter() - t0)
                    fwd.append(t1 - t0)
            dt = float(np.median(times))
            out[name] = {"fwd_bwd_ms": dt * 1e3, "forward_part_ms": float(np.median(fwd)) * 1e3, "rays_per_s_fwd_bwd": B * R * R / dt,
                         "achieved_tflops_over_3x_forward_flops": flops / dt / 1e12, "peak_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}
            del fv, gl, px, dp
            torch.cuda.empty_cache()
    finally:
        gen.siren.precision, gen.siren.backward_precision = keep
        gen.eval()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the GAN training step (BASELINE configs 3 / 4): encoder + render forward / backward + discriminator, DDP gradient all-reduce
# ---------------------------------------------------------------------------------------------------------------------
class StubGenerator(torch.nn.Module):
    """`--device cpu` only (launcher / DDP plumbing test on gloo: the render has no CPU path): a small differentiable module with
    the generator's call signature.  Never part of a measurement -- the line says "stub"."""

    def __init__(self, z_dim):
        super().__init__()
        self.lin = torch.nn.Linear(z_dim + 32, 3 * 4 * 4)
        self.step = self.epoch = 0

    def forward(self, z, cam2worlds, img_size, *args, **kwargs):
        fvol, glob = z
        px = torch.tanh(self.lin(torch.cat([glob, fvol.mean(dim=(2, 3, 4))], -1))).reshape(-1, 3, 4, 4)
        px = torch.nn.functional.interpolate(px + 0.01 * cam2worlds[:, :3, 3].reshape(-1, 3, 1, 1), size=(img_size, img_size), mode="bilinear")
        return px, px.mean(1)


def allreduce_bench(dist, dev, nbytes, reps=10):
    """Stand-alone all-reduce (sum) of one flat fp32 buffer of `nbytes`, the size of a step's gradient traffic: ms per call,
    algorithm and bus bandwidth (bus = algorithm x 2 (N - 1) / N, the per-link load of a ring; 0 for one rank)."""
    n = dist.get_world_size()
    buf = torch.ones(max(1, nbytes // 4), dtype=torch.float32, device=dev)
    sync = torch.cuda.synchronize if torch.device(dev).type == "cuda" else (lambda: None)
    for _ in range(2):
        dist.all_reduce(buf)
    sync()
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf)
    sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    alg = buf.numel() * 4 / (ms * 1e-3) / 1e9
    return {"bytes": buf.numel() * 4, "ms": ms, "algbw_GBps": alg, "busbw_GBps": alg * 2 * (n - 1) / n}


def gan_step_leg(args, dev, rank, world, dist, batch, precision, backward_precision, steps, stub=False):
    """K full GAN steps (GanTrainer.step: UNet3D encoder -> HIP render -> CCSDiscriminator, D step with R1, G step, Adam) with the
    three networks wrapped in DDP over the job's process group -- RCCL on the GPU box, also for a single rank -- on this rank's own
    synthetic batch; the reference's counterpart is train.py:36-44 + utils.py:322-352,621-842.  Returns the step time (max over
    ranks), its split into phases on rank 0 (events on torch's stream), and what DDP all-reduced per optimizer step (communication
    hooks: calls = buckets, bytes, rounds)."""
    import cnerf_amd
    from cnerf_amd import ops
    from cnerf_amd.training import GanTrainer, default_metadata
    from cnerf_amd.training.gan_step import PhaseTimer, synthetic_sample
    from cnerf_amd.training.miopen_db import use_shipped_db
    if not stub and not args.no_miopen_db:
        from cnerf_amd.training.latch import rank0_first
        rank0_first(rank, f"miopen_db_{batch}_{precision}", use_shipped_db)        # one rank merges the files, ...
        use_shipped_db(merge=False)                                                # ... every rank reads them
        # FAST find mode: a problem the database knows gets its recorded solver, one it does not falls back to immediate mode -- never
        # a minutes-long search inside the benchmark
        os.environ.setdefault("MIOPEN_FIND_MODE", "2")
    R, S = args.img_size, args.num_steps
    md = default_metadata(R, S, batch, 1, args.variant, args.hidden)
    md["discriminator"] = "CCSDiscriminator"               # the "sgdiscriminator" of BASELINE config 3
    # MIOpen find mode answered from the find results shipped with the repo (training/miopen_db.py); a shape they do not cover is
    # searched on the spot (minutes) -- `--no-miopen-db` keeps immediate mode instead (no search, 8x slower convolutions)
    md["render_precision"], md["backward_precision"], md["miopen_find"] = precision, backward_precision, not stub and not args.no_miopen_db
    modules = None
    if stub:
        md["unet"].update(f_maps=8, num_levels=2)
        md["generator"]["z_dim"] = 16
        modules = {"generator": StubGenerator(16)}
    torch.manual_seed(0)                                     # identical initial parameters on every rank
    tr = GanTrainer(md, dev, ddp=True, modules=modules)
    meters = tr.attach_comm_meters()
    sample = synthetic_sample(batch, R, 8 if stub else args.volume, dev, torch.Generator().manual_seed(100 + rank))
    np.random.seed(rank)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    if dev.type == "cuda":
        torch.cuda.reset_peak_memory_stats()
    if not stub and world > 1:
        # MIOpen compiles the solvers it selects on first use (~2 minutes for this step on a fresh node): rank 0 first, without any
        # collective (raw modules; the others wait on a marker file), then the others find the binaries in the node's kernel cache
        from cnerf_amd.training.latch import rank0_first
        rank0_first(rank, f"miopen_warm_{batch}_{precision}", lambda: (tr.warm_convolutions(sample), sync()))
        if rank != 0:
            tr.warm_convolutions(sample)
            sync()
    tr.step(sample)                                          # warm-up: kernel selection, DDP bucket rebuild, allocator
    tr.step(sample)
    for m in meters.values():
        m.take()
    tr.timer = ops.PHASE_TIMER = PhaseTimer(dev)
    try:
        elapsed = timed_region(lambda i: tr.step(sample), steps, 0, dist, sync, dev)
        phases = {k: v / steps for k, v in tr.timer.summary().items()}
    finally:
        tr.timer = ops.PHASE_TIMER = None
    comm = {k: {kk: vv / steps for kk, vv in m.take().items()} for k, m in meters.items()}
    n_params = {k: sum(p.numel() for p in m.parameters()) for k, m in (("generator", tr.generator), ("encoder", tr.encoder), ("discriminator", tr.discriminator))}
    out = {"images_per_gpu": batch, "render_precision": precision, "backward_precision": backward_precision,
           "ms_per_step": elapsed / steps * 1e3, "steps": steps,
           "rays_per_s_whole_job": world * batch * R * R * 2 * steps / elapsed,
           "rays_note": "every image is rendered twice per step: no-grad for the D step, with grad for the G step",
           "phases_ms_rank0": phases, "parameters": n_params,
           "allreduce_per_optimizer_step": comm,
           "allreduce_bytes_per_step": sum(c["bytes"] for c in comm.values()),
           "losses": {k: float(v) for k, v in tr.last.items() if k in ("d_loss", "g_loss", "photo_loss")}}
    if dev.type == "cuda":
        out["peak_mem_gib"] = torch.cuda.max_memory_allocated() / 2 ** 30
    del tr, sample
    if dev.type == "cuda":
        torch.cuda.empty_cache()
    return out


def train_step_ddp(args, dev, rank, world, dist, stub=False):
    """The `train_step_ddp` object of the JSON line (every N, also N = 1 with a one-rank group) and, at N = 1, the `gan_step`
    list (config 3 as BASELINE words it: batch 2 -- the reference's accumulation chunk, configs/thousand/special.py:24-30 -- and
    batch 8, exact fp32 and the training default fp16x3 forward / fp16 backward)."""
    steps = max(2, min(5, args.steps // 4)) if not stub else 2
    main_leg = gan_step_leg(args, dev, rank, world, dist, args.batch if not stub else 4, "fp16x3", "fp16", steps, stub)
    ddp = dict(main_leg)
    ddp.update({"n_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                "what": "GanTrainer(ddp=True).step: DDP(generator) + DDP(encoder) + DDP(discriminator), one gradient all-reduce round per "
                        "optimizer step and network (G + E in the G step, D in the D step)"})
    ar = {}
    for name, c in main_leg["allreduce_per_optimizer_step"].items():
        if c["bytes"] > 0:
            ar[name] = allreduce_bench(dist, dev, int(c["bytes"]), reps=10 if not stub else 3)
    ddp["allreduce_standalone"] = ar
    t_ar = sum(v["ms"] for v in ar.values())
    ddp["allreduce_ms_per_step_if_not_overlapped"] = t_ar
    ddp["allreduce_share_of_step"] = t_ar / main_leg["ms_per_step"]
    legs = None
    if world == 1 and not stub:
        legs = [main_leg]
        for b, pr, bp in ((args.batch, "fp32", "fp32"), (2, "fp16x3", "fp16"), (2, "fp32", "fp32")):
            legs.append(gan_step_leg(args, dev, rank, world, dist, b, pr, bp, steps))
    return ddp, legs


def host_cores():
    """Threads for the CPU leg = the CPU share this process really has: the cgroup quota if one is set, else the affinity
    mask, and never more than 16 -- the GPU box exposes all 256 host cores in the mask but grants a 1-GPU job a 16-core
    share (measured: 256 oracle threads there take 30 s per image, 16 take 7 s).  CNERF_CPU_THREADS overrides."""
    if os.environ.get("CNERF_CPU_THREADS"):
        return max(1, int(os.environ["CNERF_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def cpu_baseline(args, gen_cpu, fvol, glob, cam, draws):
    """The CPU oracle (same ATen op sequence as the reference's CPU path) on image 0 of the bench's own inputs and draws.
    Returns (baseline dict, oracle output of that image): the output is the checker of the timed GPU run."""
    from oracle import render_oracle as O
    R, S = args.img_size, args.num_steps
    params = {k: v.detach() for k, v in gen_cpu.siren.state_dict().items()}
    cores = host_cores()                       # the affinity mask the box grants this process (16 for a 1-GPU share)
    torch.set_num_threads(cores)
    f0, c0 = fvol[:1].cpu(), cam[:1].cpu()
    g0 = glob[:1].cpu() if gen_cpu.siren.spec.has_global else None
    B = fvol.shape[0]
    d0 = {k: v.reshape(B, R * R, -1)[:1].cpu() for k, v in draws.items()}       # image 0 of every draw, (1, P, S or 2S)
    times, ref = [], None
    with torch.no_grad():
        for i in range(1 + args.cpu_reps):
            t0 = time.perf_counter()
            out = O.render(args.variant, params, f0, g0, c0, R, FOV, RAY_START, RAY_END, S, True, "relu", args.noise, True, False,
                           d0["u_strat"], d0.get("eps_coarse"), d0["u_fine"], d0.get("eps_final"))
            dt = time.perf_counter() - t0
            if i == 0:
                ref = out
            else:
                times.append(dt)
    t = float(np.median(times)) if times else dt
    mask = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return ({"value": R * R / t, "unit": "rays/s", "cores": torch.get_num_threads(), "cores_in_affinity_mask": mask, "kind": "port",
             "sample": f"image 0 of the bench's own batch ({R}x{R}x{S} hierarchical fp32 no_grad), median of {max(args.cpu_reps, 1)} after 1 "
                       f"warm-up ({t:.2f} s each), oracle/render_oracle.py (ATen op sequence of the reference CPU path)"}, ref)


def scaled_err(a, b):
    """max |a-b| / max(|b|, rms(b)) -- the metric of tests/conftest.py."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    floor = max(float(np.sqrt(np.mean(b * b))), 1e-30)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def survey_metric_pass(a, b):
    """Fraction of elements inside the gate SURVEY.md 8(d) wrote down: |a-b| <= 1e-4 * max(|b|, 1e-3)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.mean(np.abs(a - b) <= 1e-4 * np.maximum(np.abs(b), 1e-3)))


def check_against_oracle(args, gen, fvol, glob, cam, meta, draws, ref, timed_pixels, timed_depth):
    """Ties the timed run to a checked image.  (i) the last timed step drew its random tensors from CHECK_SEED: re-rendering
    with those tensors injected must reproduce its output bit for bit; (ii) image 0 of that render against the CPU oracle
    on identical inputs and draws: sample positions bit-exact, coarse rgb/sigma within 1e-4, resampling bins identical
    (>= 99 %); (iii) with the oracle's fine depths forced (sample positions identical downstream of the resampling, see
    DESIGN.md 4): fine rgb/sigma, pixels and depth within the gate, merge order identical."""
    B, R, S = args.batch, args.img_size, args.num_steps
    z = zin(gen, fvol, glob)
    aux = {}
    with torch.no_grad():
        px, dp = gen(z, cam, R, FOV, RAY_START, RAY_END, S, _rng=draws, _aux=aux, **meta)
    chk = {"timed_output_reproduced": bool(torch.equal(px, timed_pixels) and torch.equal(dp, timed_depth))}
    a0 = {k: v[0].cpu().numpy() for k, v in aux.items()}
    r0 = {k: v[0].numpy() for k, v in ref.aux.items()}
    chk["points_bit_exact"] = bool(np.array_equal(a0["coarse_points"], r0["coarse_points"]) and np.array_equal(a0["coarse_z"], r0["coarse_z"]))
    chk["rgb_sigma_err"] = max(scaled_err(a0["coarse_rgb_sigma"][..., :3], r0["coarse_rgb_sigma"][..., :3]),
                               scaled_err(a0["coarse_rgb_sigma"][..., 3], r0["coarse_rgb_sigma"][..., 3]))
    chk["rgb_sigma_survey_metric_pass"] = survey_metric_pass(a0["coarse_rgb_sigma"], r0["coarse_rgb_sigma"])
    chk["inds_equal"] = float(np.mean(a0["inds"] == r0["inds"]))
    chk["pixels_mean_abs_err_free_running"] = float(np.abs(px[0].cpu().numpy() - ref.pixels[0].numpy()).mean())
    forced = dict(draws)
    fz = aux["fine_z"].clone()
    fz[0] = ref.aux["fine_z"][0].to(fz.device)
    forced["fine_z"] = fz
    aux2 = {}
    with torch.no_grad():
        px2, dp2 = gen(z, cam, R, FOV, RAY_START, RAY_END, S, _rng=forced, _aux=aux2, **meta)
    chk["fine_points_bit_exact"] = bool(np.array_equal(aux2["fine_points"][0].cpu().numpy(), r0["fine_points"]))
    f_a, f_r = aux2["fine_rgb_sigma"][0].cpu().numpy(), r0["fine_rgb_sigma"]
    chk["fine_rgb_sigma_err"] = max(scaled_err(f_a[..., :3], f_r[..., :3]), scaled_err(f_a[..., 3], f_r[..., 3]))
    # merge order: identical wherever the depths are distinct.  At 128 samples per ray a few rays per image hold two samples of
    # EQUAL fp32 depth (birthday odds ~5e-4 per ray); torch.sort is not stable, so there -- and only there -- the reference's
    # permutation may order the twins either way (the sorted depths, which is what the compositing sees, are identical).
    si_a, si_r = aux2["sort_idx"][0].cpu().numpy().astype(np.int64), r0["sort_idx"]
    allz_r = np.concatenate([r0["fine_z"], r0["coarse_z"]], -1)
    zs_a, zs_r = np.take_along_axis(allz_r, si_a, -1), np.take_along_axis(allz_r, si_r, -1)
    diff = si_a != si_r
    tie = np.zeros_like(diff)
    tie[:, 1:] |= zs_r[:, 1:] == zs_r[:, :-1]
    tie[:, :-1] |= zs_r[:, :-1] == zs_r[:, 1:]
    chk["sorted_depths_bit_exact"] = bool(np.array_equal(zs_a, zs_r))
    chk["sort_idx_differs_only_at_equal_depths"] = bool(not (diff & ~tie).any())
    chk["rays_with_equal_depths"] = int(tie.any(-1).sum())
    chk["sort_idx_equal"] = bool(chk["sorted_depths_bit_exact"] and chk["sort_idx_differs_only_at_equal_depths"])
    # Rays on which the reference's image is discontinuous in its own densities: the last merged sample of a ray is composited with
    # delta = 1e10 (volumetric_rendering.py:30-33), so under relu its alpha jumps from 0 to 1 at sigma = 0; where the reference
    # density there is inside the rgb / sigma tolerance of zero the HIP value must equal ONE of the reference algorithm's two
    # branches (oracle/checks.py::knife_edge_branches) -- a positive check: those rays stay in the maximum.
    from oracle import checks as K
    aux_r0 = {k: v[:1] for k, v in ref.aux.items()}
    edge = K.knife_edge_rays(aux_r0, "relu")
    branches = None
    if edge is not None and bool(edge.any()):
        eps0 = draws["eps_final"].reshape(B, R * R, -1)[:1].cpu() if "eps_final" in draws else None
        branches = K.knife_edge_branches(aux_r0, R, FOV, args.noise, True, False, eps0)
    chk["pixels_err_forced"], chk["depth_err_forced"], chk["rays_at_a_density_zero_crossing"] = K.image_err_with_knife_edges(
        px2[:1], dp2[:1], ref.pixels[:1], ref.depth[:1], edge, branches)
    chk["pixels_survey_metric_pass"] = survey_metric_pass(px2[0].cpu().numpy(), ref.pixels[0].numpy())
    # "as accurate as the reference", measured: both fp32 results against the field in float64 at the same (bit-identical)
    # coarse sample positions, on a strided sample of ~260 k of the image's points
    sub = torch.arange(0, R * R * S, 4)
    exact = K.field_fp64(args.variant, {k: v.detach().cpu() for k, v in gen.siren.state_dict().items()}, fvol[:1].cpu(),
                         glob[:1].cpu() if gen.siren.spec.has_global else None, ref.aux["coarse_points"][:1].reshape(1, -1, 3)[:, sub])
    chk.update(K.accuracy_vs_fp64(aux["coarse_rgb_sigma"][:1].cpu().reshape(1, -1, 4)[:, sub],
                                  ref.aux["coarse_rgb_sigma"][:1].reshape(1, -1, 4)[:, sub], exact))
    tol = 1e-4 if args.precision == "fp32" else 2e-4
    chk["tolerance"] = tol
    chk["margin_guard"] = 0.9e-4       # rgb / sigma must stay below this: a kernel change that eats the margin fails the run
    chk["pass"] = bool(chk["timed_output_reproduced"] and chk["points_bit_exact"] and chk["fine_points_bit_exact"] and
                       chk["rgb_sigma_err"] < chk["margin_guard"] and chk["fine_rgb_sigma_err"] < chk["margin_guard"] and
                       chk["inds_equal"] > 0.99 and chk["sort_idx_equal"] and chk["pixels_err_forced"] < tol and
                       chk["depth_err_forced"] < tol and chk["rays_at_a_density_zero_crossing"] <= max(2, edge.numel() // 1000) and
                       chk["hip_vs_fp64"] <= 2 * chk["ref_vs_fp64"] + 1e-6 and chk["pixels_mean_abs_err_free_running"] < 2e-3)
    chk["what"] = ("image 0 of the last timed step (draws from torch.cuda seed %d) vs oracle/render_oracle.py on the same inputs and "
                   "draws; *_forced: oracle's fine depths injected" % CHECK_SEED)
    return chk


# ---------------------------------------------------------------------------------------------------------------------
def stub_main(args, rank, world):
    """`--device cpu`: the launcher / timing / aggregation path and the DDP training leg on gloo with stand-ins (the render has no
    CPU path): a sleep for the render step, StubGenerator inside the real GanTrainer.  Used by tests/test_distributed_cpu.py;
    never a benchmark -- the line says so."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    elapsed = timed_region(lambda i: time.sleep(0.005 * (1 + rank)), args.steps, args.warmup, dist if world > 1 else None)
    args.img_size_full, args.img_size, args.num_steps, args.hidden = args.img_size, 16, 4, 64
    ddp, _ = train_step_ddp(args, torch.device("cpu"), rank, world, dist, stub=True)
    if rank == 0:
        emit_result({"metric": "rays/sec at 128x128x64spp ShapeNetCar", "unit": "rays/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "value": whole_job_rays_per_s(world, args.batch, args.img_size_full, args.steps, elapsed),
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "stub: stand-in step on the CPU (launcher test), NOT a measurement",
                          "config": {"workload": "stub"}, "train_step_ddp": ddp})
    dist.barrier()
    dist.destroy_process_group()
    return 0


_RESULT_FD = None


def emit_result(res):
    line = (json.dumps(res) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


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
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (and with it the check of the timed image)")
    ap.add_argument("--no-fast-path", action="store_true", help="skip the secondary fp16x3 measurement")
    ap.add_argument("--no-train-step", action="store_true", help="skip the secondary forward + backward measurement")
    ap.add_argument("--no-gan-step", action="store_true", help="skip the GAN training step legs (train_step_ddp, gan_step)")
    ap.add_argument("--no-miopen-db", action="store_true", help="GAN step legs: MIOpen immediate mode instead of find mode over the shipped find results")
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"], help="cpu: launcher test with a stand-in step (no measurement)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_children(args.gpus, sys.argv[1:])          # nothing above has touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    # ONE line on stdout: libraries print there too (RCCL's version banner at communicator creation, MIOpen notices), so file
    # descriptor 1 is pointed at stderr for the run and the JSON line goes to the saved original descriptor
    sys.stdout.flush()
    global _RESULT_FD
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    if args.device == "cpu":
        return stub_main(args, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RCCL process group, also for a single rank (the training leg wraps the networks in DDP at every N).  The forward metric has
    # no data-path collective: there the group only carries the barrier and the max-over-ranks of the elapsed time.
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(minutes=30))

    import __graft_entry__ as ge
    import cnerf_amd                                   # (the package loads libcnerf_hip.so lazily: importable before the build)
    from cnerf_amd.training.latch import rank0_first
    rank0_first(rank, "build", ge.build)               # rank 0 compiles, the others poll a marker file (no pending collective)
    from cnerf_amd.generators import ImplicitGenerator3d
    from cnerf_amd.generators.generators import draw_rng

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

    # The draws of the LAST timed step are known in advance (torch.cuda seed CHECK_SEED): the CPU oracle renders the expected image
    # of that step from them AFTER all GPU legs (the GPU work of a run is contiguous; the oracle's ~20 s of host time come last).
    do_check = world == 1 and not args.no_cpu_baseline
    base = ref = draws = None
    if do_check:
        torch.cuda.manual_seed(CHECK_SEED)
        draws = draw_rng(B, R * R, S, True, args.noise, dev)

    evs = HipEvents()
    events = evs.create(4 * args.steps)
    last = {}

    def step(i=None):
        ev = events[4 * i:4 * i + 4] if i is not None else None
        cnerf_amd.ops.clear_pack_cache()                      # training changes the weights every step: re-pack
        if i == args.steps - 1:
            torch.cuda.manual_seed(CHECK_SEED)                # host-side generator state only: no device work, no sync
        with torch.no_grad():
            last["out"] = gen(zin(gen, fvol, glob), cam, R, FOV, RAY_START, RAY_END, S, _field_events=ev, **meta)

    elapsed = timed_region(step, args.steps, args.warmup, dist if world > 1 else None, torch.cuda.synchronize, dev)
    out = last["out"]
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all()

    # dominant kernel: the field kernel, two launches per step (coarse, fine), same work each
    kern_ms = []
    for i in range(args.steps):
        e = events[4 * i:4 * i + 4]
        kern_ms += [evs.elapsed_ms(e[0], e[1]), evs.elapsed_ms(e[2], e[3])]
    avg_ms = float(np.mean(kern_ms))
    n_layers = n_matrices(gen.siren)
    flops_per_launch = 2.0 * macs_per_point(32, args.hidden, n_layers, gen.siren) * B * R * R * S
    pfilm = gen.siren.spec.layers[0] == "pfilm"
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
    split = args.precision == "fp16x3"
    if split:      # priced per issued flop: three fp16 MFMAs per fp32 product, against the dense fp16 MFMA peak
        roof = {"kernel": ("pw::field_pw16_kernel<8> (sample + trilinear lookup + mapping MLP + per-point FiLM-SIREN, fp16x3 split, fp32 accumulate)" if pfilm else
                           "field_h3_kernel<8> (sample + trilinear lookup + FiLM-SIREN MLP, fp16x3 split, fp32 accumulate)"),
                "bound": "mfma", "achieved": 3 * achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": 3 * achieved / PEAK_F16_MFMA_TFLOPS, "traffic": None, "algorithmic_tflops": achieved}
        pk = pmc_bytes("void cnerf::h3::field_h3_kernel")
    else:
        roof = {"kernel": ("field_pw_kernel<8> (sample + trilinear lookup + mapping MLP + per-point FiLM-SIREN, fp32 MFMA)" if pfilm else
                           "field_tile_kernel<8> (sample + trilinear lookup + FiLM-SIREN MLP, fp32 MFMA)"),
                "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None}
        pk = pmc_bytes("void cnerf::field_tile_kernel")
    # HBM bytes cannot be counted from inside this process: `traffic` is the rocprofv3 PMC measurement of the same kernel
    # on the same workload shape (profiles/pmc_traffic.json), per point, scaled to this launch's points.
    if pk and (R, S, args.volume) == (128, 64, 64):
        roof["traffic"] = pk["bytes_per_point"] * B * R * R * S
        roof["traffic_source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes ({os.path.relpath(PMC_FILE, ROOT)}), per point x points per launch"
    roof.update({"avg_launch_ms": avg_ms, "launches": len(kern_ms), "flops_per_launch": flops_per_launch,
                 "share_of_step": 2 * avg_ms / (elapsed / args.steps * 1e3)})

    rc = 0
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
            res["fp16_single_pass_path"] = fast_path(args, gen, fvol, glob, cam, meta, evs, precision="fp16")
        if world == 1 and args.precision == "fp32" and not args.no_fast_path:
            # the same fp32 step with the four draws generated inside the kernels (no torch RNG kernels, no random tensors)
            gen.rng_mode = "philox"
            try:
                with torch.no_grad():
                    for _ in range(2):
                        step(None)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        step(None)
                    torch.cuda.synchronize()
                    dtp = (time.perf_counter() - t0) / 5
            finally:
                gen.rng_mode = "torch"
            res["philox_rng_path"] = {"value": B * R * R / dtp, "unit": "rays/s", "ms_per_step": dtp * 1e3, "dtype": "f32",
                                      "note": "in-kernel Philox4x32-10 draws (cnerf_cfg.philox) instead of torch.rand / randn tensors"}
        if world == 1 and not args.no_train_step:
            res["train_step"] = train_step_timing(args, gen, fvol, glob, cam, evs)
    if not args.no_gan_step:          # every rank: the networks are wrapped in DDP over the job's process group
        ddp, legs = train_step_ddp(args, dev, rank, world, dist)
        if rank == 0:
            res["train_step_ddp"] = ddp
            if legs:
                res["gan_step"] = legs
    if rank == 0:
        if do_check:
            base, ref = cpu_baseline(args, gen_cpu, fvol, glob, cam, draws)
            res["check"] = check_against_oracle(args, gen, fvol, glob, cam, meta, draws, ref, out[0], out[1])
            res["cpu_baseline"] = base
            if not res["check"]["pass"]:
                rc = 1
        emit_result(res)
        if rc:
            sys.stderr.write("bench.py: the timed image FAILED the oracle check: " + json.dumps(res["check"]) + "\n")
    dist.barrier()
    dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
