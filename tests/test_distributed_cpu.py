"""The N > 1 path of bench.py on the CPU: two gloo ranks, the same barrier / max-over-ranks / whole-job aggregation
code the GPU run uses (bench.timed_region, bench.whole_job_rays_per_s), with a stand-in step (the render itself needs a
GPU and has no CPU fallback).  Also checks that each rank's synthetic inputs differ (independent images per rank)."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    calls = []

    def step(i):
        calls.append(i)
        time.sleep(0.01 * (1 + 2 * rank))          # rank 1 is three times slower than rank 0

    elapsed = bench.timed_region(step, steps=5, warmup=2, dist=dist)
    fvol, glob, cam = bench.synthetic_inputs(1, 4, 8, "cpu", seed=rank)
    q.put((rank, elapsed, calls, float(fvol.sum()), bench.whole_job_rays_per_s(world, 8, 128, 5, elapsed)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_timing_and_aggregation():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, e0, c0, s0, v0), (r1, e1, c1, s1, v1) = res
    assert c0 == [None, None, 0, 1, 2, 3, 4] and c1 == c0        # W untimed + exactly K timed steps on every rank
    assert abs(e0 - e1) < 1e-9                                    # every rank reports the max over ranks
    assert e0 >= 5 * 0.03 * 0.9                                   # ... which is the slow rank's time
    assert s0 != s1                                               # each rank renders its own images
    assert abs(v0 - 2 * 8 * 128 * 128 * 5 / e0) < 1e-6 * v0       # whole-job rays/s = all ranks' rays / that time


def test_single_process_region():
    import bench
    n = []
    t = bench.timed_region(lambda i: n.append(i), steps=3, warmup=1)
    assert n == [None, 0, 1, 2] and t >= 0


@pytest.mark.timeout(180)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the form the driver's N=1 command takes with N
    changed): the parent starts two ranks through torch.distributed.run as CHILD processes, relays rank 0's single JSON
    line and their exit code.  `--device cpu` swaps the render for a stand-in step on gloo (no GPU here); everything else
    -- launcher, rendezvous on 127.0.0.1, barrier, max-over-ranks, whole-job aggregation -- is the code the GPU run uses."""
    import json
    import subprocess
    import sys
    import numpy as np
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--device", "cpu"],
                       capture_output=True, text=True, env=env, timeout=170)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["ms_per_step"] >= 10.0 * 0.9                  # the slower rank (rank 1: 10 ms per step) sets the time
    assert abs(res["value"] - 2 * 8 * 128 * 128 / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]
    assert "stub" in res["data"]
    # the DDP training leg (gloo here, RCCL on the GPU box): the real GanTrainer around a stand-in generator, three DDP wrappers
    ddp = res["train_step_ddp"]
    assert ddp["n_ranks"] == 2 and ddp["backend"] == "gloo"
    ar = ddp["allreduce_per_optimizer_step"]
    assert set(ar) == {"generator", "encoder", "discriminator"}
    for name, c in ar.items():                               # ONE all-reduce round per optimizer step and network ...
        assert c["rounds"] == 1.0 and c["calls"] >= 1.0, (name, c)
        assert c["bytes"] == 4 * ddp["parameters"][name], (name, c)      # ... carrying every fp32 gradient exactly once
    assert ddp["allreduce_bytes_per_step"] == sum(c["bytes"] for c in ar.values())
    assert set(ddp["allreduce_standalone"]) == set(ar) and all(v["ms"] > 0 for v in ddp["allreduce_standalone"].values())
    assert {"d_render", "d_disc_r1_opt", "g_render_fwd", "g_backward", "g_clip_opt", "encoder_fwd"} <= set(ddp["phases_ms_rank0"])
    assert all(np.isfinite(v) for v in ddp["losses"].values())


def test_bench_launcher_propagates_failure():
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "0", "--device", "cpu"],
                       capture_output=True, text=True, env=env, timeout=170)
    assert p.returncode != 0 and not p.stdout.strip()        # steps = 0 divides by zero in every rank: no line, non-zero exit


# ---------------------------------------------------------------------------------------------------------------------
# multi-rank training correctness: GanTrainer(ddp=True) on two gloo ranks.  The render has no CPU path, so the
# generator is a small differentiable stand-in with the generator's call signature; encoder, discriminator, the step logic,
# the three DDP wrappers and the no_sync() accumulation are the real ones.
# ---------------------------------------------------------------------------------------------------------------------
class _StandInGenerator(torch.nn.Module):
    """(feature volume, global feature), cameras, img_size, ... -> (pixels (B,3,R,R), depth (B,R,R)): differentiable w.r.t. its
    own parameters, the feature volume and the global feature, like the render."""

    def __init__(self, z_dim):
        super().__init__()
        self.lin = torch.nn.Linear(z_dim + 32, 3 * 4 * 4)
        self.step = 0
        self.epoch = 0

    def forward(self, z, cam2worlds, img_size, *args, **kwargs):
        fvol, glob = z
        h = torch.cat([glob, fvol.mean(dim=(2, 3, 4))], -1)
        px = torch.tanh(self.lin(h)).reshape(-1, 3, 4, 4) + 0.01 * cam2worlds[:, :3, 3].reshape(-1, 3, 1, 1)
        px = torch.nn.functional.interpolate(px, size=(img_size, img_size), mode="bilinear", align_corners=False)
        return px, px.mean(1)


def _train_worker(rank, world, port, q, batch_split):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import cnerf_amd  # noqa: F401
    from cnerf_amd.training import GanTrainer, default_metadata, UNet3D
    from cnerf_amd.training.gan_step import synthetic_sample
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    torch.manual_seed(0)                                   # identical initial parameters on every rank
    md = default_metadata(img_size=16, num_steps=4, batch_size=4, batch_split=batch_split, hidden_dim=64)
    md["unet"].update(f_maps=8, num_levels=2)
    gen = _StandInGenerator(16)
    tr = GanTrainer(md, torch.device("cpu"), ddp=world > 1, modules={"generator": gen})
    counts = {"g": 0, "e": 0, "d": 0}
    if world > 1:
        def hook_for(name):
            def hook(state, bucket):
                counts[name] += 1
                return default_hooks.allreduce_hook(state, bucket)
            return hook
        tr.generator_ddp.register_comm_hook(None, hook_for("g"))
        tr.encoder_ddp.register_comm_hook(None, hook_for("e"))
        tr.discriminator_ddp.register_comm_hook(None, hook_for("d"))
    # the job's batch: 8 images; rank r trains on images [4r, 4r + 4) -- a single process takes the first four only
    g = torch.Generator().manual_seed(99)
    np.random.seed(99)
    full = synthetic_sample(8, 16, 8, "cpu", g)
    sample = {k: v[4 * rank:4 * rank + 4] for k, v in full.items()}
    np.random.seed(1234)                                   # (the D step draws its cameras from NumPy)
    tr.step(sample)
    after_one = dict(counts)
    np.random.seed(1234)
    tr.step(sample)
    flat = torch.cat([p.detach().flatten() for m in (tr.generator, tr.encoder, tr.discriminator) for p in m.parameters()])
    q.put((rank, flat[::97].clone().numpy(), float(flat.double().sum()), after_one, dict(counts), tr.last))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_training(world, batch_split):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, batch_split)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=280) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(600)
def test_two_rank_gan_step_parameters_and_allreduce_count():
    """(i) after two GAN steps both ranks hold identical parameters (DDP averaged every gradient); (ii) they differ from a
    single process trained on rank 0's half of the data; (iii) with batch_split = 4 the generator and encoder all-reduce
    once per optimizer step -- their communication hooks fire exactly as often as with batch_split = 1, i.e. for the last
    accumulation chunk only (the reference all-reduces on every chunk: utils.py:638-711) -- and the discriminator once."""
    import numpy as np
    two = _run_training(2, batch_split=4)
    (r0, s0, sum0, c1_0, c2_0, last0), (r1, s1, sum1, c1_1, c2_1, last1) = two
    assert np.array_equal(s0, s1) and sum0 == sum1
    for k in ("d_loss", "g_loss", "photo_loss"):
        assert np.isfinite(last0[k]) and np.isfinite(last1[k])
    assert last0["photo_loss"] != last1["photo_loss"]            # each rank saw its own images
    one = _run_training(1, batch_split=4)[0]
    assert not np.array_equal(one[1], s0)
    unsplit = _run_training(2, batch_split=1)
    n1 = unsplit[0][3]                                            # hook calls of ONE step without accumulation = one round of buckets
    assert all(v > 0 for v in n1.values())
    assert c1_0 == n1 and c1_1 == n1, (c1_0, n1)                 # four chunks, one all-reduce round
    assert c2_0 == {k: 2 * v for k, v in n1.items()}             # and again in the second step


@pytest.mark.parametrize("batch_split", [1, 2])
def test_keeping_the_encoder_output_between_the_d_and_g_pass_changes_nothing(batch_split):
    """GanTrainer.step with metadata["reuse_encoder_output"] (one encoder forward per chunk, shared by the D step's no-grad renders
    and the G step's backward) must end a step with exactly the parameters of the reference's schedule (encoder re-evaluated in the G
    step): same initial state, same batch, same NumPy camera draws."""
    import numpy as np
    import cnerf_amd  # noqa: F401
    from cnerf_amd.training import GanTrainer, default_metadata
    from cnerf_amd.training.gan_step import synthetic_sample
    out = []
    for reuse in (True, False):
        torch.manual_seed(0)
        md = default_metadata(img_size=16, num_steps=4, batch_size=4, batch_split=batch_split, hidden_dim=64)
        md["unet"].update(f_maps=8, num_levels=2)
        md["reuse_encoder_output"] = reuse
        tr = GanTrainer(md, torch.device("cpu"), modules={"generator": _StandInGenerator(16)})
        np.random.seed(99)
        sample = synthetic_sample(4, 16, 8, "cpu", torch.Generator().manual_seed(99))
        for _ in range(2):
            np.random.seed(1234)
            tr.step(sample)
        out.append(torch.cat([p.detach().flatten() for m in (tr.generator, tr.encoder, tr.discriminator) for p in m.parameters()]))
    assert torch.allclose(out[0], out[1], rtol=0, atol=1e-7)


# ---------------------------------------------------------------------------------------------------------------------
# rank-0-first start-up work must not be waited for inside a collective (the MIOpen search on rank 0 takes up to 8 minutes,
# the default watchdog of the process group is 10)
# ---------------------------------------------------------------------------------------------------------------------
def _latch_worker(rank, world, port, q, use_latch, tmp):
    import datetime
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=3))
    import cnerf_amd  # noqa: F401
    from cnerf_amd.training.latch import rank0_first
    slow = lambda: time.sleep(7.0)                           # rank 0's start-up work outlasts the group's timeout
    ok = True
    try:
        if use_latch:
            rank0_first(rank, "slow", slow, tmp)             # train.py's scheme: the others poll a marker file ...
            dist.barrier()                                   # ... and the collective that follows is entered by all ranks together
        else:
            if rank == 0:
                slow()
            dist.barrier()                                   # the round-2 scheme: rank 1 waits 7 s inside a 3-s collective
    except Exception:                                        # noqa: BLE001  (gloo raises a RuntimeError / DistBackendError on timeout)
        ok = False
    q.put((rank, ok))
    try:
        dist.destroy_process_group()
    except Exception:                                        # noqa: BLE001
        pass


@pytest.mark.timeout(180)
def test_rank0_first_work_outlasting_the_group_timeout(tmp_path):
    """A marker-file latch (training/latch.py) lets rank 0 work for longer than the process group's timeout while the other ranks
    wait; the same wait inside a barrier aborts (which is what train.py's start-up did at 72-80 % of NCCL's 10 minutes)."""
    ctx = mp.get_context("spawn")
    for use_latch, want in ((True, True), (False, False)):
        port, q = _free_port(), ctx.Queue()
        procs = [ctx.Process(target=_latch_worker, args=(r, 2, port, q, use_latch, str(tmp_path))) for r in range(2)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=90) for _ in range(2))
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
        if want:
            assert res == {0: True, 1: True}
        else:
            assert not res[1]                                # the waiting rank's collective timed out
