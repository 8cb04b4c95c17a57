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


def test_bench_launcher_propagates_failure():
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "0", "--device", "cpu"],
                       capture_output=True, text=True, env=env, timeout=170)
    assert p.returncode != 0 and not p.stdout.strip()        # steps = 0 divides by zero in every rank: no line, non-zero exit
