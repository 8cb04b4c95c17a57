#!/usr/bin/env python3
"""Is a training step host-bound?  Wall time of forward + backward against the GPU time between two events around the same calls,
and the time spent inside ops.backward_chunk / workspace allocation.   python scripts/host_gap_probe.py [variant] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnerf_amd
from cnerf_amd import ops
from cnerf_amd.generators import ImplicitGenerator3d
from cnerf_amd.generators.siren import FIELD_SPECS
V = sys.argv[1] if len(sys.argv) > 1 else "TALLSIREN_dRes"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0"); torch.manual_seed(0)
gen = ImplicitGenerator3d(V, 256 if FIELD_SPECS[V].has_global else 32, 32, 4, 256).to(dev); gen.set_device(dev)
gen.siren.precision, gen.siren.backward_precision = "fp16x3", "fp16"
fvol = torch.randn(B, 32, 64, 64, 64, device=dev, requires_grad=True); glob = torch.randn(B, 256, device=dev, requires_grad=True)
cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1); cam[:, 2, 3] = -1.0
t_chunk = [0.0]
orig = ops.backward_chunk
def timed_chunk(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); t_chunk[0] += time.perf_counter() - t0; return r
ops.backward_chunk = timed_chunk
t_free = [0.0, 0]
orig_free = ops.free_device_bytes
def timed_free(d):
    t0 = time.perf_counter(); r = orig_free(d); t_free[0] += time.perf_counter() - t0; t_free[1] += 1; return r
ops.free_device_bytes = timed_free
import gc
gc_t = [0.0]
def gc_cb(phase, info):
    if phase == "start": gc_cb.t0 = time.perf_counter()
    else: gc_t[0] += time.perf_counter() - gc_cb.t0
gc.callbacks.append(gc_cb)
for it in range(8):
    torch.cuda.synchronize(); t_chunk[0] = 0.0; t_free[0] = 0.0; t_free[1] = 0; gc_t[0] = 0.0
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter(); e0.record()
    px, dp = gen((fvol, glob) if gen.siren.spec.has_global else fvol, cam, 128, 49.134342641202636, 0.25, 1.95, 64, True, clamp_mode="relu", nerf_noise=1.0, white_back=True)
    e1.record(); t1 = time.perf_counter()
    (px.square().mean() + dp.mean()).backward()
    e2.record(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"step {it}: wall {1e3*(t3-t0):.1f} ms | GPU fwd {e0.elapsed_time(e1):.1f} bwd {e1.elapsed_time(e2):.1f} | host: fwd call returned after {1e3*(t1-t0):.1f}, bwd call after {1e3*(t2-t1):.1f} "
          f"(of which backward_chunk {1e3*t_chunk[0]:.2f}) | mem alloc retries {torch.cuda.memory_stats()['num_alloc_retries']} "
          f"cudaMalloc calls {torch.cuda.memory_stats()['segment.all.allocated']} | free_device_bytes {t_free[1]} calls {1e3*t_free[0]:.2f} ms | gc {1e3*gc_t[0]:.2f} ms", flush=True)
