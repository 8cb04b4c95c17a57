#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of field_tile_kernel (library built with -DCNERF_STAMPS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cnerf_amd
from cnerf_amd import ops, _lib as L
from cnerf_amd.generators import ImplicitGenerator3d
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, R, S = 2, 128, 64
gen = ImplicitGenerator3d("SHORTSIREN_FG", 256, 32, 4, 256).to(dev); gen.set_device(dev)
net = gen.siren
net.precision = os.environ.get("CNERF_PRECISION", "fp32")
fvol, glob = torch.randn(B, 32, 64, 64, 64, device=dev), torch.randn(B, 256, device=dev)
cam = torch.eye(4, device=dev).unsqueeze(0).repeat(B, 1, 1); cam[:, 2, 3] = -1.0
freq, phase = net.film(glob)
fcl = ops.channel_last(fvol)
cfg = ops.make_cfg(net, B, [fcl], R, S, 49.13, 0.25, 1.95, 0.0, False, True, False, "relu")   # non-hierarchical: one field launch
packed = ops.pack_field(net, cfg)
_, _, wsb = ops.sizes(cfg)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
px = torch.empty(B, 3, R, R, device=dev); dp = torch.empty(B, R, R, device=dev)
u = torch.rand(B, R * R, S, device=dev)
stamps = torch.zeros(64, dtype=torch.float32, device=dev)   # handed over in the (unused, non-hierarchical) cdf slot
r = L.Rng(); r.u_strat = u.data_ptr()
aux = L.Aux(); aux.cdf = stamps.data_ptr()
if os.environ.get("CNERF_STORE"):        # the activation-keeping forward of the half-precision backward (fp16x3 / fp16 precisions)
    kept = ops.resident_act16(net, [fcl], B, R, S, False, dev)
    aux.act16[0].feat, aux.act16[0].h, aux.act16[0].c = (t.data_ptr() for t in kept[0])
for it in range(2):
    stamps.zero_()
    vs = ops.volumes_struct([fcl])
    L.check(L.lib().cnerf_render_forward(C.byref(cfg), C.byref(vs), L.ptr(packed), L.ptr(freq.detach()), L.ptr(phase.detach()), L.ptr(cam),
                                         C.byref(r), L.ptr(px), L.ptr(dp), C.byref(aux), L.ptr(ws), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "render")
    torch.cuda.synchronize()
st = stamps[:16].view(torch.int64)[:8].cpu().tolist()
tiles = B * R * R * S // 32
if net.precision == "fp16x3":
    names = ["loop", "position (issue)", "lookup+layer0 MFMA", "layer0 epilogue", "hidden(all)", "head+store"]
else:
    names = ["loop/store", "position+lookup+layer0 MFMA", "layer0 epilogue", "hidden(all)", "head"]
tot = sum(st[:len(names)])
for n, v in zip(names, st[:len(names)]):
    print(f"{n:18s} {v / tiles:12.0f} ticks/tile  {100.0 * v / tot:5.1f}%")
print("total ticks/tile", tot / tiles, "(s_memtime ticks at 100 MHz => x ~23 for shader cycles)")
