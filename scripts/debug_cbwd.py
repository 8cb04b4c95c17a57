#!/usr/bin/env python3
"""Stage check: cnerf_merge_composite_backward vs autograd through the oracle's merge + composite."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import Golden, scaled_err
from oracle import render_oracle as O
import cnerf_amd
from cnerf_amd import ops, _lib as L
from test_gpu_parity import make_generator, G
dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    g = Golden(name); m = g.meta
    B, P, S = g["coarse_z"].shape; R = m["R"]
    T = lambda k: torch.from_numpy(np.asarray(g[k]))
    c_rs = T("coarse_rgb_sigma").clone().requires_grad_(True); f_rs = T("fine_rgb_sigma").clone().requires_grad_(True)
    c_z, f_z = T("coarse_z"), T("fine_z")
    eps = T("eps_final") if m["noise"] != 0 else None
    all_out, all_z, idx = O.merge_by_depth(f_rs, c_rs, f_z, c_z)
    rgb, dist, w = O.composite(all_out, all_z, eps, m["noise"], m["clamp"], m["white_back"], m["last_back"])
    dirs = O.camera_ray_dirs(R, m["fov"])
    pixels = rgb.reshape(B, R, R, 3).permute(0, 3, 1, 2).contiguous() * 2 - 1
    depth = (dirs[:, 2].reshape(1, P) * dist).reshape(B, R, R)
    loss = pixels.square().mean() + depth.mean()
    gc_ref, gf_ref = torch.autograd.grad(loss, [c_rs, f_rs])
    gp = (2 * pixels / pixels.numel()).detach(); gd = torch.full_like(depth, 1.0 / depth.numel())
    gen = make_generator(g, dev)
    cfg = ops.make_cfg(gen.siren, B, m["V"], R, S, m["fov"], m["ray_start"], m["ray_end"], m["noise"], True, m["white_back"], m["last_back"], m["clamp"])
    gc = torch.empty(B, P, S, 4, device=dev); gf = torch.empty(B, P, S, 4, device=dev)
    d = lambda t: t.detach().to(dev).contiguous()
    keep = [d(c_rs), d(c_z), d(f_rs), d(f_z), d(eps) if eps is not None else None, d(gp), d(gd)]
    L.check(L.lib().cnerf_merge_composite_backward(C.byref(cfg), *[L.ptr(t) for t in keep], L.ptr(gc), L.ptr(gf), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cbwd")
    torch.cuda.synchronize()
    for nm, a, b in (("coarse", gc, gc_ref), ("fine", gf, gf_ref)):
        a = a.cpu().numpy(); b = b.numpy()
        print(name, nm, "rgb", scaled_err(a[..., :3], b[..., :3]), "sigma", scaled_err(a[..., 3], b[..., 3]), "max|ref sigma grad|", np.abs(b[..., 3]).max())
        dd = np.abs(a[..., 3] - b[..., 3]); i = np.unravel_index(np.argmax(dd), dd.shape)
        print("    worst sigma idx", i, a[..., 3][i], b[..., 3][i])
