"""Thin PyTorch-side wrappers over the C ABI (include/cnerf.h).  PyTorch only provides device memory, the current
stream and autograd plumbing here; every computation below happens in libcnerf_hip.so."""
import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib as L

_pack_cache: Dict[int, tuple] = {}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def make_cfg(net, B, V, R=1, S=2, fov=30.0, ray_start=0.0, ray_end=1.0, noise_std=0.0, hierarchical=False,
             white_back=False, last_back=False, clamp_mode="relu"):
    """cnerf_cfg for a field network `net` (generators.siren.FieldNetwork)."""
    cfg = L.Cfg()
    cfg.B, cfg.R, cfg.S, cfg.V = int(B), int(R), int(S), int(V)
    cfg.C, cfg.H = int(net.input_dim), int(net.hidden_dim)
    kinds = [L.LAYER_CODE[k] for k in net.spec.layers]
    cfg.L = len(kinds)
    for i, k in enumerate(kinds):
        cfg.layer_kind[i] = k
    cfg.ray_start, cfg.ray_end, cfg.voxel_length = float(ray_start), float(ray_end), 1.2
    cfg.noise_std = float(noise_std)
    cfg.fov_deg = float(fov)
    if clamp_mode not in ("relu", "softplus"):
        raise TypeError("Need to choose clamp mode")   # the reference raises a str here, i.e. a TypeError
    flags = 0
    flags |= L.F_HIERARCHICAL if hierarchical else 0
    flags |= L.F_WHITE_BACK if white_back else 0
    flags |= L.F_LAST_BACK if last_back else 0
    flags |= L.F_SOFTPLUS if clamp_mode == "softplus" else 0
    flags |= L.F_SIGMOID_RGB if net.spec.sigmoid_rgb else 0
    cfg.flags = flags
    return cfg


def sizes(cfg, render=True):
    a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    L.check(L.lib().cnerf_workspace_bytes(C.byref(cfg), C.byref(a), C.byref(b), C.byref(c) if render else None),
            "cnerf_workspace_bytes")
    return a.value, b.value, c.value


def pack_field(net, cfg):
    """Packed MFMA-order weights of `net` (device tensor), re-packed only when a parameter changed."""
    params = [_f32(p.detach()) for p in net.field_params()]
    key = tuple((p.data_ptr(), p._version) for p in net.field_params())
    hit = _pack_cache.get(id(net))
    if hit is not None and hit[0] == key:
        return hit[1]
    nbytes, _, _ = sizes(cfg, render=False)
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=params[0].device)
    fp = L.FieldParams()
    it = iter(params)
    for i, kind in enumerate(net.spec.layers):
        fp.w[i], fp.b[i] = next(it).data_ptr(), next(it).data_ptr()
        if kind == "res":
            fp.w2[i], fp.b2[i] = next(it).data_ptr(), next(it).data_ptr()
    fp.w_final, fp.b_final = next(it).data_ptr(), next(it).data_ptr()
    L.check(L.lib().cnerf_pack_field(C.byref(cfg), C.byref(fp), L.ptr(packed), _stream()), "cnerf_pack_field")
    packed._keepalive = params
    _pack_cache[id(net)] = (key, packed)
    return packed


def channel_last(fvol):
    """(B,C,V,V,V) -> (B,V,V,V,C) on the GPU."""
    fvol = _f32(fvol)
    B, Cc, V = fvol.shape[0], fvol.shape[1], fvol.shape[-1]
    if fvol.shape[2:] != (V, V, V):
        raise L.CnerfError("feature volume must be cubic (B,C,V,V,V)")
    out = torch.empty((B, V, V, V, Cc), dtype=torch.float32, device=fvol.device)
    L.check(L.lib().cnerf_fvol_channel_last(B, Cc, V, L.ptr(fvol), L.ptr(out), _stream()), "cnerf_fvol_channel_last")
    return out


def channel_first(fvol_cl):
    fvol_cl = _f32(fvol_cl)
    B, V, Cc = fvol_cl.shape[0], fvol_cl.shape[1], fvol_cl.shape[-1]
    out = torch.empty((B, Cc, V, V, V), dtype=torch.float32, device=fvol_cl.device)
    L.check(L.lib().cnerf_fvol_channel_first(B, Cc, V, L.ptr(fvol_cl), L.ptr(out), _stream()), "cnerf_fvol_channel_first")
    return out


def gather_features(net, fvol_cl, points):
    points = _f32(points)
    B, n = points.shape[0], points.shape[1]
    cfg = make_cfg(net, B, fvol_cl.shape[1])
    out = torch.empty((B, n, fvol_cl.shape[-1]), dtype=torch.float32, device=points.device)
    L.check(L.lib().cnerf_gather_features(C.byref(cfg), L.ptr(fvol_cl), L.ptr(points), n, L.ptr(out), _stream()),
            "cnerf_gather_features")
    return out


def field_forward(net, fvol, freq, phase, points, fvol_is_channel_last=False):
    """rgb_sigma (B,n,4) of `net` at explicit world points (B,n,3)."""
    points = _f32(points)
    B, n = points.shape[0], points.shape[1]
    fvol_cl = _f32(fvol) if fvol_is_channel_last else channel_last(fvol)
    cfg = make_cfg(net, B, fvol_cl.shape[1])
    packed = pack_field(net, cfg)
    out = torch.empty((B, n, 4), dtype=torch.float32, device=points.device)
    L.check(L.lib().cnerf_field_forward(C.byref(cfg), L.ptr(fvol_cl), L.ptr(packed), L.ptr(_f32(freq)), L.ptr(_f32(phase)),
                                        L.ptr(points), n, L.ptr(out), _stream()), "cnerf_field_forward")
    return out


def composite(rgb_sigma, z, eps=None, noise_std=0.0, clamp_mode="relu", white_back=False, last_back=False):
    """fancy_integration on (rays,n,4)/(rays,n) tensors -> rgb (rays,3), dist (rays), weights (rays,n)."""
    rgb_sigma, z, eps = _f32(rgb_sigma), _f32(z), _f32(eps)
    rays, n = z.shape
    cfg = L.Cfg()
    cfg.noise_std = float(noise_std)
    if clamp_mode not in ("relu", "softplus"):
        raise TypeError("Need to choose clamp mode")
    cfg.flags = (L.F_WHITE_BACK if white_back else 0) | (L.F_LAST_BACK if last_back else 0) | \
                (L.F_SOFTPLUS if clamp_mode == "softplus" else 0)
    dev = z.device
    rgb = torch.empty((rays, 3), dtype=torch.float32, device=dev)
    dist = torch.empty((rays,), dtype=torch.float32, device=dev)
    w = torch.empty((rays, n), dtype=torch.float32, device=dev)
    L.check(L.lib().cnerf_composite(C.byref(cfg), rays, n, L.ptr(rgb_sigma), L.ptr(z), L.ptr(eps), L.ptr(rgb), L.ptr(dist),
                                    L.ptr(w), _stream()), "cnerf_composite")
    return rgb, dist, w


def resample(z, weights, u):
    """Inverse-CDF depths: z, weights, u (rays,S) -> fine_z (rays,S), inds (rays,S) int32, cdf (rays,S-1)."""
    z, weights, u = _f32(z), _f32(weights), _f32(u)
    rays, S = z.shape
    dev = z.device
    fine = torch.empty((rays, S), dtype=torch.float32, device=dev)
    inds = torch.empty((rays, S), dtype=torch.int32, device=dev)
    cdf = torch.empty((rays, S - 1), dtype=torch.float32, device=dev)
    L.check(L.lib().cnerf_resample(rays, S, L.ptr(z), L.ptr(weights), L.ptr(u), L.ptr(fine), L.ptr(inds), L.ptr(cdf),
                                   _stream()), "cnerf_resample")
    return fine, inds, cdf


AUX_SHAPES = {
    "coarse_points": lambda B, P, S, n: ((B, P, S, 3), torch.float32),
    "coarse_z": lambda B, P, S, n: ((B, P, S), torch.float32),
    "coarse_rgb_sigma": lambda B, P, S, n: ((B, P, S, 4), torch.float32),
    "coarse_weights": lambda B, P, S, n: ((B, P, S), torch.float32),
    "cdf": lambda B, P, S, n: ((B, P, S - 1), torch.float32),
    "inds": lambda B, P, S, n: ((B, P, S), torch.int32),
    "fine_z": lambda B, P, S, n: ((B, P, S), torch.float32),
    "fine_rgb_sigma": lambda B, P, S, n: ((B, P, S, 4), torch.float32),
    "sort_idx": lambda B, P, S, n: ((B, P, 2 * S), torch.int32),
    "final_weights": lambda B, P, S, n: ((B, P, n), torch.float32),
    "fine_points": lambda B, P, S, n: ((B, P, S, 3), torch.float32),
}
HIER_ONLY = {"coarse_weights", "cdf", "inds", "fine_z", "fine_rgb_sigma", "sort_idx", "fine_points"}


def render_forward(net, fvol, freq, phase, cam2world, img_size, fov, ray_start, ray_end, num_steps, hierarchical,
                   clamp_mode, noise_std, white_back=False, last_back=False, rng: Optional[dict] = None,
                   want_aux=False, fvol_is_channel_last=False, field_events=None):
    """ImplicitGenerator3d.forward on the GPU.  rng: dict with u_strat / eps_coarse / u_fine / eps_final tensors.
    field_events: optional 4 hipEvent_t handles (ints) recorded around the two field-kernel launches."""
    cam2world = _f32(cam2world)
    B, R, S = cam2world.shape[0], int(img_size), int(num_steps)
    P = R * R
    n = 2 * S if hierarchical else S
    dev = cam2world.device
    fvol_cl = _f32(fvol) if fvol_is_channel_last else channel_last(fvol)
    cfg = make_cfg(net, B, fvol_cl.shape[1], R, S, fov, ray_start, ray_end, noise_std, hierarchical, white_back,
                   last_back, clamp_mode)
    packed = pack_field(net, cfg)
    _, _, ws_bytes = sizes(cfg)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    pixels = torch.empty((B, 3, R, R), dtype=torch.float32, device=dev)
    depth = torch.empty((B, R, R), dtype=torch.float32, device=dev)
    rng = rng or {}
    keep = [_f32(rng.get(k)) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final", "fine_z")]
    r = L.Rng()
    r.u_strat, r.eps_coarse, r.u_fine, r.eps_final, r.fine_z = [None if t is None else t.data_ptr() for t in keep]
    for t in keep:
        L.ptr(t)   # validates device / contiguity
    aux_t, aux_s = {}, None
    if want_aux:
        aux_s = L.Aux()
        for k in L.AUX_FIELDS:
            if not hierarchical and k in HIER_ONLY:
                continue
            shape, dt = AUX_SHAPES[k](B, P, S, n)
            aux_t[k] = torch.empty(shape, dtype=dt, device=dev)
            setattr(aux_s, k, aux_t[k].data_ptr())
    if field_events is not None:
        aux_s = aux_s if aux_s is not None else L.Aux()
        for i, ev in enumerate(field_events):
            aux_s.field_events[i] = ev
    L.check(L.lib().cnerf_render_forward(C.byref(cfg), L.ptr(fvol_cl), L.ptr(packed), L.ptr(_f32(freq)), L.ptr(_f32(phase)),
                                         L.ptr(cam2world), C.byref(r), L.ptr(pixels), L.ptr(depth),
                                         C.byref(aux_s) if aux_s is not None else None, L.ptr(ws), _stream()),
            "cnerf_render_forward")
    return pixels, depth, aux_t
