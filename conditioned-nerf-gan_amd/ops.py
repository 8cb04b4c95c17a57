"""Thin PyTorch-side wrappers over the C ABI (include/cnerf.h).  PyTorch only provides device memory, the current
stream and autograd plumbing here; every computation below happens in libcnerf_hip.so."""
import ctypes as C
import weakref
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib as L

# packed weights per field network: dropped with the module (weak keys), re-packed when a parameter's version changes
_pack_cache = weakref.WeakKeyDictionary()
_packt_cache = weakref.WeakKeyDictionary()


def clear_pack_cache():
    """Forget every packed weight buffer (a training step that changed the weights re-packs anyway: the cache keys hold the
    parameters' version counters; bench.py calls this so that every timed step pays the packing like a training step)."""
    _pack_cache.clear()
    _packt_cache.clear()
    _pack16_cache.clear()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def as_levels(vols):
    """Channel-last volume(s) -> list of (B,V,V,V,C) tensors."""
    return list(vols) if isinstance(vols, (list, tuple)) else [vols]


def volumes_struct(levels):
    v = L.Volumes()
    for i, t in enumerate(levels):
        L.ptr(t)
        v.level[i] = t.data_ptr()
    return v


def precision_of(net):
    """Forward arithmetic of `net`: its `.precision` attribute -- "fp32" (exact fp32 MFMA), "fp16x3" (fp32-accurate three-way
    fp16 split, same parity gate) or "fp16" (plain fp16 products with fp32 sums: the reference's autocast class, ~1e-3) -- the
    every layer family has the three kernels (FiLM / plain-sine / residual: field_h3.hip, per-point FiLM: field_pw16.hip)."""
    p = getattr(net, "precision", "fp32")
    if p not in L.PREC_CODE:
        raise L.CnerfError(f"unknown precision {p!r}")
    return p


def make_cfg(net, B, vols, R=1, S=2, fov=30.0, ray_start=0.0, ray_end=1.0, noise_std=0.0, hierarchical=False,
             white_back=False, last_back=False, clamp_mode="relu", precision=None, philox=None, drop=None):
    """cnerf_cfg for a field network `net` (generators.siren.FieldNetwork); vols: channel-last volume(s) or, for calls
    that touch no volume, the side V of a single 32-channel one."""
    cfg = L.Cfg()
    if isinstance(vols, int):
        shapes = [(vols, 32)]
    else:
        shapes = [(int(t.shape[1]), int(t.shape[-1])) for t in as_levels(vols)]
    if len(shapes) > L.MAX_LEVELS:
        raise L.CnerfError(f"at most {L.MAX_LEVELS} feature volumes")
    cfg.B, cfg.R, cfg.S, cfg.V = int(B), int(R), int(S), shapes[0][0]
    cfg.C, cfg.H = sum(c for _, c in shapes), int(net.hidden_dim)
    cfg.n_levels = len(shapes)
    for i, (v, c) in enumerate(shapes):
        cfg.level_V[i], cfg.level_C[i] = v, c
    k0 = 3 if net.spec.input == "xyz" else cfg.C + (3 if net.spec.input == "feat_xyz" else 0)
    if not isinstance(vols, int) and k0 != int(net.input_dim):
        raise L.CnerfError(f"{net.variant}: layer 0 expects {net.input_dim} inputs, the feature volumes provide {k0}")
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
    flags |= L.F_INPUT_XYZ if net.spec.input == "feat_xyz" else 0
    cfg.flags = flags
    cfg.precision = L.PREC_CODE[precision if precision is not None else precision_of(net)]
    if philox is not None:            # (seed, offset): draws without a tensor are generated in the kernels
        cfg.philox, cfg.philox_seed, cfg.philox_offset = 1, int(philox[0]) & 0xFFFFFFFFFFFFFFFF, int(philox[1]) & 0xFFFFFFFF
    if drop is not None:              # (p, (seed, offset)): dropout of a network in training mode, see drop_of()
        cfg.drop_p = float(drop[0])
        if philox is None:            # the key of the keep decisions (the four draws stay tensors: cfg.philox = 0)
            cfg.philox_seed, cfg.philox_offset = int(drop[1][0]) & 0xFFFFFFFFFFFFFFFF, int(drop[1][1]) & 0xFFFFFFFF
    return cfg


def drop_of(rng):
    """rng["drop"] = (p, (seed, offset)) when the network drops out in this call (ImplicitGenerator3d.forward sets it in training mode)."""
    return (rng or {}).get("drop")


def _u8(t):
    return None if t is None else t.to(torch.uint8).contiguous()


def sizes(cfg, render=True):
    a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    L.check(L.lib().cnerf_workspace_bytes(C.byref(cfg), C.byref(a), C.byref(b), C.byref(c) if render else None),
            "cnerf_workspace_bytes")
    return a.value, b.value, c.value


def _field_params_struct(net, params):
    """cnerf_field_params from the flat tensor list of net.field_params() (mapping MLP first for the per-point family)."""
    fp = L.FieldParams()
    it = iter(params)
    if net.spec.input == "xyz":
        fp.map_w1, fp.map_b1, fp.map_w2, fp.map_b2 = (next(it).data_ptr() for _ in range(4))
    for i, kind in enumerate(net.spec.layers):
        fp.w[i], fp.b[i] = next(it).data_ptr(), next(it).data_ptr()
        if kind == "res":
            fp.w2[i], fp.b2[i] = next(it).data_ptr(), next(it).data_ptr()
    fp.w_final, fp.b_final = next(it).data_ptr(), next(it).data_ptr()
    return fp


def pack_field(net, cfg):
    """Packed MFMA-order weights of `net` (device tensor), re-packed only when a parameter changed."""
    params = [_f32(p.detach()) for p in net.field_params()]
    key = tuple((p.data_ptr(), p._version) for p in net.field_params()) + (cfg.precision,)
    hit = _pack_cache.setdefault(net, {}).get(cfg.precision)
    if hit is not None and hit[0] == key:
        return hit[1]
    nbytes, _, _ = sizes(cfg, render=False)
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=params[0].device)
    fp = _field_params_struct(net, params)
    L.check(L.lib().cnerf_pack_field(C.byref(cfg), C.byref(fp), L.ptr(packed), _stream()), "cnerf_pack_field")
    packed._keepalive = params
    _pack_cache[net][cfg.precision] = (key, packed)
    return packed


def channel_last_levels(vols):
    """Channel-first volume or list of volumes -> list of channel-last volumes."""
    return [channel_last(v) for v in as_levels(vols)]


def channel_last(fvol):
    """(B,C,V,V,V) -> (B,V,V,V,C) on the GPU (C a multiple of 32).  A volume that already lives in memory channel-last
    (torch.channels_last_3d, e.g. an encoder whose last convolution writes that format: SURVEY.md 8f-1) is taken as it is:
    no transpose kernel, no copy."""
    if fvol.dim() == 5 and fvol.dtype == torch.float32 and fvol.is_cuda:
        view = fvol.detach().permute(0, 2, 3, 4, 1)
        if view.is_contiguous() and fvol.shape[2] == fvol.shape[3] == fvol.shape[4]:
            return view
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
    cfg = make_cfg(net, B, int(fvol_cl.shape[1]))
    out = torch.empty((B, n, fvol_cl.shape[-1]), dtype=torch.float32, device=points.device)
    L.check(L.lib().cnerf_gather_features(C.byref(cfg), L.ptr(fvol_cl), L.ptr(points), n, L.ptr(out), _stream()),
            "cnerf_gather_features")
    return out


def field_forward(net, fvol, freq, phase, points, fvol_is_channel_last=False, drop=None):
    """rgb_sigma (B,n,4) of `net` at explicit world points (B,n,3).  drop = (p, (seed, offset)): dropout of a training-mode call."""
    points = _f32(points)
    B, n = points.shape[0], points.shape[1]
    levels = [_f32(v) for v in as_levels(fvol)] if fvol_is_channel_last else channel_last_levels(fvol)
    cfg = make_cfg(net, B, levels, drop=drop)
    packed = pack_field(net, cfg)
    out = torch.empty((B, n, 4), dtype=torch.float32, device=points.device)
    vs = volumes_struct(levels)
    L.check(L.lib().cnerf_field_forward(C.byref(cfg), C.byref(vs), L.ptr(packed), L.ptr(_f32(freq)), L.ptr(_f32(phase)),
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


def philox_fill(seed, offset, stream_id, n, normal, device):
    """The draws the kernels generate under cnerf_cfg.philox, as a tensor: stream 0 u_strat, 1 eps_coarse, 2 u_fine, 3 eps_final."""
    out = torch.empty(int(n), dtype=torch.float32, device=device)
    L.check(L.lib().cnerf_philox_fill(int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFF, int(stream_id), int(n), 1 if normal else 0,
                                      L.ptr(out), _stream()), "cnerf_philox_fill")
    return out


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
                   want_aux=False, fvol_is_channel_last=False, field_events=None, aux_keys=None, act16=None):
    """ImplicitGenerator3d.forward on the GPU.  rng: dict with u_strat / eps_coarse / u_fine / eps_final tensors.
    field_events: optional 4 hipEvent_t handles (ints) recorded around the two field-kernel launches."""
    cam2world = _f32(cam2world)
    B, R, S = cam2world.shape[0], int(img_size), int(num_steps)
    P = R * R
    n = 2 * S if hierarchical else S
    dev = cam2world.device
    levels = [_f32(v) for v in as_levels(fvol)] if fvol_is_channel_last else channel_last_levels(fvol)
    cfg = make_cfg(net, B, levels, R, S, fov, ray_start, ray_end, noise_std, hierarchical, white_back,
                   last_back, clamp_mode, philox=(rng or {}).get("philox"), drop=drop_of(rng))
    vs = volumes_struct(levels)
    packed = pack_field(net, cfg)
    _, _, ws_bytes = sizes(cfg)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    pixels = torch.empty((B, 3, R, R), dtype=torch.float32, device=dev)
    depth = torch.empty((B, R, R), dtype=torch.float32, device=dev)
    rng = rng or {}
    keep = [_f32(rng.get(k)) for k in ("u_strat", "eps_coarse", "u_fine", "eps_final", "fine_z")]
    r = L.Rng()
    r.u_strat, r.eps_coarse, r.u_fine, r.eps_final, r.fine_z = [None if t is None else t.data_ptr() for t in keep]
    keep += [_u8(rng.get(k)) for k in ("drop_coarse", "drop_fine")]       # injected dropout decisions (tests)
    r.drop_coarse, r.drop_fine = [None if t is None else t.data_ptr() for t in keep[-2:]]
    for t in keep:
        L.ptr(t)   # validates device / contiguity
    aux_t, aux_s = {}, None
    if want_aux or aux_keys:
        aux_s = L.Aux()
        for k in (L.AUX_FIELDS if want_aux else aux_keys):
            if not hierarchical and k in HIER_ONLY:
                continue
            shape, dt = AUX_SHAPES[k](B, P, S, n)
            aux_t[k] = torch.empty(shape, dtype=dt, device=dev)
            setattr(aux_s, k, aux_t[k].data_ptr())
    if field_events is not None:
        aux_s = aux_s if aux_s is not None else L.Aux()
        for i, ev in enumerate(field_events):
            aux_s.field_events[i] = ev
    if act16 is not None:          # keep the field passes' activations (fp16 tile blocks) for the half-precision backward
        aux_s = aux_s if aux_s is not None else L.Aux()
        for i, bufs in enumerate(act16):
            aux_s.act16[i].feat, aux_s.act16[i].h, aux_s.act16[i].c = (t.data_ptr() for t in bufs[:3])
            if len(bufs) > 3:
                aux_s.act16[i].amax = bufs[3].data_ptr()
    L.check(L.lib().cnerf_render_forward(C.byref(cfg), C.byref(vs), L.ptr(packed), L.ptr(_f32(freq)), L.ptr(_f32(phase)),
                                         L.ptr(cam2world), C.byref(r), L.ptr(pixels), L.ptr(depth),
                                         C.byref(aux_s) if aux_s is not None else None, L.ptr(ws), _stream()),
            "cnerf_render_forward")
    return pixels, depth, aux_t


# ---------------------------------------------------------------------------------------------------------------------
# autograd
# ---------------------------------------------------------------------------------------------------------------------
SAVED_KEYS = ("coarse_rgb_sigma", "coarse_z", "fine_rgb_sigma", "fine_z")


def pack_field_transposed(net, cfg):
    params = [_f32(p.detach()) for p in net.field_params()]
    key = tuple((p.data_ptr(), p._version) for p in net.field_params())
    hit = _packt_cache.get(net)
    if hit is not None and hit[0] == key:
        return hit[1]
    nb = C.c_size_t(0)
    L.check(L.lib().cnerf_backward_bytes(C.byref(cfg), C.byref(nb)), "cnerf_backward_bytes")
    packed_t = torch.empty(nb.value // 4, dtype=torch.float32, device=params[0].device)
    fp = _field_params_struct(net, params)
    L.check(L.lib().cnerf_pack_field_transposed(C.byref(cfg), C.byref(fp), L.ptr(packed_t), _stream()),
            "cnerf_pack_field_transposed")
    packed_t._keepalive = params
    _packt_cache[net] = (key, packed_t)
    return packed_t


DEBUG_CAPTURE = None            # set to a dict to capture the chunk buffers of the per-point FiLM backward (debugging)
PHASE_TIMER = None              # an object with begin() / end(name, start) (training.gan_step.PhaseTimer): RenderFunction.backward
                                # reports its span as "render_bwd" so that a caller can split an autograd pass (bench.py gan_step)
ACT_BUDGET_BYTES = 128 << 30    # chunk buffers of the per-point FiLM backward: at most this much, and MEMORY_FRACTION of what is free


def _pfilm_backward(net, o, cfg, levels, cam2world, rng, saved, gc, gf):
    """Field gradients of the per-point FiLM family (TALLSIREN, siren.py:232-331).  Per pass and chunk of images
    cnerf_field_backward re-runs the forward storing its rows, runs the gradient chain of the eight FiLM layers on the MFMA
    units and leaves g_pre_l = d/d(W_l y_{l-1} + b_l) and G = d/d(output of the mapping network's second Linear) in the
    chunk buffers (include/cnerf.h); what remains are reductions over those matrices: cnerf_weight_grad for the (H,H) layer
    matrices, library GEMMs for the (2LH, 256) mapping Linear and the two thin ones, cnerf_scatter_features for the volume."""
    fvol = levels[0]
    B, R, S, hier = o["B"], o["R"], o["S"], o["hier"]
    dev = fvol.device
    H, nl = int(net.hidden_dim), len(net.spec.layers)
    npi = R * R * S
    vs = volumes_struct(levels)
    packed = pack_field(net, cfg)
    packed_t = pack_field_transposed(net, cfg)
    ps = [_f32(p.detach()) for p in net.field_params()]       # Wm1, bm1, Wm2, bm2, (W_l, b_l) x L, W_head, b_head
    grads = [torch.zeros_like(p) for p in ps]
    g_level = torch.zeros_like(fvol)
    gvs = volumes_struct([g_level])
    c_rs, c_z, f_rs, f_z, c_pts, f_pts = saved[:6]
    per_image = npi * (32 + 256 + 7 * nl * H + 4) * 4
    nb = max(1, min(B, int(min(ACT_BUDGET_BYTES, MEMORY_FRACTION * free_device_bytes(dev))) // per_image))
    u_strat = _f32(rng.get("u_strat"))
    act = None
    passes = [(0, gc, c_rs, c_pts)] + ([(1, gf, f_rs, f_pts)] if hier else [])
    for pss, g_out, saved_out, pts_all in passes:
        pts_all = pts_all.reshape(B, npi, 3)
        for b0 in range(0, B, nb):
            cnt = min(nb, B - b0)
            n = cnt * npi
            if act is None or act[0].shape[0] != n:
                act = None          # release the previous chunk before allocating a differently sized one
                act = (torch.empty((n, 32), dtype=torch.float32, device=dev),
                       torch.empty(nl * n * H + n * 256, dtype=torch.float32, device=dev),
                       torch.empty((3 * nl, n, H), dtype=torch.float32, device=dev),
                       torch.empty(3 * nl * n * H, dtype=torch.float32, device=dev),
                       torch.empty((n, 4), dtype=torch.float32, device=dev))
            a_feat, a_h, a_c, a_g, a_go = act
            L.check(L.lib().cnerf_field_backward(C.byref(cfg), pss, b0, cnt, C.byref(vs), L.ptr(packed), L.ptr(packed_t), None, None,
                                                 L.ptr(cam2world), L.ptr(u_strat), L.ptr(f_z) if hier else None, L.ptr(g_out),
                                                 L.ptr(saved_out), L.ptr(a_feat), L.ptr(a_h), L.ptr(a_c), L.ptr(a_g), L.ptr(a_go),
                                                 C.byref(gvs), L.ptr(_u8(rng.get("drop_fine" if pss else "drop_coarse"))), _stream()),
                    "cnerf_field_backward")
            y, m = a_h[:nl * n * H].view(nl, n, H), a_h[nl * n * H:].view(n, 256)
            gp, G = a_g[:nl * n * H].view(nl, n, H), a_g[nl * n * H:].view(n, 2 * nl * H)
            pts = pts_all[b0:b0 + cnt].reshape(n, 3)
            if DEBUG_CAPTURE is not None:
                DEBUG_CAPTURE.setdefault(("pfilm", pss), dict(feat=a_feat.clone(), y=y.clone(), m=m.clone(), c=a_c.clone(), gp=gp.clone(),
                                                              G=G.clone(), go=a_go.clone(), pts=pts.clone()))
            grads[4] += gp[0].t() @ pts                                   # layer 0 reads the sample position: (H, 3)
            grads[5] += gp[0].sum(0)
            for l in range(1, nl):
                dWl = torch.zeros((cnt, H, H), dtype=torch.float32, device=dev)
                cs = torch.zeros((cnt, H), dtype=torch.float32, device=dev)
                L.check(L.lib().cnerf_weight_grad(cnt, npi, H, H, L.ptr(gp[l]), L.ptr(y[l - 1]), L.ptr(dWl), L.ptr(cs), _stream()),
                        "cnerf_weight_grad")
                grads[4 + 2 * l] += dWl.sum(0)
                grads[5 + 2 * l] += cs.sum(0)
            grads[4 + 2 * nl] += a_go.t() @ y[nl - 1]
            grads[5 + 2 * nl] += a_go.sum(0)
            # mapping network: Linear(C, 256) -> LeakyReLU(0.2) -> Linear(256, 2 L H)
            grads[2] += G.t() @ m
            grads[3] += G.sum(0)
            g_m = G @ ps[2]
            g_m *= torch.where(m > 0, 1.0, 0.2)
            grads[0] += g_m.t() @ a_feat
            grads[1] += g_m.sum(0)
            d_feat = (g_m @ ps[0]).contiguous()
            cfgc = make_cfg(net, cnt, int(fvol.shape[1]))
            L.check(L.lib().cnerf_scatter_features(C.byref(cfgc), L.ptr(pts), npi, L.ptr(d_feat), L.ptr(g_level[b0:b0 + cnt]), _stream()),
                    "cnerf_scatter_features")
    return [g_level], None, None, grads


def backward_precision_of(net):
    """Arithmetic of the backward's gradient GEMMs: `net.backward_precision` = "fp32" (exact fp32 MFMA chain and weight
    gradients, the default) or "fp16" (fp16 operands, fp32 sums: the reference's own training numerics, utils.py:643 autocast):
    bwd16.hip for FiLM / plain-sine / residual-block networks, chain_pw16.hip for the per-point FiLM family."""
    p = getattr(net, "backward_precision", "fp32")
    if p not in ("fp32", "fp16"):
        raise L.CnerfError(f"unknown backward precision {p!r}")
    return p


def n_matrices(net):
    """Weight matrices before the head ("slabs" of the backward's buffers): a residual block has two."""
    return sum(2 if k == "res" else 1 for k in net.spec.layers)


_pack16_cache = weakref.WeakKeyDictionary()


def pack_field_chain16(net, cfg):
    """Transposed fp16 weight units of the half-precision gradient chain (cnerf_pack_field_chain16), cached per parameter version."""
    params = [_f32(p.detach()) for p in net.field_params()]
    key = tuple((p.data_ptr(), p._version) for p in net.field_params())
    hit = _pack16_cache.get(net)
    if hit is not None and hit[0] == key:
        return hit[1]
    nb = C.c_size_t(0)
    L.check(L.lib().cnerf_backward16_bytes(C.byref(cfg), C.byref(nb)), "cnerf_backward16_bytes")
    packed16 = torch.empty(nb.value, dtype=torch.uint8, device=params[0].device)
    fp = _field_params_struct(net, params)
    L.check(L.lib().cnerf_pack_field_chain16(C.byref(cfg), C.byref(fp), L.ptr(packed16), _stream()), "cnerf_pack_field_chain16")
    packed16._keepalive = params
    _pack16_cache[net] = (key, packed16)
    return packed16


RESIDENT_BUDGET_BYTES = 160 << 30    # fp16 activations kept from the forward for the backward: at most this much (288 GB of HBM per GPU) ...
RESIDENT_FRACTION = 0.5              # ... and at most this share of the memory that is free when the forward starts


def resident_act16(net, levels, B, R, S, hier, dev):
    """fp16 tile-block buffers that keep the activations of the forward's field passes for the half-precision backward (one
    set per pass: x0, sin, cos), or None when they would not fit the budget -- the backward then re-computes them chunk-wise."""
    H, NT = int(net.hidden_dim), int(net.hidden_dim) // 32
    n_in = sum(int(t.shape[-1]) for t in levels) // 32 + (1 if net.spec.input == "feat_xyz" else 0)
    nslab = n_matrices(net)
    T = B * ((R * R * S + 31) // 32)
    n_pass = 2 if hier else 1
    f16 = dict(dtype=torch.float16, device=dev)
    if net.spec.layers[0] == "pfilm":      # field_pw16.hip: feature + position | y_l slabs then m | cos, cos f, cos 15 pre per layer | per-point maxima
        blocks = 2 + nslab * NT + 8 + 3 * nslab * NT
        if n_pass * T * 2048 * blocks + T * 2048 * (3 * nslab * NT + 9) > min(RESIDENT_BUDGET_BYTES, RESIDENT_FRACTION * free_device_bytes(dev)):
            return None
        return [(torch.empty((T, 2, 32, 32), **f16), torch.empty(((nslab * NT + 8) * T, 32, 32), **f16), torch.empty((3 * nslab, T, NT, 32, 32), **f16),
                 torch.empty((nslab, T * 32), dtype=torch.float32, device=dev)) for _ in range(n_pass)]
    per_pass = T * 2048 * (n_in + 2 * nslab * NT)
    if n_pass * per_pass + T * 2048 * (nslab * NT + 1) > min(RESIDENT_BUDGET_BYTES, RESIDENT_FRACTION * free_device_bytes(dev)):
        return None
    return [(torch.empty((T, n_in, 32, 32), **f16), torch.empty((nslab, T, NT, 32, 32), **f16), torch.empty((nslab, T, NT, 32, 32), **f16))
            for _ in range(n_pass)]


def _field_param_grads_struct(net, grads):
    """cnerf_field_param_grads over the flat tensor list `grads` (same order as net.field_params(): mapping MLP first for the per-point family)."""
    gp = L.FieldParamGrads()
    it = iter(grads)
    if net.spec.input == "xyz":
        gp.map_w1, gp.map_b1, gp.map_w2, gp.map_b2 = (next(it).data_ptr() for _ in range(4))
    for i, kind in enumerate(net.spec.layers):
        gp.w[i], gp.b[i] = next(it).data_ptr(), next(it).data_ptr()
        if kind == "res":
            gp.w2[i], gp.b2[i] = next(it).data_ptr(), next(it).data_ptr()
    gp.w_final, gp.b_final = next(it).data_ptr(), next(it).data_ptr()
    return gp


def free_device_bytes(dev):
    """What an allocation could get right now: the driver's free memory plus what torch's caching allocator holds unused."""
    free, _ = torch.cuda.mem_get_info(dev)
    return free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)


def backward_chunk(cfg, bprec_code, B, have_act16, dev):
    """(images per chunk, workspace bytes) of cnerf_render_backward: the largest chunk whose workspace fits MEMORY_FRACTION of the
    memory that is free NOW -- not a constant sized for an empty 288 GB part: whatever the encoder, the discriminator
    and DDP already hold is accounted for, and a smaller or shared device gets smaller chunks instead of an out-of-memory error."""
    budget = int(MEMORY_FRACTION * free_device_bytes(dev))
    need = C.c_size_t(0)
    for nb in ([B] if have_act16 else range(B, 0, -1)):
        L.check(L.lib().cnerf_backward_workspace_bytes(C.byref(cfg), bprec_code, nb, 1 if have_act16 else 0, C.byref(need)), "cnerf_backward_workspace_bytes")
        if need.value <= budget or nb == 1:
            break
    if need.value > budget:
        raise L.CnerfError(f"render backward: one image needs {need.value / 2**30:.1f} GiB of chunk buffers, {budget / 2**30:.1f} GiB are free")
    return nb, need.value


MEMORY_FRACTION = 0.8
LAST_SATURATED = None          # device int32 tensor of the most recent fp16 backward: clamped (tile, matrix) blocks, see include/cnerf.h


def render_backward(net, o, levels, freq, phase, cam2world, rng, saved, grad_pixels, grad_depth, act16=None):
    """Gradients of one render w.r.t. (channel-last feature volumes, freq, phase, [field parameters]): ONE call into the library
    (cnerf_render_backward); the exact fp32 backward of the per-point FiLM family finishes its mapping-MLP gradients with library
    GEMMs (_pfilm_backward)."""
    global LAST_SATURATED
    B, R, S, hier = o["B"], o["R"], o["S"], o["hier"]
    dev = cam2world.device
    bprec = backward_precision_of(net)
    # cfg: precision of the forward (the activation-storing re-run follows it; the fp16 backward re-runs the fp16x3 kernel)
    cfg = make_cfg(net, B, levels, R, S, o["fov"], o["ray_start"], o["ray_end"], o["noise_std"], hier,
                   o["white_back"], o["last_back"], o["clamp_mode"], precision="fp16x3" if bprec == "fp16" else None,
                   philox=rng.get("philox"), drop=drop_of(rng))
    c_rs, c_z, f_rs, f_z = saved[:4]
    grad_pixels = _f32(grad_pixels)
    grad_depth = _f32(grad_depth) if grad_depth is not None else None
    if net.spec.layers[0] == "pfilm" and bprec == "fp32":      # (fp32 chain: the activation-storing re-run and its packed weights are the fp32 kernel's)
        cfg = make_cfg(net, B, levels, R, S, o["fov"], o["ray_start"], o["ray_end"], o["noise_std"], hier, o["white_back"], o["last_back"],
                       o["clamp_mode"], precision="fp32", philox=rng.get("philox"), drop=drop_of(rng))
        gc = torch.empty_like(c_rs)
        gf = torch.empty_like(f_rs) if hier else None
        eps_final = _f32(rng.get("eps_final")) if o["noise_std"] != 0 else None
        L.check(L.lib().cnerf_merge_composite_backward(C.byref(cfg), L.ptr(c_rs), L.ptr(c_z), L.ptr(f_rs) if hier else None,
                                                       L.ptr(f_z) if hier else None, L.ptr(eps_final), L.ptr(grad_pixels),
                                                       L.ptr(grad_depth), L.ptr(gc), L.ptr(gf) if hier else None, _stream()),
                "cnerf_merge_composite_backward")
        return _pfilm_backward(net, o, cfg, levels, cam2world, rng, saved, gc, gf)
    packed = pack_field(net, cfg)
    if bprec == "fp16":
        packed_bwd = pack_field_chain16(net, cfg)
    else:
        cfg32 = make_cfg(net, B, levels, R, S, precision="fp32")
        packed_bwd = pack_field_transposed(net, cfg32)
    params = [_f32(p.detach()) for p in net.field_params()]
    grads = [torch.zeros_like(p) for p in params]
    fp, gp = _field_params_struct(net, params), _field_param_grads_struct(net, grads)
    n_film = sum(1 for k in net.spec.layers if k == "film")
    H = int(net.hidden_dim)
    g_freq = torch.zeros((B, n_film * H), dtype=torch.float32, device=dev) if n_film else None
    g_phase = torch.zeros_like(g_freq) if n_film else None
    grad_levels = [torch.zeros_like(v) for v in levels]
    vs, gvs = volumes_struct(levels), volumes_struct(grad_levels)
    code = L.PREC_CODE[bprec]
    nb, ws_bytes = backward_chunk(cfg, code, B, act16 is not None, dev)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    sat = torch.zeros(1, dtype=torch.int32, device=dev)
    keep = [_f32(rng.get(k)) for k in ("u_strat", "eps_final")] + [_u8(rng.get(k)) for k in ("drop_coarse", "drop_fine")]
    r = L.Rng()
    r.u_strat, r.eps_final, r.drop_coarse, r.drop_fine = [None if t is None else t.data_ptr() for t in keep]
    for t in keep:
        L.ptr(t)
    sv = L.Saved()
    sv.coarse_rgb_sigma, sv.coarse_z = c_rs.data_ptr(), c_z.data_ptr()
    if hier:
        sv.fine_rgb_sigma, sv.fine_z = f_rs.data_ptr(), f_z.data_ptr()
    for t in (c_rs, c_z) + ((f_rs, f_z) if hier else ()):
        L.ptr(t)
    aux = None
    if act16 is not None:
        aux = L.Aux()
        for i, bufs in enumerate(act16):
            aux.act16[i].feat, aux.act16[i].h, aux.act16[i].c = (t.data_ptr() for t in bufs[:3])
            if len(bufs) > 3:
                aux.act16[i].amax = bufs[3].data_ptr()
    L.check(L.lib().cnerf_render_backward(C.byref(cfg), code, nb, C.byref(vs), C.byref(fp), L.ptr(packed), L.ptr(packed_bwd), L.ptr(freq), L.ptr(phase),
                                          L.ptr(cam2world), C.byref(r), C.byref(sv), C.byref(aux) if aux is not None else None,
                                          L.ptr(grad_pixels), L.ptr(grad_depth), C.byref(gp), L.ptr(g_freq), L.ptr(g_phase), C.byref(gvs),
                                          L.ptr(sat) if bprec == "fp16" else None, L.ptr(ws), _stream()), "cnerf_render_backward")
    if bprec == "fp16":
        LAST_SATURATED = sat
    return grad_levels, g_freq, g_phase, grads


class RenderFunction(torch.autograd.Function):
    """ImplicitGenerator3d.forward as one autograd node: differentiable w.r.t. the feature volume(s), freq/phase (and
    through them the mapping network and the global feature) and the field parameters; not w.r.t. cameras or sample
    positions (the reference runs those under no_grad: generators.py:57,111)."""

    @staticmethod
    def forward(ctx, net, o, rng, cam2world, freq, phase, n_vols, *rest):
        vols, params = rest[:n_vols], rest[n_vols:]
        levels = [channel_last(v.detach()) for v in vols]
        # a volume that arrived channel-last (zero-copy above) gets its gradient back in the same memory format, again a view
        ctx.vol_is_cl = [lv.data_ptr() == v.data_ptr() and not v.is_contiguous() for lv, v in zip(levels, vols)]
        fr = freq.detach() if freq is not None else None
        ph = phase.detach() if phase is not None else None
        need_grad = o.get("need_grad", any(ctx.needs_input_grad))   # (decided by ops.render: grad mode is always off in here)
        keys = SAVED_KEYS + (("coarse_points", "fine_points") if net.spec.layers[0] == "pfilm" else ())   # see _pfilm_backward
        act16 = None
        if need_grad and backward_precision_of(net) == "fp16" and precision_of(net) in (("fp16x3",) if net.spec.layers[0] == "pfilm" else ("fp16x3", "fp16")):
            act16 = resident_act16(net, levels, cam2world.shape[0], o["R"], o["S"], o["hier"], cam2world.device)
        pixels, depth, aux = render_forward(net, levels, fr, ph, cam2world, o["R"], o["fov"], o["ray_start"], o["ray_end"],
                                            o["S"], o["hier"], o["clamp_mode"], o["noise_std"], o["white_back"],
                                            o["last_back"], rng, want_aux=o["want_aux"], fvol_is_channel_last=True,
                                            field_events=o.get("field_events"),
                                            aux_keys=keys if need_grad and not o["want_aux"] else None, act16=act16)
        ctx.net, ctx.o, ctx.rng, ctx.n_vols = net, o, rng, n_vols
        ctx.act16 = act16
        if need_grad:
            saved = [aux.get(k) for k in keys]
            if o["hier"] and rng.get("fine_z") is not None:
                saved[3] = _f32(rng["fine_z"]).reshape(saved[1].shape)   # teacher-forced depths are what the fine pass used
            ctx.saved = (levels, fr, ph, _f32(cam2world), tuple(saved))
        RenderFunction.last_aux = aux
        return pixels, depth

    @staticmethod
    def backward(ctx, grad_pixels, grad_depth):
        levels, fr, ph, cam2world, saved = ctx.saved
        t0 = PHASE_TIMER.begin() if PHASE_TIMER is not None else None
        g_levels, g_freq, g_phase, g_params = render_backward(ctx.net, ctx.o, levels, fr, ph, cam2world, ctx.rng, saved,
                                                              grad_pixels.contiguous(),
                                                              grad_depth.contiguous() if grad_depth is not None else None,
                                                              act16=ctx.act16)
        ctx.act16 = None
        g_vols = [g.permute(0, 4, 1, 2, 3) if cl else channel_first(g) for g, cl in zip(g_levels, ctx.vol_is_cl)]
        if t0 is not None:
            PHASE_TIMER.end("render_bwd", t0)
        return (None, None, None, None, g_freq, g_phase, None, *g_vols, *g_params)


def render(net, fvol, freq, phase, cam2world, img_size, fov, ray_start, ray_end, num_steps, hierarchical, clamp_mode,
           noise_std, white_back=False, last_back=False, rng=None, want_aux=False, field_events=None):
    """Differentiable entry used by ImplicitGenerator3d.forward.  fvol: (B,C,V,V,V) or a list of pyramid levels.
    Returns (pixels, depth, aux dict)."""
    o = dict(B=cam2world.shape[0], R=int(img_size), S=int(num_steps), fov=float(fov), ray_start=float(ray_start),
             ray_end=float(ray_end), hier=bool(hierarchical), clamp_mode=clamp_mode, noise_std=float(noise_std),
             white_back=bool(white_back), last_back=bool(last_back), want_aux=bool(want_aux), field_events=field_events)
    if clamp_mode not in ("relu", "softplus"):
        raise TypeError("Need to choose clamp mode")
    vols = as_levels(fvol)
    # Will anything back-propagate through this render?  Decided HERE: inside Function.forward grad mode is always off, and
    # ctx.needs_input_grad is True for every parameter even under torch.no_grad() -- the D step's no-grad renders then ran the
    # activation-storing forward (14.7 instead of 9.9 ms per launch at batch 8) and allocated the kept activations for nothing.
    params = net.field_params()
    o["need_grad"] = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (freq, phase, *vols, *params))
    pixels, depth = RenderFunction.apply(net, o, rng or {}, cam2world, freq, phase, len(vols), *vols, *params)
    return pixels, depth, (RenderFunction.last_aux if want_aux else {})
