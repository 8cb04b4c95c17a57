"""Forward + backward of the render path driven through the C ABI ALONE: ctypes calls into libcnerf_hip.so with raw device
pointers -- no cnerf_amd.ops, no autograd.Function, no module classes.  What a non-PyTorch host (the cgo / JNI / plain-C caller of
INTEGRATION.md) would do; torch appears only as the allocator of device memory and, for the one piece SURVEY.md 2.1 K6 leaves on the
host, the FiLM mapping Linear.  Gradients are compared with the reference's own autograd stored in the fixtures."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import scaled_err

pytestmark = pytest.mark.gpu


def dptr(t):
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("name,fwd,bwd", [("short_fg_small", "fp32", "fp32"), ("short_fg_small", "fp16x3", "fp16"), ("short_fres_small", "fp32", "fp32"),
                                          ("tall_dres_small", "fp16x3", "fp16"), ("short_fg_nohier", "fp32", "fp32"), ("tallsiren_small", "fp16x3", "fp16")])
def test_forward_and_backward_through_ctypes_only(golden, name, fwd, bwd):
    import cnerf_amd
    L = cnerf_amd._lib                      # struct definitions and prototypes of include/cnerf.h; nothing else of the package is used
    lib = L.lib()
    dev = torch.device("cuda:0")
    g = golden(name)
    m = g.meta
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ok = lambda rc, what: (_ for _ in ()).throw(AssertionError(f"{what}: rc {rc}: {lib.cnerf_last_error().decode()}")) if rc else None

    spec_layers = {"SHORTSIREN_FG": ["film"] * 4, "SHORTSIREN_FRes": ["sine", "res", "sine"], "TALLSIREN_dRes": ["sine", "res", "res", "sine"],
                   "TALLSIREN": ["pfilm"] * 8}[m["variant"]]
    sigmoid = m["variant"] not in ("TALLSIREN_dRes", "TALLSIREN")
    pfilm = spec_layers[0] == "pfilm"
    B, R, S, H, V, Cc = m["B"], m["R"], m["S"], m["H"], g["feature_volume"].shape[-1], m["C"]
    hier = bool(m["hierarchical"])
    P, npi = R * R, R * R * S
    prm = {k: T(v) for k, v in g.params().items()}

    cfg = L.Cfg()
    cfg.B, cfg.R, cfg.S, cfg.V, cfg.C, cfg.H, cfg.L = B, R, S, V, Cc, H, len(spec_layers)
    for i, k in enumerate(spec_layers):
        cfg.layer_kind[i] = L.LAYER_CODE[k]
    cfg.ray_start, cfg.ray_end, cfg.voxel_length, cfg.noise_std, cfg.fov_deg = m["ray_start"], m["ray_end"], 1.2, m["noise"], m["fov"]
    cfg.flags = ((L.F_HIERARCHICAL if hier else 0) | (L.F_WHITE_BACK if m["white_back"] else 0) | (L.F_LAST_BACK if m["last_back"] else 0) |
                 (L.F_SOFTPLUS if m["clamp"] == "softplus" else 0) | (L.F_SIGMOID_RGB if sigmoid else 0))
    cfg.n_levels, cfg.level_V[0], cfg.level_C[0] = 1, V, Cc
    cfg.precision = L.PREC_CODE[fwd]

    # raw parameters -> cnerf_field_params, gradient buffers -> cnerf_field_param_grads
    fp, gp = L.FieldParams(), L.FieldParamGrads()
    names, grads = [], {}

    def bind(slot_w, slot_b, gw, gb, i, key):
        for slot, gslot, suffix in ((slot_w, gw, "weight"), (slot_b, gb, "bias")):
            k = f"{key}.{suffix}"
            grads[k] = torch.zeros_like(prm[k])
            if i is None:
                setattr(fp, slot, prm[k].data_ptr())
                setattr(gp, gslot, grads[k].data_ptr())
            else:
                getattr(fp, slot)[i] = prm[k].data_ptr()
                getattr(gp, gslot)[i] = grads[k].data_ptr()
            names.append(k)

    for i, k in enumerate(spec_layers):
        if k == "res":
            bind("w", "b", "w", "b", i, f"network.{i}.fc1")
            bind("w2", "b2", "w2", "b2", i, f"network.{i}.fc2")
        else:
            bind("w", "b", "w", "b", i, f"network.{i}.layer")
    bind("w_final", "b_final", "w_final", "b_final", None, "final_layer")
    if pfilm:            # the mapping MLP is evaluated per point inside the kernels (ABI v7: its gradients come back through cnerf_field_param_grads)
        bind("map_w1", "map_b1", "map_w1", "map_b1", None, "mapping_network.network.0")
        bind("map_w2", "map_b2", "map_w2", "map_b2", None, "mapping_network.network.2")

    # FiLM mapping Linear on the host (K6): freq = 15 * f + 30
    n_film = spec_layers.count("film")
    glob = T(g["global_feature"]) if n_film else None
    if n_film:
        fo = torch.nn.functional.linear(glob, prm["mapping_network.weight"], prm["mapping_network.bias"])
        freq, phase = (fo[:, :n_film * H] * 15 + 30).contiguous(), fo[:, n_film * H:].contiguous()
    else:
        freq = phase = None
    nul = C.c_void_p(None)
    p_or_null = lambda t: nul if t is None else dptr(t)

    # feature volume channel-first -> channel-last
    fvol = T(g["feature_volume"])
    fcl = torch.empty((B, V, V, V, Cc), device=dev)
    ok(lib.cnerf_fvol_channel_last(B, Cc, V, dptr(fvol), dptr(fcl), stream), "channel_last")
    vols = L.Volumes()
    vols.level[0] = fcl.data_ptr()

    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    ok(lib.cnerf_workspace_bytes(C.byref(cfg), C.byref(a), C.byref(b), C.byref(c)), "workspace_bytes")
    packed = torch.empty(a.value // 4, device=dev)
    ok(lib.cnerf_pack_field(C.byref(cfg), C.byref(fp), dptr(packed), stream), "pack_field")
    ws = torch.empty(c.value, dtype=torch.uint8, device=dev)

    rng = L.Rng()
    keep = {k: T(g[k]).reshape(B, P, -1).contiguous() for k in ("u_strat", "eps_coarse", "u_fine", "eps_final") if g.get(k) is not None}
    if hier:
        keep["fine_z"] = T(g["fine_z"]).reshape(B, P, S).contiguous()          # the reference's resampled depths, forced (see test_render_teacher_forced)
    for k, t in keep.items():
        if m["noise"] == 0 and k.startswith("eps"):
            continue
        setattr(rng, k, t.data_ptr())
    aux = L.Aux()
    sv = {"coarse_rgb_sigma": torch.empty((B, P, S, 4), device=dev), "coarse_z": torch.empty((B, P, S), device=dev)}
    if hier:
        sv.update(fine_rgb_sigma=torch.empty((B, P, S, 4), device=dev), fine_z=torch.empty((B, P, S), device=dev))
    for k, t in sv.items():
        setattr(aux, k, t.data_ptr())
    cam = T(g["cam2worlds"]).reshape(B, 4, 4).contiguous()
    pixels, depth = torch.empty((B, 3, R, R), device=dev), torch.empty((B, R, R), device=dev)
    ok(lib.cnerf_render_forward(C.byref(cfg), C.byref(vols), dptr(packed), p_or_null(freq), p_or_null(phase), dptr(cam), C.byref(rng), dptr(pixels),
                                dptr(depth), C.byref(aux), dptr(ws), stream), "render_forward")
    assert scaled_err(pixels.cpu().numpy(), g["pixels"]) < 2e-4 and scaled_err(depth.cpu().numpy(), g["depth"]) < 2e-4

    # loss = pixels.square().mean() + depth.mean()  (what make_golden.py differentiated)
    loss = pixels.square().mean() + depth.mean()
    assert abs(loss.item() - float(g["loss"])) < 2e-4 * max(1.0, abs(float(g["loss"])))
    grad_pixels = (2.0 * pixels / pixels.numel()).contiguous()
    grad_depth = torch.full_like(depth, 1.0 / depth.numel())

    # ---- the backward: one call ------------------------------------------------------------------------------------------------
    bcode = L.PREC_CODE[bwd]
    nb = C.c_size_t()
    if bwd == "fp16":
        ok(lib.cnerf_backward16_bytes(C.byref(cfg), C.byref(nb)), "backward16_bytes")
        packed_bwd = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        ok(lib.cnerf_pack_field_chain16(C.byref(cfg), C.byref(fp), dptr(packed_bwd), stream), "pack_field_chain16")
    else:
        ok(lib.cnerf_backward_bytes(C.byref(cfg), C.byref(nb)), "backward_bytes")
        packed_bwd = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        ok(lib.cnerf_pack_field_transposed(C.byref(cfg), C.byref(fp), dptr(packed_bwd), stream), "pack_field_transposed")
    chunk = max(1, B - 1) if B > 1 else 1            # (two chunks where the fixture has two images: the chunk loop is exercised)
    ok(lib.cnerf_backward_workspace_bytes(C.byref(cfg), bcode, chunk, 0, C.byref(nb)), "backward_workspace_bytes")
    bws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    g_freq = torch.zeros_like(freq) if n_film else None
    g_phase = torch.zeros_like(phase) if n_film else None
    g_vol = torch.zeros_like(fcl)
    gvols = L.Volumes()
    gvols.level[0] = g_vol.data_ptr()
    saved = L.Saved()
    for k, t in sv.items():
        setattr(saved, k, t.data_ptr())
    if hier:
        saved.fine_z = keep["fine_z"].data_ptr()           # the depths the fine pass used
    sat = torch.zeros(1, dtype=torch.int32, device=dev)
    ok(lib.cnerf_render_backward(C.byref(cfg), bcode, chunk, C.byref(vols), C.byref(fp), dptr(packed), dptr(packed_bwd), p_or_null(freq),
                                 p_or_null(phase), dptr(cam), C.byref(rng), C.byref(saved), None, dptr(grad_pixels), dptr(grad_depth), C.byref(gp),
                                 p_or_null(g_freq), p_or_null(g_phase), C.byref(gvols), dptr(sat), dptr(bws), stream), "render_backward")
    torch.cuda.synchronize()
    assert int(sat.item()) == 0

    # ---- against the reference's autograd ----------------------------------------------------------------------------------------
    # fp32 backward: the tolerance of test_backward_teacher_forced (2e-3 or 2.5 x the reference's own fp32 noise, bounded here by 2e-2 for
    # the 4-layer FiLM net); fp16 backward: relative L2 <= 2e-3 like test_backward_half_precision
    ref = {k[len("grad/siren."):]: g[k] for k in g.d.files if k.startswith("grad/siren.")}

    def close(got, want, k):
        got, want = got.cpu().numpy(), np.asarray(want)
        rel_l2 = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
        assert rel_l2 < (3e-3 if bwd == "fp16" else 2e-3), (k, rel_l2)
        assert scaled_err(got, want) < (5e-2 if bwd == "fp16" else 2e-2), (k, scaled_err(got, want))

    for k in names:
        close(grads[k], ref[k], k)
    gv_cf = torch.empty_like(fvol)
    ok(lib.cnerf_fvol_channel_first(B, Cc, V, dptr(g_vol), dptr(gv_cf), stream), "channel_first")
    close(gv_cf, g["grad_feature_volume"], "feature_volume")
    if n_film:           # the host finishes the mapping Linear: d fo = [15 dfreq | dphase]
        d_fo = torch.cat([15.0 * g_freq, g_phase], -1)
        close(d_fo.t() @ glob, ref["mapping_network.weight"], "mapping_network.weight")
        close(d_fo.sum(0), ref["mapping_network.bias"], "mapping_network.bias")
        close(d_fo @ prm["mapping_network.weight"], g["grad_global_feature"], "global_feature")
