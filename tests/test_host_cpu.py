"""CPU-side checks of the host layer: C-ABI exports, struct layout, module/state-dict compatibility, init parity
with the reference (same seed -> same parameters), camera helpers.  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, GOLDEN_NAMES


def test_build_and_exports():
    import __graft_entry__ as ge
    ge.build()
    import cnerf_amd
    lib = cnerf_amd._lib.lib()
    hdr = open(os.path.join(ROOT, "include", "cnerf.h")).read()
    declared = set(re.findall(r"\b(cnerf_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cnerf.h but not exported"
    assert declared == set(cnerf_amd._lib.PROTOTYPES), "ctypes prototypes out of sync with the header"
    assert lib.cnerf_abi_version() == cnerf_amd._lib.ABI_VERSION


def test_cfg_validation_without_gpu():
    """Argument validation is host code: bad shapes are refused with a message, no launch is attempted."""
    import cnerf_amd
    L = cnerf_amd._lib
    cfg = L.Cfg()
    cfg.B, cfg.R, cfg.S, cfg.V, cfg.C, cfg.H, cfg.L = 1, 8, 8, 8, 32, 256, 4
    cfg.voxel_length, cfg.fov_deg = 1.2, 30.0
    a, b, c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    assert L.lib().cnerf_workspace_bytes(ctypes.byref(cfg), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    n = 8 * 8 * 8
    assert c.value >= n * 10 * 4 and b.value >= 8 ** 3 * 32 * 4
    # packed = 32->256, 3x 256->256, head (one 32-row tile), biases
    assert a.value >= (32 * 256 + 3 * 256 * 256 + 32 * 256 + 4 * 256 + 4) * 4
    cfg.H = 100
    assert L.lib().cnerf_workspace_bytes(ctypes.byref(cfg), ctypes.byref(a), None, None) == -22
    assert b"H=100" in L.lib().cnerf_last_error()
    cfg.H, cfg.S = 256, 1
    assert L.lib().cnerf_workspace_bytes(ctypes.byref(cfg), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == -22
    with pytest.raises(L.CnerfError):
        L.ptr(torch.zeros(3))          # CPU tensors never reach the kernels


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_init_and_state_dict_match_reference(golden, name):
    """Same torch seed -> same parameters as the reference's module (network / mapping untouched by the fixture's head
    scaling), and the reference's state-dict loads with strict=True."""
    from cnerf_amd.generators import ImplicitGenerator3d
    g = golden(name)
    m = g.meta
    torch.manual_seed(m["seed"])
    if m["variant"] == "TALLSIREN":
        gen = ImplicitGenerator3d(m["variant"], z_dim=m["C"], input_dim=3, output_dim=4, hidden_dim=m["H"])
    elif m["has_global"]:
        gen = ImplicitGenerator3d(m["variant"], z_dim=m["Z"], input_dim=m.get("input_dim", m["C"]), output_dim=4, hidden_dim=m["H"])
    else:
        gen = ImplicitGenerator3d(m["variant"], z_dim=m["C"], input_dim=m["C"], output_dim=4, hidden_dim=m["H"])
    sd = gen.state_dict()
    ref = {k[len("param/"):]: g[k] for k in g.d.files if k.startswith("param/")}
    assert set(sd) == set(ref)
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k].shape
        if "final_layer" not in k:
            assert np.array_equal(v.numpy(), ref[k]), k
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)
    assert gen.step == 0 and gen.epoch == 0 and hasattr(gen, "siren")


def test_camera_helpers():
    from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
    np.random.seed(0)
    o = sample_camera_positions("cpu", "y", 0.7, 1.5, 16)
    r = o.norm(dim=-1)
    assert ((r >= 0.7 - 1e-6) & (r <= 1.5 + 1e-6)).all()
    m = create_cam2world_matrix(o, "y")
    rot = m[:, :3, :3]
    assert torch.allclose(rot @ rot.transpose(1, 2), torch.eye(3).expand(16, 3, 3), atol=1e-5)
    assert torch.allclose(m[:, :3, 3], o)
    fwd = rot[:, :, 2]                       # camera looks at the world origin
    assert torch.allclose(fwd, -o / r.unsqueeze(-1), atol=1e-5)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_camera_helpers_match_reference(golden, name):
    """sample_camera_positions + create_cam2world_matrix against the reference's own output: every fixture stores the
    `cam2worlds` the reference produced right after np.random.seed(seed) (tests/golden/make_golden.py), so the same seed
    must give the same matrices bit for bit (NumPy draw order theta, phi, r; float64 -> float32; columns -left, -up,
    forward: volumetric_rendering.py:212-287)."""
    from cnerf_amd.generators.volumetric_rendering import sample_camera_positions, create_cam2world_matrix
    g = golden(name)
    m = g.meta
    np.random.seed(m["seed"])
    o = sample_camera_positions("cpu", "y", cam_r_start=0.7, cam_r_end=1.5, n=m["B"])
    cam = create_cam2world_matrix(o, "y", device="cpu").float()
    assert np.array_equal(cam.numpy(), g["cam2worlds"])


def test_unknown_variant_and_dropout():
    from cnerf_amd.generators import ImplicitGenerator3d
    with pytest.raises(AttributeError):
        ImplicitGenerator3d("TALLSIREN_dg", 8, 32, 4, 64)      # a name the reference's configs still mention
    gen = ImplicitGenerator3d("SHORTSIREN_FG", 8, 32, 4, 64, drop_out=0.1)
    gen.train()
    gen.siren.check_supported()                    # dropout runs in the fp32 kernels ...
    gen.siren.precision = "fp16x3"
    with pytest.raises(NotImplementedError):       # ... and only there
        gen.siren.check_supported()
    gen.eval()
    gen.siren.check_supported()                    # eval mode: nothing is dropped, any precision


def test_philox_known_answers():
    """oracle/philox.py against the known-answer vectors of Philox4x32-10 (Random123 kat_vectors): zeros, all ones, and the
    digits-of-pi counter / key."""
    from oracle import philox as P
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(x) for x in P.philox4x32_10(*ctr, *key))
        assert got == want, (ctr, key, [hex(g) for g in got])
    u = P.uniform(1234, 7, 0, 100000)
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 5e-3 and np.all(u * 2 ** 24 == np.round(u * 2 ** 24))
    n = P.normal(1234, 7, 1, 200000)
    assert abs(n.mean()) < 1e-2 and abs(n.std() - 1) < 1e-2 and np.isfinite(n).all()


def test_philox_dropout_keep_layout():
    """oracle/philox.py::dropout_keep (the NumPy twin of the kernels' dropout decisions): one Philox block per 4 consecutive
    channels, counter index ((point * n_drop + d) * H + c) / 4, word c % 4 keeps iff >= round(p * 2^32)."""
    from oracle import philox as P
    seed, off, n_pts, n_drop, H, p = 77, 3, 50, 3, 64, 0.25
    keep = P.dropout_keep(seed, off, 4, n_pts, n_drop, H, p)
    assert keep.shape == (n_drop, n_pts, H) and keep.dtype == np.uint8 and set(np.unique(keep)) <= {0, 1}
    assert abs(keep.mean() - (1 - p)) < 0.02
    pt, d, c = 17, 2, 41                                  # one element by hand
    idx = ((pt * n_drop + d) * H + c) // 4
    words = P.philox4x32_10(idx & 0xFFFFFFFF, idx >> 32, 4, off, seed & 0xFFFFFFFF, seed >> 32)
    assert keep[d, pt, c] == int(int(words[c % 4]) >= int(p * 2 ** 32 + 0.5))
    assert not np.array_equal(keep, P.dropout_keep(seed, off, 5, n_pts, n_drop, H, p))      # the fine pass draws from its own stream
    assert P.dropout_keep(seed, off, 4, n_pts, n_drop, H, 0.0).all()


def test_backward_workspace_validation_without_gpu():
    """cnerf_backward_workspace_bytes (host code of the one-call backward): sizes grow with the chunk, kept activations shrink them,
    and what the call cannot do is refused with a message -- per-point FiLM (CNERF_ENOSYS), an fp16 backward behind an fp32 forward cfg."""
    import cnerf_amd
    L = cnerf_amd._lib
    cfg = L.Cfg()
    cfg.B, cfg.R, cfg.S, cfg.V, cfg.C, cfg.H, cfg.L = 4, 16, 8, 8, 32, 64, 4
    cfg.voxel_length, cfg.fov_deg, cfg.flags = 1.2, 30.0, L.F_HIERARCHICAL
    n = ctypes.c_size_t()
    sizes = {}
    for prec in (L.PREC_FP32, L.PREC_FP16):
        cfg.precision = L.PREC_FP32 if prec == L.PREC_FP32 else L.PREC_FP16X3
        for cnt in (1, 4):
            assert L.lib().cnerf_backward_workspace_bytes(ctypes.byref(cfg), prec, cnt, 0, ctypes.byref(n)) == 0, L.lib().cnerf_last_error()
            sizes[(prec, cnt)] = n.value
    npi = 16 * 16 * 8
    assert sizes[(L.PREC_FP32, 4)] > sizes[(L.PREC_FP32, 1)] >= 2 * 4 * npi * 16 + npi * (32 + 3 * 4 * 64 + 4) * 4
    assert sizes[(L.PREC_FP16, 4)] < sizes[(L.PREC_FP32, 4)]                      # fp16 tile blocks are half the bytes
    assert L.lib().cnerf_backward_workspace_bytes(ctypes.byref(cfg), L.PREC_FP16, 4, 1, ctypes.byref(n)) == 0
    assert n.value < sizes[(L.PREC_FP16, 4)]                                       # kept activations are not part of the workspace
    assert L.lib().cnerf_backward_workspace_bytes(ctypes.byref(cfg), L.PREC_FP16, 2, 1, ctypes.byref(n)) == -22      # kept: all images at once
    assert L.lib().cnerf_backward_workspace_bytes(ctypes.byref(cfg), L.PREC_FP16, 5, 0, ctypes.byref(n)) == -22
    cfg.precision = L.PREC_FP32
    assert L.lib().cnerf_backward_workspace_bytes(ctypes.byref(cfg), L.PREC_FP16, 1, 0, ctypes.byref(n)) == -22
    assert b"fp16x3" in L.lib().cnerf_last_error()
    for l in range(4):
        cfg.layer_kind[l] = L.LAYER_PFILM
    assert L.lib().cnerf_backward_workspace_bytes(ctypes.byref(cfg), L.PREC_FP32, 1, 0, ctypes.byref(n)) == -38
