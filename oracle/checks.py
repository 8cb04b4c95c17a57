"""Shared comparison helpers of the parity checks.  TEST INFRASTRUCTURE ONLY (same rule as render_oracle.py: only tests/,
__graft_entry__.smoke() and bench.py's check / cpu_baseline leg import this; the product package never does).

Everything here compares a HIP result with the CPU oracle's; nothing here is on a measured or shipped path.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import render_oracle as O


def scaled_err(a, b) -> float:
    """max |a-b| / max(|b|, rms(b)) -- the metric of tests/conftest.py."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    floor = max(float(np.sqrt(np.mean(b * b))), 1e-30)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def rgb_sigma_err(a, b) -> float:
    """scaled_err of colour and of density, each against its own scale: the larger of the two."""
    a, b = _np(a), _np(b)
    return max(scaled_err(a[..., :3], b[..., :3]), scaled_err(a[..., 3], b[..., 3]))


# ---------------------------------------------------------------------------------------------------------------------
# rays on which the reference's image is discontinuous in its own densities (volumetric_rendering.py:30-33)
# ---------------------------------------------------------------------------------------------------------------------
def knife_edge_rays(ref_aux: Dict[str, torch.Tensor], clamp: str, tol: float = 1e-4) -> Optional[torch.Tensor]:
    """The last merged sample of a ray is composited with delta = 1e10, so under the relu clamp its alpha is 0 for sigma <= 0 and
    1 for any sigma > 0: a reference density within the rgb / sigma tolerance of zero there may land on either side.  Returns the
    mask (B, P) of rays whose last sample's reference density lies inside |sigma| <= tol * rms(sigma); None for softplus
    (continuous).  These rays are NOT excluded from the comparison: see knife_edge_branches."""
    if clamp != "relu":
        return None
    f, c = ref_aux.get("fine_rgb_sigma"), ref_aux["coarse_rgb_sigma"]
    sig = torch.cat([f[..., 3], c[..., 3]], -1) if f is not None else c[..., 3]
    last = torch.gather(sig, -1, ref_aux["sort_idx"][..., -1:].long())[..., 0] if f is not None else sig[..., -1]
    return last.abs() <= tol * float(sig.square().mean().sqrt())


def knife_edge_branches(ref_aux: Dict[str, torch.Tensor], R: int, fov: float, noise_std: float, white_back: bool, last_back: bool,
                        eps_final: Optional[torch.Tensor] = None) -> Tuple[Tuple[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]:
    """The two images the reference's algorithm yields when the alpha of every ray's LAST merged sample is forced to 0 and to 1
    (relu clamp; everything else -- the oracle's own merged samples, depths and noise draws -- unchanged):
    ((pixels0, depth0), (pixels1, depth1)), shaped like the render's outputs.  On a knife-edge ray a correct implementation must
    reproduce one of the two within tolerance; on every other ray the reference's own image IS one of the two, bit for bit
    (tests/test_oracle_golden.py::test_knife_edge_branches_contain_the_reference)."""
    c = ref_aux["coarse_rgb_sigma"]
    B, P = c.shape[0], c.shape[1]
    if ref_aux.get("fine_rgb_sigma") is not None:
        all_out, all_z, _ = O.merge_by_depth(ref_aux["fine_rgb_sigma"], c, ref_aux["fine_z"], ref_aux["coarse_z"])
    else:
        all_out, all_z = c, ref_aux["coarse_z"]
    dirs_cam = O.camera_ray_dirs(R, fov)
    out = []
    for forced in (-1.0, 1.0):
        o = all_out.clone()
        # relu(sigma + eps * noise): a magnitude no draw can overturn
        big = 1e6 * (1.0 + abs(noise_std))
        o[..., -1, 3] = forced * big
        rgb, dist, _ = O.composite(o, all_z, eps_final, noise_std, "relu", white_back, last_back)
        pixels = rgb.reshape(B, R, R, 3).permute(0, 3, 1, 2).contiguous() * 2 - 1
        depth = (dirs_cam[:, 2].reshape(1, P) * dist).reshape(B, R, R)
        out.append((pixels, depth))
    return out[0], out[1]


def image_err_with_knife_edges(px, dp, ref_px, ref_dp, edge, branches) -> Tuple[float, float, int]:
    """scaled_err of pixels and depth where every knife-edge ray (mask `edge`, (B, P)) is compared with the CLOSER of the
    reference algorithm's two admissible values (knife_edge_branches) instead of the reference's own; the error there is still
    measured and enters the maximum.  Returns (pixel err, depth err, number of knife-edge rays)."""
    px, dp, ref_px, ref_dp = (torch.as_tensor(_np(t)).clone() for t in (px, dp, ref_px, ref_dp))
    n_edge = 0
    if edge is not None and bool(edge.any()):
        B, R = dp.shape[0], dp.shape[-1]
        m = edge.reshape(B, R, R)
        n_edge = int(m.sum())
        (p0, d0), (p1, d1) = branches
        # per ray: distance to either branch, pixels and depth together, each on the image's own scale
        s_p = max(float(ref_px.square().mean().sqrt()), 1e-30)
        s_d = max(float(ref_dp.square().mean().sqrt()), 1e-30)
        e0 = torch.maximum((px - p0).abs().amax(1) / s_p, (dp - d0).abs() / s_d)
        e1 = torch.maximum((px - p1).abs().amax(1) / s_p, (dp - d1).abs() / s_d)
        pick1 = e1 < e0
        tgt_p = torch.where(pick1.unsqueeze(1), p1, p0)
        tgt_d = torch.where(pick1, d1, d0)
        ref_px = torch.where(m.unsqueeze(1), tgt_p, ref_px)
        ref_dp = torch.where(m, tgt_d, ref_dp)
    return scaled_err(px.numpy(), ref_px.numpy()), scaled_err(dp.numpy(), ref_dp.numpy()), n_edge


def merge_order_matches(sort_idx, ref_sort_idx, fine_z, coarse_z) -> bool:
    """The merge permutation equals the reference's wherever the merged depths are distinct, and the sorted depths are identical
    bit for bit everywhere (torch.sort is not stable: twins of EQUAL fp32 depth may come in either order)."""
    si, sr = _np(sort_idx).astype(np.int64), _np(ref_sort_idx).astype(np.int64)
    allz = np.concatenate([_np(fine_z), _np(coarse_z)], -1)
    za, zr = np.take_along_axis(allz, si, -1), np.take_along_axis(allz, sr, -1)
    tie = np.zeros(sr.shape, bool)
    tie[..., 1:] |= zr[..., 1:] == zr[..., :-1]
    tie[..., :-1] |= zr[..., :-1] == zr[..., 1:]
    return bool(np.array_equal(za, zr) and not ((si != sr) & ~tie).any())


# ---------------------------------------------------------------------------------------------------------------------
# "as accurate as the reference": distance of both fp32 implementations from the same algorithm in float64
# ---------------------------------------------------------------------------------------------------------------------
def field_fp64(variant: str, params: Dict[str, torch.Tensor], fvol, global_feature, points: torch.Tensor) -> torch.Tensor:
    """rgb_sigma (B,N,4) of the oracle's field network evaluated in float64 at the GIVEN fp32 sample positions (the positions
    are inputs here, not results: both fp32 implementations evaluate the field at exactly these)."""
    D = lambda t: None if t is None else t.detach().double()
    spec = O.FIELD_SPECS[variant]
    vols = [D(v) for v in fvol] if isinstance(fvol, (list, tuple)) else D(fvol)
    keep = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        with torch.no_grad():
            out, _ = O.field_eval(spec, {k: D(v) for k, v in params.items()}, vols, D(global_feature), D(points))
    finally:
        torch.set_default_dtype(keep)
    return out


def accuracy_vs_fp64(hip_rgb_sigma, ref_rgb_sigma, exact_rgb_sigma) -> Dict[str, float]:
    """{hip_vs_fp64, ref_vs_fp64, ratio}: rgb_sigma_err of the HIP result and of the fp32 reference against the float64
    evaluation at the same positions.  The 1e-4 parity gate stands for "as accurate as the reference"; this is that statement
    measured: ratio <= 2 means the HIP kernel is no further from exact than twice the reference's own fp32 rounding."""
    ex = _np(exact_rgb_sigma).reshape(_np(ref_rgb_sigma).shape)
    h, r = rgb_sigma_err(hip_rgb_sigma, ex), rgb_sigma_err(ref_rgb_sigma, ex)
    return {"hip_vs_fp64": h, "ref_vs_fp64": r, "ratio": h / max(r, 1e-30)}
