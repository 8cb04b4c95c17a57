"""CPU oracle for the conditioned-NeRF render path.  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 restatement (torch CPU ops, no autograd tricks, no GPU) of
the algorithm the reference implements in
    generators/volumetric_rendering.py, generators/siren.py, generators/generators.py
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the
product package never does.  It never reads /root/reference.

Parity pin: tests/test_oracle_golden.py checks every function below against the golden
vectors in tests/golden/*.npz, which tests/golden/make_golden.py produced by importing the
reference itself in the build container (the reference ships no tests or fixtures of its
own: SURVEY.md section 4).

All tensors are float32 on the CPU.  Rays: P = R*R per image, pixel p = row*R + col.
RNG is injected (u_strat, eps_coarse, u_fine, eps_final) in the reference's draw order.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

VOXEL_LENGTH = 1.2  # generators/siren.py:555 (hard-coded in every feature-volume variant)

# ---------------------------------------------------------------------------------------
# field-network table: what each SIREN variant is made of
# ---------------------------------------------------------------------------------------
# layer kinds:  "film"  y = sin(freq * (x W^T + b) + phase), freq/phase per image   (siren.py:146-160)
#               "sine"  y = sin(x W^T + b)                                           (siren.py:180-199)
#               "res"   y = sin(x + fc2(sin(fc1 x)))                                 (siren.py:218-230)
#               "pfilm" y = sin(freq(p) * (x W^T + b) + phase(p)), freq/phase per POINT from a mapping MLP of the
#                       looked-up feature                                             (siren.py:163-177, 81-101, 232-331)
# input kinds:  "feat" looked-up feature; "feat_xyz" feature || world xyz (siren.py:1158); "pyramid" features of several
#               volumes concatenated (siren.py:1444-1473); "xyz" world position only (TALLSIREN, siren.py:312)
@dataclass(frozen=True)
class FieldSpec:
    layers: Tuple[str, ...]
    sigmoid_rgb: bool
    has_global: bool
    input: str = "feat"

    @property
    def n_film(self) -> int:
        return sum(1 for k in self.layers if k == "film")


FIELD_SPECS: Dict[str, FieldSpec] = {
    "SHORTSIREN_FG": FieldSpec(("film",) * 4, True, True),       # siren.py:583-668
    "TALLSIREN_FG": FieldSpec(("film",) * 8, True, True),        # siren.py:491-580
    "DOUBLESIREN_FG": FieldSpec(("film",) * 2, True, True),      # siren.py:744-827
    "SingleSIREN_dg": FieldSpec(("film",), False, True),         # siren.py:983-1065
    "SHORTSIREN_F": FieldSpec(("sine",) * 4, True, False),       # siren.py:830-904
    "SHORTSIREN_FRes": FieldSpec(("sine", "res", "sine"), True, False),          # siren.py:906-979
    "TALLSIREN_dRes": FieldSpec(("sine", "res", "res", "sine"), False, False),   # siren.py:333-408
    "TALLSIREN_dResLong": FieldSpec(("sine", "res", "res", "res", "res", "sine"), False, False),  # :411-488
    "TALLSIREN_dgx": FieldSpec(("film",) * 8, False, True, "feat_xyz"),                           # :1068-1169
    "SHORTSIREN_FG_Pyrmd": FieldSpec(("film",) * 4, True, True, "pyramid"),                       # :671-741
    "TALLSIREN": FieldSpec(("pfilm",) * 8, False, False, "xyz"),                                  # :232-331
}


# ---------------------------------------------------------------------------------------
# a1-a4  rays, stratified depths, camera transform     (volumetric_rendering.py:73-199)
# ---------------------------------------------------------------------------------------
def camera_ray_dirs(R: int, fov_deg: float) -> torch.Tensor:
    """Unit ray directions in camera space, (R*R, 3).  x follows the column, y the row, no flip."""
    lin = torch.linspace(-1, 1, R)
    x = lin.repeat(R)                     # column index runs fastest
    y = lin.repeat_interleave(R)
    zc = torch.ones(R * R) / math.tan((2 * math.pi * fov_deg / 360) / 2)
    d = torch.stack([x, y, zc], -1)
    return d / torch.norm(d, dim=-1, keepdim=True)


def stratified_depths(B: int, R: int, S: int, ray_start: float, ray_end: float,
                      u_strat: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns (z_lin (S,), offset (B,P,S), z (B,P,S)): offset = (u - 0.5) * (z_lin[1] - z_lin[0]), z = z_lin + offset."""
    z_lin = torch.linspace(ray_start, ray_end, S)
    offset = (u_strat.reshape(B, R * R, S) - 0.5) * (z_lin[1] - z_lin[0])
    return z_lin, offset, z_lin.reshape(1, 1, S) + offset


def coarse_world_points(cam2world: torch.Tensor, dirs_cam: torch.Tensor, z_lin: torch.Tensor,
                        offset: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Camera-space sample = dir*z_lin + offset*dir (two roundings, as the reference adds the jitter to
    already-formed points), then the 4x4 cam2world; (B,P,S,3), dirs (B,P,3), origins (B,3)."""
    B, P, S = offset.shape
    pc = dirs_cam.reshape(1, P, 1, 3) * z_lin.reshape(1, 1, S, 1) + offset.unsqueeze(-1) * dirs_cam.reshape(1, P, 1, 3)
    # The reference does these two transforms with torch.bmm (volumetric_rendering.py:168-192); its CPU BLAS evaluates
    # each output as the k-ordered fused chain fma(m2, z, fma(m1, y, m0*x)) (+ m3): verified bit-for-bit against the
    # golden vectors.  Spelled out here (fma emulated through float64) so that the oracle does not depend on which
    # GEMM micro-kernel the host's BLAS picks for a 3x3 or 4x4 product.
    m = cam2world.reshape(B, 1, 1, 4, 4)
    pw = _fma_chain3(m[..., :3, 0], pc[..., 0:1], m[..., :3, 1], pc[..., 1:2], m[..., :3, 2], pc[..., 2:3]) + m[..., :3, 3]
    d = dirs_cam.reshape(1, P, 3)
    mr = cam2world.reshape(B, 1, 4, 4)
    dirs_w = _fma_chain3(mr[..., :3, 0], d[..., 0:1], mr[..., :3, 1], d[..., 1:2], mr[..., :3, 2], d[..., 2:3])
    origins = cam2world[:, :3, 3].clone()
    return pw.contiguous(), dirs_w.contiguous(), origins


def _fma32(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """float32 fma(a, b, c): the product of two float32 is exact in float64, one rounding to float32 at the end."""
    return (a.double() * b.double() + c.double()).float()


def _fma_chain3(m0, x, m1, y, m2, z):
    return _fma32(m2, z, _fma32(m1, y, m0 * x))


# ---------------------------------------------------------------------------------------
# a5  trilinear lookup           (siren.py:555-571; ATen grid_sampler_3d, bilinear/border/align_corners=False)
# ---------------------------------------------------------------------------------------
def trilinear_lookup(fvol_cf: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """fvol_cf (B,C,V,V,V) channel-first, points (B,N,3) world -> (B,N,C).  The ATen op itself."""
    B, N, _ = points.shape
    grid = (points / (VOXEL_LENGTH / 2)).reshape(B, 1, 1, N, 3)
    out = F.grid_sample(fvol_cf, grid, mode="bilinear", align_corners=False, padding_mode="border")
    return out.reshape(B, fvol_cf.shape[1], N).permute(0, 2, 1).contiguous()


def trilinear_corners(points: torch.Tensor, V: int):
    """Corner indices and weights, spelled out.  grid[...,0]=x -> W (last dim), 1=y -> H, 2=z -> D.

    Returns (i0 (B,N,3) int64 = floor of the clamped unnormalised coordinate in x,y,z order,
             frac weights as the 8-tuple in the accumulation order the ATen CPU kernel uses).
    """
    g = points / (VOXEL_LENGTH / 2)
    ic = ((g + 1) * V - 1) / 2
    ic = torch.clamp(ic, 0, V - 1)                      # padding_mode="border"
    i0f = torch.floor(ic)
    lo = ic - i0f                                       # weight of the +1 corner
    hi = (i0f + 1) - ic                                 # weight of the floor corner
    return i0f.long(), lo, hi


def trilinear_lookup_explicit(fvol_cf: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """Same result as trilinear_lookup, from explicit indices, in the ATen accumulation order:
    corners (dz,dy,dx) = 000,001,010,011,100,101,110,111 with x fastest; a corner at index V is skipped."""
    B, C, V = fvol_cf.shape[0], fvol_cf.shape[1], fvol_cf.shape[-1]
    i0, lo, hi = trilinear_corners(points, V)
    N = points.shape[1]
    flat = fvol_cf.reshape(B, C, V * V * V)
    out = torch.zeros(B, N, C)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                wx = lo[..., 0] if dx else hi[..., 0]
                wy = lo[..., 1] if dy else hi[..., 1]
                wz = lo[..., 2] if dz else hi[..., 2]
                w = wx * wy * wz
                ix, iy, iz = i0[..., 0] + dx, i0[..., 1] + dy, i0[..., 2] + dz
                inb = (ix < V) & (iy < V) & (iz < V)
                lin = (iz.clamp(max=V - 1) * V + iy.clamp(max=V - 1)) * V + ix.clamp(max=V - 1)
                vals = torch.gather(flat, 2, lin.unsqueeze(1).expand(B, C, N)).permute(0, 2, 1)
                out = out + torch.where(inb.unsqueeze(-1), vals * w.unsqueeze(-1), torch.zeros(()))
    return out


# ---------------------------------------------------------------------------------------
# a6-a9  field network
# ---------------------------------------------------------------------------------------
def film_params(spec: FieldSpec, params: Dict[str, torch.Tensor], global_feature: Optional[torch.Tensor],
                H: int) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """freq, phase (B, n_film*H); freq already *15+30   (siren.py:550-553)."""
    if not spec.has_global:
        return None, None
    fo = F.linear(global_feature, params["mapping_network.weight"], params["mapping_network.bias"])
    half = fo.shape[-1] // 2
    return fo[..., :half] * 15 + 30, fo[..., half:]


def field_mlp(spec: FieldSpec, params: Dict[str, torch.Tensor], feats: torch.Tensor,
              global_feature: Optional[torch.Tensor], points: Optional[torch.Tensor] = None,
              drop: Optional[Tuple[float, torch.Tensor]] = None) -> torch.Tensor:
    """feats (B,N,C) -> rgb_sigma (B,N,4).  points (B,N,3) is needed by the input kinds that see the position.
    drop = (p, keep (n_drop, B, N, H) of 0/1): training-mode nn.Dropout(p) behind the sine of every FiLM / per-point FiLM / sine
    layer (siren.py:158-159,175-176,197-198; residual blocks have none) with the keep decisions given -- F.dropout is
    x * bernoulli(1 - p) / (1 - p) (ATen Dropout.cpp: noise.bernoulli_(1 - p).div_(1 - p); input * noise)."""
    H = params["final_layer.weight"].shape[1]
    n_dropped = 0

    def dropout(x):
        nonlocal n_dropped
        if drop is None:
            return x
        noise = drop[1][n_dropped].to(x.dtype) / (1 - drop[0])
        n_dropped += 1
        return x * noise
    freq, phase = film_params(spec, params, global_feature, H)
    x = feats
    if spec.input == "feat_xyz":
        x = torch.cat([feats, points], -1)
    elif spec.input == "xyz":
        # per-point FiLM parameters from the looked-up feature: Linear -> LeakyReLU(0.2) -> Linear (siren.py:81-101)
        mh = F.leaky_relu(F.linear(feats, params["mapping_network.network.0.weight"], params["mapping_network.network.0.bias"]), 0.2)
        fo = F.linear(mh, params["mapping_network.network.2.weight"], params["mapping_network.network.2.bias"])
        half = fo.shape[-1] // 2
        pfreq, pphase = fo[..., :half] * 15 + 30, fo[..., half:]
        x = points
    f = 0
    for i, kind in enumerate(spec.layers):
        pre = f"network.{i}."
        if kind == "film":
            x = F.linear(x, params[pre + "layer.weight"], params[pre + "layer.bias"])
            fr = freq[:, f * H:(f + 1) * H].unsqueeze(1)
            ph = phase[:, f * H:(f + 1) * H].unsqueeze(1)
            x = dropout(torch.sin(fr * x + ph))
            f += 1
        elif kind == "pfilm":
            x = F.linear(x, params[pre + "layer.weight"], params[pre + "layer.bias"])
            x = dropout(torch.sin(pfreq[..., i * H:(i + 1) * H] * x + pphase[..., i * H:(i + 1) * H]))
        elif kind == "sine":
            x = dropout(torch.sin(F.linear(x, params[pre + "layer.weight"], params[pre + "layer.bias"])))
        elif kind == "res":
            h = torch.sin(F.linear(x, params[pre + "fc1.weight"], params[pre + "fc1.bias"]))
            h = F.linear(h, params[pre + "fc2.weight"], params[pre + "fc2.bias"])
            x = torch.sin(x + h)
        else:
            raise ValueError(kind)
    out = F.linear(x, params["final_layer.weight"], params["final_layer.bias"])
    if spec.sigmoid_rgb:                                   # siren.py:1227-1234
        out = torch.cat([torch.sigmoid(out[..., :3]), out[..., 3:]], -1)
    return out


def field_eval(spec: FieldSpec, params, fvol_cf, global_feature, points, explicit_lookup=False, drop=None):
    """fvol_cf: one (B,C,V,V,V) volume, or a list of them for the "pyramid" input kind."""
    look = trilinear_lookup_explicit if explicit_lookup else trilinear_lookup
    if isinstance(fvol_cf, (list, tuple)):
        feats = torch.cat([look(v, points) for v in fvol_cf], -1)
    else:
        feats = look(fvol_cf, points)
    return field_mlp(spec, params, feats, global_feature, points, drop), feats


# ---------------------------------------------------------------------------------------
# a11  alpha compositing                  (volumetric_rendering.py:18-70)
# ---------------------------------------------------------------------------------------
def composite(rgb_sigma: torch.Tensor, z: torch.Tensor, eps: Optional[torch.Tensor], noise_std: float,
              clamp_mode: str, white_back: bool = False, last_back: bool = False):
    """rgb_sigma (B,P,S,4), z (B,P,S), eps (B,P,S) standard normal or None -> rgb (B,P,3), dist (B,P), weights (B,P,S)."""
    sigma = rgb_sigma[..., 3]
    delta = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], 1e10)], -1)
    noisy = sigma + (eps * noise_std if eps is not None else torch.zeros_like(sigma))
    if clamp_mode == "relu":
        dens = torch.relu(noisy)
    elif clamp_mode == "softplus":
        dens = F.softplus(noisy)
    else:
        raise TypeError("Need to choose clamp mode")       # the reference raises a str -> TypeError
    alpha = 1 - torch.exp(-delta * dens)
    shifted = torch.cat([torch.ones_like(alpha[..., :1]), 1 - alpha + 1e-10], -1)
    trans = torch.cumprod(shifted, -1)[..., :-1]           # exclusive product
    w = alpha * trans
    wsum = w.sum(-1)
    if last_back:
        w = w.clone()
        w[..., -1] += 1 - wsum
    rgb = (w.unsqueeze(-1) * rgb_sigma[..., :3]).sum(-2)
    dist = (w * z).sum(-1)
    if white_back:
        rgb = rgb + 1 - wsum.unsqueeze(-1)
    return rgb, dist, w


# ---------------------------------------------------------------------------------------
# a12  inverse-CDF resampling            (generators.py:123-137, volumetric_rendering.py:297-342)
# ---------------------------------------------------------------------------------------
def importance_depths(z: torch.Tensor, weights: torch.Tensor, u_fine: torch.Tensor):
    """z, weights, u_fine (B,P,S) -> fine_z (B,P,S), inds (B,P,S) int64, cdf (B,P,S-1)."""
    B, P, S = z.shape
    zz = z.reshape(B * P, S)
    w = weights.reshape(B * P, S) + 1e-5
    bins = 0.5 * (zz[:, :-1] + zz[:, 1:])                  # S-1 mid-points
    wi = w[:, 1:-1] + 1e-5                                 # S-2 interior weights, eps added twice in all
    pdf = wi / wi.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros(B * P, 1), torch.cumsum(pdf, -1)], -1)   # S-1 entries
    u = u_fine.reshape(B * P, S).contiguous()
    inds = torch.searchsorted(cdf, u)                      # first i with cdf[i] >= u
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=S - 2)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    fine = b0 + (u - c0) / den * (b1 - b0)
    return fine.reshape(B, P, S), inds.reshape(B, P, S), cdf.reshape(B, P, S - 1)


# ---------------------------------------------------------------------------------------
# a13  merge by depth                     (generators.py:162-167)
# ---------------------------------------------------------------------------------------
def merge_by_depth(fine_out, coarse_out, fine_z, coarse_z):
    """Concatenate [fine, coarse] and sort ascending by z.  Returns (all_out, all_z, sort_idx)."""
    all_out = torch.cat([fine_out, coarse_out], -2)
    all_z = torch.cat([fine_z, coarse_z], -1)
    all_z_sorted, idx = torch.sort(all_z, dim=-1)
    all_out = torch.gather(all_out, -2, idx.unsqueeze(-1).expand(-1, -1, -1, 4))
    return all_out, all_z_sorted, idx


# ---------------------------------------------------------------------------------------
# whole path                               (generators.py:33-187)
# ---------------------------------------------------------------------------------------
@dataclass
class RenderOut:
    pixels: torch.Tensor
    depth: torch.Tensor
    aux: Dict[str, torch.Tensor] = field(default_factory=dict)


def render(variant: str, params: Dict[str, torch.Tensor], fvol_cf: torch.Tensor,
           global_feature: Optional[torch.Tensor], cam2world: torch.Tensor, R: int, fov: float,
           ray_start: float, ray_end: float, S: int, hierarchical: bool, clamp_mode: str,
           noise_std: float, white_back: bool, last_back: bool, u_strat: torch.Tensor,
           eps_coarse: Optional[torch.Tensor] = None, u_fine: Optional[torch.Tensor] = None,
           eps_final: Optional[torch.Tensor] = None, explicit_lookup: bool = False,
           forced_fine_z: Optional[torch.Tensor] = None, drop_p: float = 0.0,
           drop_coarse: Optional[torch.Tensor] = None, drop_fine: Optional[torch.Tensor] = None) -> RenderOut:
    """forced_fine_z (B,P,S): test hook, replaces the resampled depths downstream of the resampling (teacher forcing,
    the twin of cnerf_rng.fine_z); inds / cdf / the resampled depths are still returned in aux.
    drop_p > 0 with drop_coarse / drop_fine (n_drop, B, P*S, H): the network in training mode with these dropout keep decisions
    in its two forward calls (twin of cnerf_cfg.drop_p / cnerf_rng.drop_coarse, drop_fine)."""
    spec = FIELD_SPECS[variant]
    B, P = cam2world.shape[0], R * R
    aux: Dict[str, torch.Tensor] = {}
    with torch.no_grad():                                   # generators.py:57-93: rays and samples carry no gradient
        dirs_cam = camera_ray_dirs(R, fov)
        z_lin, offset, z = stratified_depths(B, R, S, ray_start, ray_end, u_strat)
        pts, dirs_w, origins = coarse_world_points(cam2world, dirs_cam, z_lin, offset)
    c_out, c_feat = field_eval(spec, params, fvol_cf, global_feature, pts.reshape(B, P * S, 3), explicit_lookup,
                               (drop_p, drop_coarse) if drop_p > 0 else None)
    c_out = c_out.reshape(B, P, S, 4)
    aux.update(coarse_points=pts, coarse_z=z, coarse_feat=c_feat, coarse_rgb_sigma=c_out)
    if hierarchical:
        with torch.no_grad():                               # generators.py:111-142: resampling is not differentiated
            _, _, w = composite(c_out, z, eps_coarse, noise_std, clamp_mode)
            fine_z, inds, cdf = importance_depths(z, w, u_fine)
            aux["resampled_z"] = fine_z
            if forced_fine_z is not None:
                fine_z = forced_fine_z.reshape(B, P, S)
            fpts = origins.reshape(B, 1, 1, 3) + dirs_w.unsqueeze(2) * fine_z.unsqueeze(-1)
        f_out, _ = field_eval(spec, params, fvol_cf, global_feature, fpts.reshape(B, P * S, 3), explicit_lookup,
                              (drop_p, drop_fine) if drop_p > 0 else None)
        f_out = f_out.reshape(B, P, S, 4)
        all_out, all_z, sort_idx = merge_by_depth(f_out, c_out, fine_z, z)
        aux.update(coarse_weights=w, cdf=cdf, inds=inds, fine_z=fine_z, fine_points=fpts,
                   fine_rgb_sigma=f_out, sort_idx=sort_idx)
    else:
        all_out, all_z = c_out, z
    rgb, dist, wfin = composite(all_out, all_z, eps_final, noise_std, clamp_mode, white_back, last_back)
    aux.update(final_weights=wfin)
    pixels = rgb.reshape(B, R, R, 3).permute(0, 3, 1, 2).contiguous() * 2 - 1      # generators.py:182-183
    depth = (dirs_cam[:, 2].reshape(1, P) * dist).reshape(B, R, R)                  # volumetric_rendering.py:345-356
    return RenderOut(pixels, depth, aux)
