"""Caller-side helpers of the render path.

The per-ray arithmetic of the reference's generators/volumetric_rendering.py (rays, jitter, compositing, inverse-CDF
resampling) lives in the HIP kernels; what stays here is (a) the camera sampling the trainer does on the host
(volumetric_rendering.py:212-287) and (b) function-level entry points with the reference's names that run the
corresponding stage kernel, for callers and tests that use the stages on their own."""
import numpy as np
import torch

from .math_utils_torch import normalize_vecs
from .. import ops


def sample_camera_positions(device, up_direction, cam_r_start=0, cam_r_end=1, n=1):
    """n camera origins, uniform on spherical shells r in [cam_r_start, cam_r_end] (NumPy RNG, like the reference:
    theta = arccos(1 - U) clipped away from the poles, phi = 2 pi U, r uniform)."""
    assert up_direction in ["y", "z"]
    theta = np.clip(np.arccos(1 - np.random.rand(n)), 1e-5, np.pi - 1e-5)
    phi = np.random.rand(n) * np.pi * 2
    r = np.random.rand(n) * (cam_r_end - cam_r_start) + cam_r_start
    horiz, vert = r * np.sin(theta) * np.sin(phi), r * np.cos(theta)
    origin = np.zeros((n, 3))
    origin[:, 0] = r * np.sin(theta) * np.cos(phi)
    origin[:, 1], origin[:, 2] = (horiz, vert) if up_direction == "z" else (vert, horiz)
    return torch.from_numpy(origin).type(torch.float32).to(device)


def create_cam2world_matrix(origin, up_direction, device=None):
    """Look-at-the-world-origin camera: rotation columns (-left, -up, forward), translation = origin."""
    assert up_direction in ["y", "z"]
    forward = normalize_vecs(-origin)
    axis = [0.0, 1.0, 0.0] if up_direction == "y" else [0.0, 0.0, 1.0]
    up0 = torch.tensor(axis, dtype=torch.float, device=device).expand_as(forward)
    left = normalize_vecs(torch.cross(up0, forward, dim=-1))
    up = normalize_vecs(torch.cross(forward, left, dim=-1))
    n = forward.shape[0]
    rot = torch.eye(4, device=device).unsqueeze(0).repeat(n, 1, 1)
    rot[:, :3, :3] = torch.stack((-left, -up, forward), dim=-1)
    trans = torch.eye(4, device=device).unsqueeze(0).repeat(n, 1, 1)
    trans[:, :3, 3] = origin
    return trans @ rot


def fancy_integration(rgb_sigma, z_vals, device=None, noise_std=0.5, last_back=False, white_back=False,
                      clamp_mode=None, fill_mode=None, noise=None):
    """NeRF compositing on (B,P,S,4)/(B,P,S,1) tensors -> rgb (B,P,3), depth (B,P,1), weights (B,P,S,1).
    `noise` injects the standard-normal draw (default: torch.randn, like the reference)."""
    B, P, S = rgb_sigma.shape[:3]
    if noise is None:
        noise = torch.randn((B, P, S, 1), device=rgb_sigma.device)
    rgb, dist, w = ops.composite(rgb_sigma.reshape(B * P, S, 4), z_vals.reshape(B * P, S),
                                 noise.reshape(B * P, S) if noise_std != 0 else None, noise_std, clamp_mode, white_back,
                                 last_back)
    rgb = rgb.reshape(B, P, 3)
    if fill_mode is not None:
        # debug paints of the reference (volumetric_rendering.py:62-67), on the composited result: `weights_sum` there is taken
        # BEFORE the last_back correction, i.e. the plain sum of alpha * transmittance
        wsum = w.reshape(B, P, S).sum(-1, keepdim=True)
        if last_back:
            wsum = wsum - (w.reshape(B, P, S)[..., -1:] - _last_weight_before_back(rgb_sigma, z_vals, noise, noise_std, clamp_mode))
        if fill_mode == "debug":
            rgb = torch.where(wsum < 0.9, torch.tensor([1.0, 0.0, 0.0], device=rgb.device).expand_as(rgb), rgb)
        elif fill_mode == "weight":
            rgb = wsum.expand_as(rgb).clone()
    return rgb, dist.reshape(B, P, 1), w.reshape(B, P, S, 1)


def _last_weight_before_back(rgb_sigma, z_vals, noise, noise_std, clamp_mode):
    """Weight of the last sample without the last_back correction (one more composite call with the flag off)."""
    B, P, S = rgb_sigma.shape[:3]
    _, _, w = ops.composite(rgb_sigma.reshape(B * P, S, 4), z_vals.reshape(B * P, S), noise.reshape(B * P, S) if noise_std != 0 else None,
                            noise_std, clamp_mode, False, False)
    return w.reshape(B, P, S)[..., -1:]


def importance_sample(z_vals, weights, u):
    """The resampling block of ImplicitGenerator3d.forward (generators.py:123-137) on (rays,S) tensors."""
    return ops.resample(z_vals, weights, u)


def distance2depth(distance, ray):
    """Distance along the ray -> camera-space z (volumetric_rendering.py:345-356)."""
    return ray[..., -1:] * distance
