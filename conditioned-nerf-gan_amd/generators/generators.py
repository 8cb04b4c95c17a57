"""ImplicitGenerator3d: the generator API of the reference (generators/generators.py:9-197), backed by the gfx950
render path.  Same constructor, attributes, forward signature and state-dict keys, so the reference's
train.py / inference.py / extract_shapes.py call sites work unchanged:

    pixels, depth_map = generator(z, cam2worlds, img_size, fov, ray_start, ray_end, num_steps,
                                  hierarchical_sample, **metadata)
"""
import torch
import torch.nn as nn

from . import siren
from .. import ops


def draw_rng(B, P, S, hierarchical, noise_std, dev):
    """The random tensors of one forward, drawn with the reference's shapes and in its order (rand (B,P,S,1),
    randn (B,P,S,1), rand (B*P,S), randn (B,P,2S,1)), so the torch generator advances exactly as it does there; the noise
    tensors are drawn even when nerf_noise == 0 (volumetric_rendering.py:39) and only handed on when they are used."""
    rng = {"u_strat": torch.rand((B, P, S, 1), device=dev)}
    if hierarchical:
        eps_c = torch.randn((B, P, S, 1), device=dev)
        rng["u_fine"] = torch.rand((B * P, S), device=dev)
        eps_f = torch.randn((B, P, 2 * S, 1), device=dev)
        if noise_std != 0:
            rng["eps_coarse"], rng["eps_final"] = eps_c, eps_f
    else:
        eps_f = torch.randn((B, P, S, 1), device=dev)
        if noise_std != 0:
            rng["eps_final"] = eps_f
    return rng


class ImplicitGenerator3d(nn.Module):
    _instances = 0

    def _philox_key(self):
        """(seed, offset) of this forward's in-kernel draws: see __init__."""
        if self._rng_step != self.step:
            self._rng_step, self._rng_calls = self.step, 0
        offset = ((int(self.step) & 0xFFFFFF) << 8) | (self._rng_calls & 0xFF)
        self._rng_calls += 1
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        seed = (torch.cuda.initial_seed() + 0x9E3779B97F4A7C15 * (self._rng_salt + 1) + 0xD1B54A32D192ED03 * rank) & 0xFFFFFFFFFFFFFFFF
        return seed, offset

    def __init__(self, siren_type, z_dim, input_dim, output_dim, hidden_dim, drop_out=0):
        super().__init__()
        self.z_dim = z_dim
        field_cls = getattr(siren, siren_type)
        self.siren = field_cls(z_dim=z_dim, input_dim=input_dim, output_dim=output_dim, hidden_dim=hidden_dim,
                               drop_out=drop_out, device=None)
        self.epoch = 0
        self.step = 0
        self.device = None
        # where the four random draws of a forward come from: "torch" (torch.rand / randn on the device in the reference's
        # order and shapes -- the default: the torch generator advances as it does there) or "philox" (in-kernel)
        self.rng_mode = "torch"
        # Key and counter of the in-kernel draws (rng_mode "philox", dropout decisions).  Key = torch's CUDA seed mixed with a salt that
        # is different for every generator of the process (a teacher and a student under one seed must not draw the same streams) and
        # with the rank of a distributed job (ranks seeded alike still draw their own).  Counter offset = (self.step, calls within
        # that step): `step` is what checkpoints store (utils.py:467), so a resumed run continues the stream instead of replaying the
        # draws of steps 0..k -- nothing else has to be persisted.
        self._rng_salt = ImplicitGenerator3d._instances
        ImplicitGenerator3d._instances += 1
        self._rng_step, self._rng_calls = None, 0

    def set_device(self, device):
        self.device = device
        self.siren.device = device

    def forward(self, z, cam2worlds, img_size, fov, ray_start, ray_end, num_steps, hierarchical_sample, **kwargs):
        """z: feature volume (B,C,V,V,V) or (feature volume, global feature (B,z_dim)).
        kwargs read: clamp_mode, nerf_noise (required, like the reference), white_back, last_back; every other key
        of the splatted metadata dict is ignored.  `_rng` (dict of tensors) injects the four random draws and
        `_aux` (dict) receives intermediates, `_field_events` (4 hipEvent_t handles) times the field kernel -- test and
        benchmark hooks the reference does not have.
        Returns pixels (B,3,R,R) = 2*rgb-1 and depth_map (B,R,R)."""
        net = self.siren
        net.check_supported()
        clamp_mode, noise_std = kwargs["clamp_mode"], kwargs["nerf_noise"]
        white_back, last_back = kwargs.get("white_back", False), kwargs.get("last_back", False)
        fvol, glob = net.split_z(z)
        B, R, S = cam2worlds.shape[0], int(img_size), int(num_steps)
        dev = cam2worlds.device
        freq, phase = net.film(glob)
        rng = kwargs.get("_rng")
        if rng is None:
            if self.rng_mode == "philox":
                # draws generated inside the kernels (Philox4x32-10): no RNG kernels, no tensors; keyed per module, one counter value
                # per forward (see __init__)
                rng = {"philox": self._philox_key()}
            else:
                rng = draw_rng(B, R * R, S, bool(hierarchical_sample), noise_std, dev)
        if net.drop_out and net.training and "drop" not in rng:
            # dropout decisions are Philox draws inside the field kernels (or rng["drop_coarse"/"drop_fine"] bytes), keyed like
            # the philox rng mode
            rng = dict(rng, drop=(float(net.drop_out), rng.get("philox") or self._philox_key()))
        aux_out = kwargs.get("_aux")
        pixels, depth, aux = ops.render(net, fvol, freq, phase, cam2worlds, R, fov, ray_start, ray_end, S,
                                        bool(hierarchical_sample), clamp_mode, noise_std, white_back, last_back, rng,
                                        want_aux=aux_out is not None, field_events=kwargs.get("_field_events"))
        if aux_out is not None:
            aux_out.update(aux)
        return pixels, depth

    def generate_avg_frequencies(self):
        """Mean FiLM frequencies / phase shifts over 10000 random latents (generators.py:189-197)."""
        zs = torch.randn((10000, self.z_dim), device=self.siren.device)
        with torch.no_grad():
            fo = self.siren.mapping_network(zs)
        half = fo.shape[-1] // 2
        self.avg_frequencies = fo[..., :half].mean(0, keepdim=True)
        self.avg_phase_shifts = fo[..., half:].mean(0, keepdim=True)
        return self.avg_frequencies, self.avg_phase_shifts
