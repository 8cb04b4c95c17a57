"""Field networks of the render path: parameter holders with the reference's names, shapes and initialisation
(generators/siren.py), evaluated by the fused gfx950 kernel through the C ABI (cnerf_field_forward /
cnerf_render_forward).  One table describes every supported variant; the classes are generated from it, so
`getattr(siren, siren_type)` (generators.py:15 in the reference) keeps working and reference checkpoints load
(state-dict keys network.{i}.layer.*, network.{i}.fc1/fc2.*, final_layer.*, mapping_network.*).
"""
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import ops

VOXEL_LENGTH = 1.2  # siren.py:555 of the reference: the voxel grid spans the 1.2^3 cube


@dataclass(frozen=True)
class FieldSpec:
    layers: Tuple[str, ...]     # "film" | "sine" | "res"                      (siren.py:146-230)
    init_freq: float            # frequency_init(f): U(+-sqrt(6/in)/f)          (siren.py:134-143)
    sigmoid_rgb: bool           # _sigmoid_rgb on the head                      (siren.py:1227-1234)
    has_global: bool            # z = (feature_volume, global_feature)
    input_is_zdim: bool = False  # the dRes family overrides input_dim = z_dim  (siren.py:349)
    input: str = "feat"         # "feat" | "feat_xyz" (siren.py:1158) | "pyramid" (siren.py:1444-1473)


FIELD_SPECS = {
    "SHORTSIREN_FG": FieldSpec(("film",) * 4, 12, True, True),
    "TALLSIREN_FG": FieldSpec(("film",) * 8, 25, True, True),
    "DOUBLESIREN_FG": FieldSpec(("film",) * 2, 12, True, True),
    "SingleSIREN_dg": FieldSpec(("film",), 25, False, True),
    "SHORTSIREN_F": FieldSpec(("sine",) * 4, 12, True, False),
    "SHORTSIREN_FRes": FieldSpec(("sine", "res", "sine"), 12, True, False),
    "TALLSIREN_dRes": FieldSpec(("sine", "res", "res", "sine"), 25, False, False, True),
    "TALLSIREN_dResLong": FieldSpec(("sine",) + ("res",) * 4 + ("sine",), 25, False, False, True),
    "TALLSIREN_dgx": FieldSpec(("film",) * 8, 25, False, True, False, "feat_xyz"),
    "SHORTSIREN_FG_Pyrmd": FieldSpec(("film",) * 4, 12, True, True, False, "pyramid"),
}


class _LinearSine(nn.Module):
    """Holder for one FiLM / plain-sine layer: `.layer` is the nn.Linear, like FiLMLayer / SirenLayer."""

    def __init__(self, n_in, n_out, drop_out_prob=0):
        super().__init__()
        self.layer = nn.Linear(n_in, n_out)
        self.dropout_layer = nn.Dropout(drop_out_prob)
        self.drop_out_prob = drop_out_prob


class FiLMLayer(_LinearSine):
    kind = "film"


class SirenLayer(_LinearSine):
    kind = "sine"


class ResSirenBlock(nn.Module):
    """Holder for sin(x + fc2(sin(fc1 x))) (siren.py:218-230)."""
    kind = "res"

    def __init__(self, hidden_dim):
        super().__init__()
        self.fc1 = nn.Linear(hidden_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)


def _uniform_weights(module: nn.Module, bound_of_fan_in):
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, nn.Linear):
                b = bound_of_fan_in(m.weight.size(-1))
                m.weight.uniform_(-b, b)


class FieldNetwork(nn.Module):
    """Base of every generated variant.  `forward(points, z, img_size, num_steps)` is the siren sub-API
    (extract_shapes.py:63-69): points (B,N,3) world -> rgb_sigma (B,N,4)."""

    spec: FieldSpec = None
    variant: str = None
    _drop_calls = 0          # forwards of this module that drew dropout decisions (Philox counter offset)

    def __init__(self, input_dim=3, z_dim=100, hidden_dim=256, output_dim=4, drop_out=0, device=None, **kwargs):
        super().__init__()
        spec = self.spec
        if spec.input_is_zdim:
            input_dim = z_dim
        self.device = device
        self.input_dim, self.z_dim, self.hidden_dim, self.output_dim = input_dim, z_dim, hidden_dim, output_dim
        self.drop_out = drop_out
        # construction order = the reference's, so the same torch seed yields the same parameters
        blocks = []
        for i, kind in enumerate(spec.layers):
            n_in = input_dim if i == 0 else hidden_dim
            if kind == "film":
                blocks.append(FiLMLayer(n_in, hidden_dim, drop_out))
            elif kind == "sine":
                blocks.append(SirenLayer(n_in, hidden_dim, drop_out))
            else:
                blocks.append(ResSirenBlock(hidden_dim))
        self.network = nn.ModuleList(blocks)
        self.final_layer = nn.Linear(hidden_dim, 4)
        if spec.has_global:
            self.mapping_network = nn.Linear(z_dim, len(self.network) * hidden_dim * 2)
        f = spec.init_freq
        _uniform_weights(self.network, lambda n: math.sqrt(6 / n) / f)
        _uniform_weights(self.final_layer, lambda n: math.sqrt(6 / n) / f)
        _uniform_weights(self.network[0], lambda n: 1 / n)   # first_layer_film_sine_init (siren.py:40-44)

    # -- pieces the renderer needs -------------------------------------------------------------------------------
    def split_z(self, z):
        """-> (feature volume (B,C,V,V,V) or list of pyramid levels, global_feature or None)."""
        if self.spec.has_global:
            fvol, glob = z
            return fvol, glob
        return z, None

    def film(self, global_feature: Optional[torch.Tensor]):
        """freq, phase (B, n_film*H), freq already *15+30 (siren.py:645-650).  Stays in PyTorch (tiny GEMM, K6)."""
        if not self.spec.has_global:
            return None, None
        fo = self.mapping_network(global_feature)
        half = fo.shape[-1] // 2
        return (fo[..., :half] * 15 + 30).contiguous(), fo[..., half:].contiguous()

    def field_params(self):
        """Flat list of raw parameter tensors in the order ops.pack_field expects."""
        out = []
        for blk in self.network:
            if isinstance(blk, ResSirenBlock):
                out += [blk.fc1.weight, blk.fc1.bias, blk.fc2.weight, blk.fc2.bias]
            else:
                out += [blk.layer.weight, blk.layer.bias]
        out += [self.final_layer.weight, self.final_layer.bias]
        return out

    def check_supported(self):
        """Dropout (training mode, drop_out > 0: siren.py:158-159,175-176,197-198) runs in the fp32 kernels only."""
        if self.drop_out and self.training and (ops.precision_of(self) != "fp32" or ops.backward_precision_of(self) != "fp32"):
            raise NotImplementedError("drop_out > 0 in training mode needs precision = backward_precision = 'fp32'")

    def forward(self, points, z, img_size=None, num_steps=None):
        self.check_supported()
        fvol, glob = self.split_z(z)
        freq, phase = self.film(glob)
        drop = None
        if self.drop_out and self.training:     # keep decisions: Philox under torch's CUDA seed, one counter value per call
            drop = (float(self.drop_out), (torch.cuda.initial_seed(), self._drop_calls))
            self._drop_calls += 1
        return ops.field_forward(self, fvol, freq, phase, points, drop=drop)


class PointFeaturesMappingNetwork(nn.Module):
    """Per-point FiLM parameters from the looked-up feature: Linear -> LeakyReLU(0.2) -> Linear, kaiming-leaky init, last
    layer scaled by 0.25 (siren.py:47-52, 81-101).  Parameter holder; evaluated inside the field kernel."""

    def __init__(self, z_dim, map_hidden_dim, map_output_dim):
        super().__init__()
        self.network = nn.Sequential(nn.Linear(z_dim, map_hidden_dim), nn.LeakyReLU(0.2, inplace=True),
                                     nn.Linear(map_hidden_dim, map_output_dim))
        for m in self.network:
            if isinstance(m, nn.Linear):
                torch.nn.init.kaiming_normal_(m.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        with torch.no_grad():
            self.network[-1].weight *= 0.25


class PointwiseFiLMLayer(_LinearSine):
    kind = "pfilm"


class TALLSIREN(FieldNetwork):
    """pi-GAN style field (siren.py:232-331): input = world xyz, eight FiLM layers whose frequencies / phases come per
    POINT from a mapping MLP of the looked-up feature (z is the bare feature volume, z_dim = its channel count).
    Parameters, names and initialisation mirror the reference so its checkpoints load.  Forward: field_pw_kernel (fp32) or
    field_pw16_kernel (fp16x3 / fp16); backward: storing forward + field_pw_backward_kernel + library GEMMs (fp32,
    ops._pfilm_backward) or chain_pre_kernel + pw_gm_kernel + weight_grad16 behind cnerf_render_backward (backward_precision "fp16")."""
    variant = "TALLSIREN"
    spec = FieldSpec(("pfilm",) * 8, 25, False, False, False, "xyz")

    def __init__(self, input_dim=3, z_dim=100, hidden_dim=256, output_dim=4, drop_out=0, device=None, **kwargs):
        nn.Module.__init__(self)
        self.device = device
        self.input_dim, self.z_dim, self.hidden_dim, self.output_dim = input_dim, z_dim, hidden_dim, output_dim
        self.drop_out = drop_out
        self.network = nn.ModuleList([PointwiseFiLMLayer(input_dim if i == 0 else hidden_dim, hidden_dim, drop_out)
                                      for i in range(8)])
        self.final_layer = nn.Linear(hidden_dim, 4)
        self.mapping_network = PointFeaturesMappingNetwork(z_dim, 256, len(self.network) * hidden_dim * 2)
        _uniform_weights(self.network, lambda n: math.sqrt(6 / n) / 25)
        _uniform_weights(self.final_layer, lambda n: math.sqrt(6 / n) / 25)
        _uniform_weights(self.network[0], lambda n: 1 / n)

    def field_params(self):
        mp = self.mapping_network.network
        out = [mp[0].weight, mp[0].bias, mp[2].weight, mp[2].bias]
        for blk in self.network:
            out += [blk.layer.weight, blk.layer.bias]
        return out + [self.final_layer.weight, self.final_layer.bias]

    def check_supported(self):
        super().check_supported()
        if self.input_dim != 3:
            raise NotImplementedError("TALLSIREN reads the world position: input_dim must be 3")


def _make(name):
    return type(name, (FieldNetwork,), {"spec": FIELD_SPECS[name], "variant": name, "__doc__": f"{name} field network"})


for _n in FIELD_SPECS:
    globals()[_n] = _make(_n)
del _n
