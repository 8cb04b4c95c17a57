"""MI355X-native render path of the conditioned-NeRF GAN (drop-in for generators/volumetric_rendering.py +
generators/siren.py of zzhuolun/conditioned-nerf-gan behind the ImplicitGenerator3d API).

The directory name contains hyphens, so import it through the alias module at the repo root:

    import cnerf_amd
    from cnerf_amd.generators import ImplicitGenerator3d
"""
from . import _lib, ops  # noqa: F401
from .generators import ImplicitGenerator3d  # noqa: F401

__all__ = ["ImplicitGenerator3d", "ops", "generators"]
