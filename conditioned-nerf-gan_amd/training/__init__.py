"""Thin training harness around the render path: the voxel encoder and the discriminator stay in
plain PyTorch-ROCm, the generator is the HIP render path, data parallelism is DDP over RCCL."""
from .encoder import UNet3D  # noqa: F401
from .discriminator import ProgressiveDiscriminator, CCSDiscriminator  # noqa: F401
from .gan_step import GanTrainer, default_metadata  # noqa: F401
from .formats import load_voxel_npz, load_cam2world, save_checkpoint, load_checkpoint  # noqa: F401
