"""On-disk formats of the reference, restated for the MI355X harness: the voxel grid and camera files of a
ShapeNet object directory (datasets.py:104-148) and the training checkpoint dictionary (utils.py:463-501).  No dataset
ships with either repository; tests/test_training_cpu.py round-trips synthetic files."""
import os

import numpy as np
import torch


def load_voxel_npz(path, resolution=64):
    """`voxel.npz["voxel"]` is stored (X, Y, Z, 4) with channels [occupancy, r, g, b]; the encoder wants (4, Z, Y, X) float32
    (datasets.py:131-139: `.permute(3, 2, 1, 0)`).  Resolution 32 lives in `voxel_32.npz` there; pass the file you mean."""
    with np.load(path, allow_pickle=False) as f:
        vox = torch.from_numpy(np.ascontiguousarray(f["voxel"]))
    if vox.dim() != 4 or vox.shape[-1] != 4:
        raise ValueError(f"{path}: expected an (X,Y,Z,4) array under 'voxel', got {tuple(vox.shape)}")
    if vox.shape[0] != resolution:
        raise ValueError(f"{path}: voxel resolution {vox.shape[0]} != {resolution}")
    return vox.permute(3, 2, 1, 0).float().contiguous()


def load_cam2world(cameras_npz, view):
    """cameras.npz holds one 4x4 camera-to-world matrix per rendered view under `world_mat_inv_<view>` (datasets.py:105-110)."""
    with np.load(cameras_npz, allow_pickle=False) as f:
        m = torch.from_numpy(np.asarray(f[f"world_mat_inv_{int(view)}"])).float()
    if m.shape != (4, 4):
        raise ValueError(f"{cameras_npz}: world_mat_inv_{view} has shape {tuple(m.shape)}")
    return m


CHECKPOINT_KEYS = ("step", "generator_state_dict", "optimizer_G_state_dict", "encoder_state_dict", "optimizer_E_state_dict",
                   "discriminator_state_dict", "optimizer_D_state_dict")


def save_checkpoint(trainer, directory):
    """`<step>.tar` with the reference's key names (utils.py:473-501), so either code base can resume the other's run.  The
    reference also stores its AMP `scaler_state_dict` and loss histories; this harness trains in fp32 and keeps the losses."""
    step = int(trainer.generator.step)
    ck = {"step": step,
          "generator_state_dict": trainer.generator.state_dict(),
          "optimizer_G_state_dict": trainer.optimizer_G.state_dict(),
          "encoder_state_dict": trainer.encoder.state_dict(),
          "optimizer_E_state_dict": trainer.optimizer_E.state_dict(),
          "generator_losses": list(trainer.losses["g"]),
          "discriminator_losses": list(trainer.losses["d"]),
          "photometry_losses": list(trainer.losses["photo"])}
    if trainer.metadata.get("enable_discriminator", True):
        ck["discriminator_state_dict"] = trainer.discriminator.state_dict()
        ck["optimizer_D_state_dict"] = trainer.optimizer_D.state_dict()
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{step}.tar")
    torch.save(ck, path)
    return path


def load_checkpoint(trainer, path, map_location=None):
    """Resume from a `<step>.tar` written by save_checkpoint or by the reference's Trainer.save_models (utils.py:296-330 reads
    the same keys back).  Loaded with weights_only=True: nothing in the file is executed."""
    ck = torch.load(path, map_location=map_location or trainer.device, weights_only=True)
    trainer.generator.load_state_dict(ck["generator_state_dict"], strict=True)
    trainer.encoder.load_state_dict(ck["encoder_state_dict"], strict=True)
    trainer.optimizer_G.load_state_dict(ck["optimizer_G_state_dict"])
    trainer.optimizer_E.load_state_dict(ck["optimizer_E_state_dict"])
    if "discriminator_state_dict" in ck and trainer.metadata.get("enable_discriminator", True):
        trainer.discriminator.load_state_dict(ck["discriminator_state_dict"], strict=True)
        trainer.optimizer_D.load_state_dict(ck["optimizer_D_state_dict"])
    trainer.generator.step = int(ck["step"])          # (utils.py:318: the reference resumes at the stored step as well)
    trainer.discriminator.step = int(ck["step"])
    return ck
