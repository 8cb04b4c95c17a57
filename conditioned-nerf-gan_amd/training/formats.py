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


# every key Trainer.save_models writes (utils.py:473-501) and Trainer.load_models reads back unconditionally
# (utils.py:318-336, 407-410): a file without one of them raises KeyError in the reference's loader
CHECKPOINT_KEYS = ("step", "generator_state_dict", "optimizer_G_state_dict", "scaler_state_dict", "encoder_state_dict",
                   "optimizer_E_state_dict", "photometry_losses_val", "depth_losses_val", "photometry_losses_test",
                   "depth_losses_test")
DISCRIMINATOR_KEYS = ("discriminator_state_dict", "optimizer_D_state_dict", "generator_losses", "discriminator_losses")


def _fresh_scaler_state():
    """state_dict() of an untouched torch.cuda.amp.GradScaler (what the reference's loader hands to scaler.load_state_dict,
    utils.py:336).  This harness trains in fp32 and has no scaler; the dict is spelled out because constructing a GradScaler
    on a machine without a GPU warns and disables itself (its state_dict is then empty)."""
    return {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 0}


def save_checkpoint(trainer, directory):
    """`<step>.tar` with the key set of the reference's Trainer.save_models (utils.py:473-501), so that either code base can
    resume the other's run: the reference's loader reads `scaler_state_dict` and the four validation / test loss histories
    unconditionally, so they are written too (a fresh GradScaler state -- this harness trains in fp32 -- and the histories
    the trainer carries, empty when it never evaluated)."""
    step = int(trainer.generator.step)
    hist = getattr(trainer, "eval_losses", {})
    ck = {"step": step,
          "generator_state_dict": trainer.generator.state_dict(),
          "optimizer_G_state_dict": trainer.optimizer_G.state_dict(),
          "scaler_state_dict": _fresh_scaler_state(),
          "encoder_state_dict": trainer.encoder.state_dict(),
          "optimizer_E_state_dict": trainer.optimizer_E.state_dict(),
          "photometry_losses": list(trainer.losses["photo"]),
          "photometry_losses_val": list(hist.get("photometry_losses_val", [])),
          "depth_losses_val": list(hist.get("depth_losses_val", [])),
          "photometry_losses_test": list(hist.get("photometry_losses_test", [])),
          "depth_losses_test": list(hist.get("depth_losses_test", []))}
    if trainer.metadata.get("enable_discriminator", True):
        ck["discriminator_state_dict"] = trainer.discriminator.state_dict()
        ck["optimizer_D_state_dict"] = trainer.optimizer_D.state_dict()
        ck["generator_losses"] = list(trainer.losses["g"])
        ck["discriminator_losses"] = list(trainer.losses["d"])
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{step}.tar")
    torch.save(ck, path)
    return path


def load_checkpoint(trainer, path, map_location=None):
    """Resume from a `<step>.tar` written by save_checkpoint or by the reference's Trainer.save_models (same keys; the AMP
    scaler state of a reference file is ignored: fp32 training).  Model / optimizer states, the step counter and the loss
    histories are restored.  Loaded with weights_only=True: nothing in the file is executed."""
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
    trainer.losses = {"g": list(ck.get("generator_losses", [])), "d": list(ck.get("discriminator_losses", [])),
                      "photo": list(ck.get("photometry_losses", []))}
    trainer.eval_losses = {k: list(ck.get(k, [])) for k in ("photometry_losses_val", "depth_losses_val", "photometry_losses_test",
                                                            "depth_losses_test")}
    return ck
