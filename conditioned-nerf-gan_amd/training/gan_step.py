"""One GAN step around the HIP generator (counterpart of Trainer.train_discriminator / train_generator / set_alpha,
utils.py:610-842 of the reference, which is not importable: F4 of SURVEY.md).  fp32 by default; where the reference wraps
the step in CUDA AMP (utils.py:327,643) this harness has explicit switches instead -- metadata["render_precision"] /
["backward_precision"] for the render path, ["encoder_autocast"] for the U-Net.  Adam(beta = (0, 0.9)), non-saturating logistic losses
softplus(+-pred), R1 penalty on real images, gradient clipping, `batch_split` gradient accumulation.  Under
torch.distributed every rank trains on its own images; DDP (backend "nccl" = RCCL over xGMI) averages the gradients, and
all but the last accumulation chunk run under no_sync() so there is one all-reduce per optimizer step (the reference
all-reduces on every chunk)."""
import contextlib
import math

import torch
import torch.nn.functional as F
from torch.nn.parallel import DistributedDataParallel as DDP

import time

from ..generators import ImplicitGenerator3d
from ..generators.volumetric_rendering import create_cam2world_matrix, sample_camera_positions
from . import discriminator as discriminators
from .encoder import UNet3D


class PhaseTimer:
    """Wall time of the phases of a GAN step: event pairs on the current stream of a GPU (torch's stream is the one every kernel of
    the step -- MIOpen, rocBLAS, the render path through the C ABI -- is launched on), perf_counter on the CPU.  Nothing is
    synchronised while a step runs; summary() waits once and adds up the occurrences of each phase name."""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        self.spans = []

    def begin(self):
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        return time.perf_counter()

    def end(self, name, start):
        self.spans.append((name, start, self.begin()))

    def summary(self):
        if self.cuda:
            torch.cuda.synchronize()
        out = {}
        for name, a, b in self.spans:
            out[name] = out.get(name, 0.0) + (a.elapsed_time(b) if self.cuda else (b - a) * 1e3)
        self.spans = []
        return out


class CommMeter:
    """Communication hook of a DDP wrapper that counts what it sends: all-reduce calls (= gradient buckets) and bytes, and how
    often the LAST bucket went out (= all-reduce rounds: one per backward that synchronises).  The reduction itself is DDP's
    default (average over the process group: RCCL on a GPU job, gloo in the CPU tests)."""

    def __init__(self):
        self.calls = self.bytes = self.rounds = 0

    def hook(self, state, bucket):
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        buf = bucket.buffer()
        self.calls += 1
        self.bytes += buf.numel() * buf.element_size()
        self.rounds += 1 if bucket.is_last() else 0
        return default_hooks.allreduce_hook(state, bucket)

    def take(self):
        out = {"calls": self.calls, "bytes": self.bytes, "rounds": self.rounds}
        self.calls = self.bytes = self.rounds = 0
        return out


def default_metadata(img_size=128, num_steps=64, batch_size=8, batch_split=4, siren_type="SHORTSIREN_FG", hidden_dim=256):
    """The render / optimisation hyper-parameters of configs/thousand/{default,special}.py for the direct-feature-volume
    setting (unet3d encoder -> (feature volume, global feature) -> FG field networks).  Field networks without a global feature
    (TALLSIREN -- per-point FiLM from the looked-up feature, input = world position --, the plain-sine / residual families) take the bare
    feature volume: the encoder then returns it alone and z_dim is its channel count (siren.py:232-331, 333-488)."""
    from ..generators import siren as _siren
    spec = getattr(_siren, siren_type).spec
    gen = {"siren_type": siren_type, "z_dim": 256, "input_dim": 32, "output_dim": 4, "hidden_dim": hidden_dim}
    if not spec.has_global:
        gen.update(z_dim=32, input_dim=3 if spec.input == "xyz" else 32)
    return {
        "img_size": img_size, "num_steps": num_steps, "batch_size": batch_size, "batch_split": batch_split,
        "fov": 49.134342641202636, "ray_start": 0.25, "ray_end": 1.95, "cam_r_start": 0.7, "cam_r_end": 1.5,
        "white_back": True, "last_back": False, "clamp_mode": "relu", "hierarchical_sample": True, "nerf_noise": 1.0,
        "fade_steps": 2000, "r1_lambda": 10, "grad_clip": 1, "betas": (0.0, 0.9), "weight_decay": 0,
        "gen_lr": 5e-5, "disc_lr": 2e-4, "enc_lr": 5e-5, "photo_loss": True, "enable_discriminator": True,
        "random_gen_img": True,
        "generator": gen,
        "unet": {"in_channels": 4, "out_channels": 32, "f_maps": 32, "num_levels": 4, "return_global": bool(spec.has_global)},
    }


class GanTrainer:
    """metadata["discriminator"] names the class (ProgressiveDiscriminator, the reference's default, or CCSDiscriminator);
    `modules` lets a caller hand in ready-made generator / encoder / discriminator modules (tests: a CPU stand-in generator
    with the same call signature for the multi-rank gloo test; modules built from a fixture's parameters)."""

    def __init__(self, metadata, device, ddp=False, modules=None):
        self.metadata, self.device, self.ddp = metadata, device, ddp
        modules = modules or {}
        # MIOpen find mode for the 3-D convolutions of the encoder (and the discriminator): without it the immediate-mode
        # heuristic picks a naive weight-gradient solver for several Conv3d shapes -- encoder forward + backward of 2 voxel
        # grids 371 ms against 22 ms with the search (scripts/encoder_profile.py), i.e. 75 % of the whole GAN step.
        # Opt-in (train.py turns it on): the search itself takes minutes on a fresh machine, once per convolution shape.
        if metadata.get("miopen_find", False):
            torch.backends.cudnn.benchmark = True
        if "generator" in modules:
            self.generator = modules["generator"].to(device)
        else:
            self.generator = ImplicitGenerator3d(**metadata["generator"]).to(device)
            self.generator.set_device(device)
            # arithmetic of the forward render (both the no-grad D-step images and the G-step forward): "fp32" or the
            # fp32-accurate split "fp16x3" (same parity gate, 2.7x faster)
            self.generator.siren.precision = metadata.get("render_precision", "fp32")
            # arithmetic of the backward's gradient GEMMs: "fp32", or "fp16" (fp16 operands, fp32 sums -- the class of the
            # reference's own autocast training, 2x faster; with render_precision "fp16x3" the forward's activations are kept
            # for it instead of being re-computed)
            self.generator.siren.backward_precision = metadata.get("backward_precision", "fp32")
        self.encoder = (modules["encoder"] if "encoder" in modules else UNet3D(**metadata["unet"])).to(device)
        if metadata.get("encoder_channels_last", False):     # NDHWC convolutions (MIOpen); off by default: measured below
            self.encoder = self.encoder.to(memory_format=torch.channels_last_3d)
        if "discriminator" in modules:
            self.discriminator = modules["discriminator"].to(device)
        else:
            self.discriminator = getattr(discriminators, metadata.get("discriminator", "ProgressiveDiscriminator"))().to(device)
        ids = [device.index] if device.type == "cuda" else None          # (gloo / CPU ranks: no device ids)
        wrap = (lambda m, unused: DDP(m, device_ids=ids, find_unused_parameters=unused)) if ddp else (lambda m, unused: m)
        self.generator_ddp = wrap(self.generator, True)
        self.encoder_ddp = wrap(self.encoder, False)
        self.discriminator_ddp = wrap(self.discriminator, True)
        adam = lambda m, lr: torch.optim.Adam(m.parameters(), lr=lr, betas=metadata["betas"], weight_decay=metadata["weight_decay"])
        self.optimizer_G = adam(self.generator_ddp, metadata["gen_lr"])
        self.optimizer_E = adam(self.encoder_ddp, metadata["enc_lr"])
        self.optimizer_D = adam(self.discriminator_ddp, metadata["disc_lr"])
        self.alpha = 1.0
        self.losses = {"d": [], "g": [], "photo": []}
        self._z = {}            # chunk index -> encoder output kept between the D and the G pass of one step
        self.last = {}          # diagnostics of the most recent step: loss terms and pre-clip gradient norms
        self.render_rng = None  # test hook: callable(chunk_index, phase) -> dict of injected draws for that render
        self.timer = None       # a PhaseTimer while a caller wants the step split into phases (bench.py), else None
        self.comm = {}          # name -> CommMeter once attach_comm_meters() ran (DDP only)

    def attach_comm_meters(self):
        """Counts the all-reduces of the three DDP wrappers from here on (bench.py, tests); {} without DDP."""
        if self.ddp and not self.comm:
            for name, m in (("generator", self.generator_ddp), ("encoder", self.encoder_ddp), ("discriminator", self.discriminator_ddp)):
                self.comm[name] = CommMeter()
                m.register_comm_hook(None, self.comm[name].hook)
        return self.comm

    @contextlib.contextmanager
    def _phase(self, name):
        if self.timer is None:
            yield
            return
        t0 = self.timer.begin()
        try:
            yield
        finally:
            self.timer.end(name, t0)

    def _disc(self, imgs, frozen=False):
        """frozen: the G step's use of the discriminator -- only d(prediction)/d(images) is needed there (the reference lets the
        backward fill the discriminator's parameter gradients too and zeroes them afterwards, utils.py:665-741).  The raw module
        runs with its parameters' requires_grad off: no weight-gradient kernels, and under DDP no all-reduce of 47 MiB of
        gradients nobody reads.  Same images' gradients, same step."""
        if frozen:
            out = self.discriminator(imgs, self.alpha, **self.metadata)
        else:
            out = self.discriminator_ddp(imgs, self.alpha, **self.metadata)
        return out[0] if isinstance(out, tuple) else out       # (CCSDiscriminator returns (prediction, None, None))

    # utils.py:610-618
    def set_alpha(self, step_last_upsample=0):
        step = self.generator.step
        self.alpha = min(1.0, (step - step_last_upsample) / self.metadata["fade_steps"]) if self.metadata["fade_steps"] > 0 else 1.0
        self.metadata["nerf_noise"] = max(0.0, 1.0 - step / 5000.0)

    def _encode(self, voxels):
        """The 3D U-Net.  metadata["encoder_autocast"] = "bf16" / "fp16" runs its convolutions under torch.autocast, as the
        reference's GPU training does with the whole step (utils.py:327,643: autocast + GradScaler; bf16 needs no scaler); the
        render path takes fp32 volumes either way."""
        amp = self.metadata.get("encoder_autocast")
        if not amp:
            return self.encoder_ddp(voxels)
        with torch.autocast(self.device.type, dtype=torch.bfloat16 if amp == "bf16" else torch.float16):
            out = self.encoder_ddp(voxels)
        return tuple(o.float() for o in out) if isinstance(out, (tuple, list)) else out.float()

    def _render(self, voxels, cams, chunk=0, phase="g"):
        z = self._z.get(chunk) if phase == "d" else self._z.pop(chunk, None)      # encoder output kept by step() (see there)
        if z is None:
            z = self._encode(voxels)
        extra = {"_rng": self.render_rng(chunk, phase)} if self.render_rng is not None else {}
        return self.generator_ddp(z, cams, **self.metadata, **extra)

    def _chunks(self, n):
        size = n // self.metadata["batch_split"]
        return [slice(i * size, (i + 1) * size) for i in range(self.metadata["batch_split"])]

    # utils.py:743-842
    def train_discriminator(self, sample):
        md = self.metadata
        real = sample["img"].to(self.device).requires_grad_(True)
        voxels = sample["voxel"].to(self.device)
        n = real.shape[0]
        with torch.no_grad():
            if md.get("random_gen_img", True):
                cams = create_cam2world_matrix(sample_camera_positions(self.device, "y", md["cam_r_start"], md["cam_r_end"], n), "y", self.device)
            else:
                cams = sample["cam2world"].to(self.device)
            with self._phase("d_render"):
                fake = torch.cat([self._render(voxels[c], cams[c], i, "d")[0] for i, c in enumerate(self._chunks(n))], 0)
        t_d = self.timer.begin() if self.timer is not None else None
        r_preds = self._disc(real)
        penalty = 0.0
        if md["r1_lambda"] > 0:
            (grad_real,) = torch.autograd.grad(r_preds.sum(), real, create_graph=True)
            penalty = 0.5 * md["r1_lambda"] * grad_real.reshape(n, -1).norm(2, dim=1).pow(2).mean()
        g_preds = self._disc(fake)
        d_loss = F.softplus(g_preds).mean() + F.softplus(-r_preds).mean() + penalty
        self.optimizer_D.zero_grad()
        d_loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(self.discriminator_ddp.parameters(), md["grad_clip"])
        self.optimizer_D.step()
        if t_d is not None:
            self.timer.end("d_disc_r1_opt", t_d)
        self.losses["d"].append(d_loss.item())
        self.last.update(d_loss=d_loss.item(), r1_penalty=float(penalty.detach()) if torch.is_tensor(penalty) else float(penalty), d_grad_norm=float(gn), fake_mean=float(fake.mean()))

    # utils.py:621-741
    def train_generator(self, sample):
        md = self.metadata
        imgs, cams, voxels = (sample[k].to(self.device) for k in ("img", "cam2world", "voxel"))
        chunks = self._chunks(imgs.shape[0])
        g_acc = p_acc = 0.0
        d_params = [p for p in self.discriminator.parameters() if p.requires_grad]
        for p in d_params:                  # see _disc(frozen=True)
            p.requires_grad_(False)
        try:
            for i, c in enumerate(chunks):
                last = i == len(chunks) - 1
                ctx = contextlib.ExitStack()
                if self.ddp and not last:       # one all-reduce per optimizer step, not one per chunk
                    for m in (self.generator_ddp, self.encoder_ddp):
                        ctx.enter_context(m.no_sync())
                with ctx:
                    with self._phase("g_render_fwd" if i in self._z else "g_encoder_render_fwd"):
                        gen_imgs, _ = self._render(voxels[c], cams[c], i, "g")
                    with self._phase("g_disc_loss_fwd"):
                        if md["enable_discriminator"]:
                            loss_g = F.softplus(-self._disc(gen_imgs, frozen=True)).mean()
                        else:
                            loss_g = gen_imgs.new_zeros(())
                        photo = F.mse_loss(gen_imgs, imgs[c]) if md["photo_loss"] else gen_imgs.new_zeros(())
                    with self._phase("g_backward"):
                        (loss_g + photo).backward()
                g_acc += loss_g.item()
                p_acc += photo.item()
        finally:
            for p in d_params:
                p.requires_grad_(True)
        with self._phase("g_clip_opt"):
            for name, model, opt in (("g", self.generator_ddp, self.optimizer_G), ("e", self.encoder_ddp, self.optimizer_E)):
                self.last[name + "_grad_norm"] = float(torch.nn.utils.clip_grad_norm_(model.parameters(), md.get("grad_clip", 0.3)))
                opt.step()
                opt.zero_grad()
        self.losses["g"].append(g_acc / len(chunks))
        self.losses["photo"].append(p_acc / len(chunks))
        self.last.update(g_loss=g_acc / len(chunks), photo_loss=p_acc / len(chunks))
        # half-precision render backward: (tile, matrix) blocks whose stored gradients left fp16's range and were clamped -- the
        # per-matrix scale comes from a sampled maximum (include/cnerf.h, cnerf_render_backward).  Non-zero: outlier gradients were
        # cut; train with backward_precision "fp32" if that matters.
        from .. import ops
        if getattr(ops, "LAST_SATURATED", None) is not None:
            self.last["render_bwd_clamped_blocks"] = int(ops.LAST_SATURATED.item())
            ops.LAST_SATURATED = None

    def warm_convolutions(self, sample):
        """One throw-away forward + backward of the encoder and the discriminator on the RAW modules (no DDP collective, no
        optimizer step, gradients discarded), with the shapes a training step uses.  With MIOpen find mode on, this is where
        the convolution-kernel search happens (minutes on a fresh machine).  train.py runs it on rank 0 first and on the other
        ranks after a barrier: MIOpen keeps its find results in the user database (MIOPEN_USER_DB_PATH, one directory per
        node), so ranks 1..N-1 read rank 0's results instead of each repeating the search concurrently."""
        md = self.metadata
        n = sample["img"].shape[0]
        c = self._chunks(n)[0]
        vox = sample["voxel"][c].to(self.device)
        fv, glob = self.encoder(vox)
        (fv.square().mean() + glob.square().mean()).backward()
        real = sample["img"].to(self.device).requires_grad_(True)
        out = self.discriminator(real, self.alpha, **md)
        pred = out[0] if isinstance(out, tuple) else out
        (grad_real,) = torch.autograd.grad(pred.sum(), real, create_graph=True)
        (F.softplus(-pred).mean() + grad_real.square().mean()).backward()
        out = self.discriminator(real.detach()[c], self.alpha, **md)          # the G-step chunk size
        (out[0] if isinstance(out, tuple) else out).mean().backward()
        for m in (self.encoder, self.discriminator):
            m.zero_grad(set_to_none=True)

    def step(self, sample):
        """One D step and one G step on `sample`.  Both passes render the same voxel grids and the encoder's parameters do not
        change in between (the D step only updates the discriminator), so with metadata["reuse_encoder_output"] (default on)
        the encoder runs ONCE per accumulation chunk, with its autograd graph, before the D step: the D step's no-grad renders
        read that output, the G step back-propagates through it.  Same numbers as the reference's recomputation
        (utils.py:653-657,778-781), one encoder forward per chunk less (64 ms of a 307 ms step at 128x128x64, batch 8).  Under
        DDP the last of several chunks is left out: its forward must be the one directly followed by the backward that all-reduces."""
        self.set_alpha()
        self._z = {}
        if self.metadata["enable_discriminator"] and self.metadata.get("reuse_encoder_output", True):
            voxels = sample["voxel"].to(self.device)
            chunks = self._chunks(voxels.shape[0])
            for i, c in enumerate(chunks):
                last = i == len(chunks) - 1
                if self.ddp and last and len(chunks) > 1:
                    continue                 # (a single chunk is fine: no other encoder backward sits between its forward and its own)
                with (self.encoder_ddp.no_sync() if self.ddp and not last else contextlib.nullcontext()), self._phase("encoder_fwd"):
                    self._z[i] = self._encode(voxels[c])
        if self.metadata["enable_discriminator"]:
            self.train_discriminator(sample)
        self.train_generator(sample)
        self._z = {}
        self.generator.step += 1
        self.discriminator.step += 1


def synthetic_sample(batch, img_size, voxel_res, device, generator=None):
    """ShapeNetCar-shaped random batch: voxels (B,4,V,V,V) = [occupancy, r, g, b] (datasets.py:105-110), images in [-1,1],
    cameras on the training shell."""
    g = generator
    occ = (torch.rand(batch, 1, voxel_res, voxel_res, voxel_res, generator=g) > 0.9).float()
    rgb = torch.rand(batch, 3, voxel_res, voxel_res, voxel_res, generator=g) * occ
    cams = create_cam2world_matrix(sample_camera_positions("cpu", "y", 0.7, 1.5, batch), "y")
    return {"voxel": torch.cat([occ, rgb], 1).to(device), "img": (torch.rand(batch, 3, img_size, img_size, generator=g) * 2 - 1).to(device),
            "cam2world": cams.to(device)}
