import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import cnerf_amd
from cnerf_amd.training import GanTrainer, default_metadata
from cnerf_amd.training.gan_step import synthetic_sample
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(0)
md = default_metadata(128, 64, 8, 4, "SHORTSIREN_FG", 256)
md["render_precision"] = "fp16x3"
md["encoder_channels_last"] = bool(int(os.environ.get("CNERF_ENCODER_CHANNELS_LAST", "0")))
tr = GanTrainer(md, dev)
gen = torch.Generator().manual_seed(1)
sample = synthetic_sample(8, 128, 64, dev, gen)
for _ in range(2): tr.step(sample)
torch.cuda.synchronize()
def timed(f, n=2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
vox = sample["voxel"].to(dev)[:2]
print("encoder fwd (2 images, no grad)  %.1f ms" % timed(lambda: tr.encoder(vox)) if False else "", end="")
with torch.no_grad():
    print("encoder fwd no_grad x2 images: %.1f ms" % timed(lambda: tr.encoder(vox)))
def enc_fb():
    fv, g = tr.encoder(vox); (fv.square().mean() + g.square().mean()).backward()
print("encoder fwd+bwd x2 images: %.1f ms" % timed(enc_fb))
img = sample["img"].to(dev)
def d_step():
    real = img.clone().requires_grad_(True)
    r = tr.discriminator(real, 1.0)
    (gr,) = torch.autograd.grad(r.sum(), real, create_graph=True)
    (torch.nn.functional.softplus(-r).mean() + 0.5 * gr.reshape(8, -1).norm(2, dim=1).pow(2).mean()).backward()
print("discriminator real + R1 fwd/bwd x8: %.1f ms" % timed(d_step))
print("D step total: %.1f ms" % timed(lambda: tr.train_discriminator(sample)))
print("G step total: %.1f ms" % timed(lambda: tr.train_generator(sample)))
