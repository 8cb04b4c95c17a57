"""GPU smoke of the GAN step harness: D step with R1, G step with split accumulation through the HIP
render path and its backward, plain and under DDP with a one-rank RCCL process group."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(ddp):
    import cnerf_amd
    from cnerf_amd.training import GanTrainer, default_metadata
    from cnerf_amd.training.gan_step import synthetic_sample
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    md = default_metadata(img_size=32, num_steps=12, batch_size=4, batch_split=2, hidden_dim=64)
    md["unet"].update(f_maps=8, num_levels=3)
    md["generator"]["z_dim"] = 32                      # deepest encoder level: 8 * 2**2 channels
    tr = GanTrainer(md, dev, ddp=ddp)
    before = {k: v.detach().clone() for k, v in tr.generator.state_dict().items()}
    enc_before = tr.encoder.final_conv.weight.detach().clone()
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        tr.step(synthetic_sample(4, 32, 16, dev, g))
    assert all(map(lambda x: x == x and abs(x) < 1e4, tr.losses["d"] + tr.losses["g"] + tr.losses["photo"]))
    changed = [k for k, v in tr.generator.state_dict().items() if not torch.equal(v, before[k])]
    assert len(changed) == len(before), "every generator parameter receives a gradient through the HIP backward"
    assert not torch.equal(tr.encoder.final_conv.weight, enc_before), "the encoder is trained through d(feature volume)"
    assert tr.generator.step == 2 and 0 < tr.alpha <= 1 and md["nerf_noise"] < 1.0


def test_gan_step_single_process():
    _run(ddp=False)


def test_gan_step_ddp_one_rank_rccl():
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        _run(ddp=True)
    finally:
        dist.destroy_process_group()
