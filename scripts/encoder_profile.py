import sys; sys.path.insert(0, "/root/repo")
import torch, cnerf_amd
from cnerf_amd.training import UNet3D
import os
torch.backends.cudnn.benchmark = bool(int(os.environ.get("BENCHMARK", "0")))
dev = torch.device("cuda:0"); torch.manual_seed(0)
net = UNet3D(in_channels=4, out_channels=32, f_maps=32, num_levels=4, return_global=True).to(dev)
vox = torch.rand(2, 4, 64, 64, 64, device=dev)
def fb():
    fv, g = net(vox); (fv.square().mean() + g.square().mean()).backward()
for _ in range(2): fb()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    fb(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
