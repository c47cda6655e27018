"""Which ATen ops (i.e. launches that are NOT ours) does one training step still issue?  torch.profiler, grouped by op + stack."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), R]
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from e4t.trainer import E4TTrainer
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
unet, enc, text, vae = bench.build_models(dev, "sd14", 0)
tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, class_token_id=1125, device=dev)
B = 16
g = torch.Generator(device=dev); g.manual_seed(0)
bt = (torch.rand((B, 3, 512, 512), generator=g, device=dev) * 2 - 1, torch.randint(0, 49000, (B, 77), generator=g, device=dev),
      torch.randint(1, 20, (B,), generator=g, device=dev))
for _ in range(2): tr.train_step(*bt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(*bt); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=50, max_src_column_width=110))
