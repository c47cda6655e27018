"""ATen-level view of one TUNING step (all UNet weights trainable): which non-native ops remain and where they come from."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), R]
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from e4t.trainer import E4TTrainer
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
unet, enc, text, vae = bench.build_models(dev, "sd14", 0)
tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, class_token_id=1125, device=dev, tuning=True, max_grad_norm=1.0)
B = 16
g = torch.Generator(device=dev); g.manual_seed(0)
px = (torch.rand((1, 3, 512, 512), generator=g, device=dev) * 2 - 1).expand(B, -1, -1, -1).contiguous()
lat = tr.encode_latents(px, torch.randn((B, 4, 64, 64), generator=g, device=dev))
ids = torch.randint(0, 49000, (1, 77), generator=g, device=dev).expand(B, -1).contiguous()
pidx = torch.full((B,), 4, device=dev)
for _ in range(3): tr.train_step(px, ids, pidx, latents=lat)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(4): tr.train_step(px, ids, pidx, latents=lat)
torch.cuda.synchronize(); print("tuning step ms", (time.perf_counter() - t0) / 4 * 1e3)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(px, ids, pidx, latents=lat); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70))
ev = [e for e in prof.events() if e.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::sum", "aten::add_", "aten::to", "aten::_to_copy")]
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    st = [s for s in (e.stack or []) if "e4t" in s or "trainer" in s][:3]
    k = (e.name, " <- ".join(s.split("/")[-1] for s in st))
    agg[k][0] += 1; agg[k][1] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{v[1]/1e3:8.2f} ms {v[0]:5d}x {k[0]:18s} {k[1]}")
