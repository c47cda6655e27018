"""cProfile of the HOST side of one training step (the step is launch-bound at small batch)."""
import os, sys, cProfile, pstats, io
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), R]
import torch
import bench
from e4t.trainer import E4TTrainer
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
unet, enc, text, vae = bench.build_models(dev, "sd14", 0)
tr = E4TTrainer(unet, enc, text, vae, lr=1e-6, class_token_id=1125, device=dev)
B = 2
g = torch.Generator(device=dev); g.manual_seed(0)
bt = (torch.rand((B, 3, 512, 512), generator=g, device=dev) * 2 - 1, torch.randint(0, 49000, (B, 77), generator=g, device=dev),
      torch.randint(1, 20, (B,), generator=g, device=dev))
for _ in range(3): tr.train_step(*bt)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.train_step(*bt)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
