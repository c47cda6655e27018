import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd")]
import torch
sys.path.insert(0, os.path.join(R, 'tests'))
from torch_twins import CLIPTextModel as TorchText
from e4t.checkpoint_trees import CLIP_TEXT_L
from e4t.text import CLIPTextModel as NativeText
dev = torch.device("cuda:0")
torch.manual_seed(0)
ref = TorchText(**CLIP_TEXT_L).requires_grad_(False).to(dev).to(torch.bfloat16)
nat = NativeText(**CLIP_TEXT_L).requires_grad_(False).to(dev)
B = 16
def run(m, dt):
    e = (torch.randn(B, 77, 768, device=dev, dtype=dt) * 0.3).requires_grad_(True)
    y = m(inputs_embeds=e)[0]
    y.float().sum().backward()
for name, m, dt in (("torch bf16", ref, torch.bfloat16), ("native", nat, torch.float32)):
    for _ in range(3): run(m, dt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run(m, dt)
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms fwd+bwd (B=16, 77 tokens, CLIP-L)")
