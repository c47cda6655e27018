"""Debug: backward attention outputs of the current library (E4T_LIB) on a fixed seeded problem, saved for a cross-library comparison."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
out, scale_in = sys.argv[1], float(sys.argv[2])
B, H, T, S, DH = (int(x) for x in sys.argv[3:8]) if len(sys.argv) > 7 else (16, 8, 4096, 4096, 40)
dev = torch.device("cuda:0")
hip = ops.HipBackend()
g = torch.Generator(device=dev).manual_seed(5)
d = H * DH
qkv = (torch.randn(B * T, 3 * d, device=dev, generator=g) * scale_in).to(torch.bfloat16)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
do = (torch.randn(B * T, d, device=dev, generator=g) * 0.5).to(torch.bfloat16)
o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5)
gr = torch.zeros_like(qkv)
hip.attention_bwd(q, k, v, o, do, lse, gr[:, :d], gr[:, d:2 * d], gr[:, 2 * d:], B, H, T, S, DH, DH ** -0.5)
torch.cuda.synchronize()
torch.save(dict(o=o.cpu(), lse=lse.cpu(), g=gr.cpu()), out)
print("saved", out, float(gr.float().abs().mean()), float(lse.abs().max()))
