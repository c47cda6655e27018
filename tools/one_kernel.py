"""Run one kernel shape a few times (for rocprofv3 --pmc):  python tools/one_kernel.py conv|gemm|attnbwd [tile]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd")]
import torch
from e4t import ops
hip = ops.HipBackend(); dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
what = sys.argv[1]; tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if what == "conv":
    B, H, Cin, Cout = 16, 64, 320, 320
    x, w = r(B * H * H, Cin), r(Cout, 9 * Cin)
    for _ in range(4): hip.conv3x3(x, w, B, H, H, H, H, 1, tile=tile, splitk=1)
elif what == "gemm":
    a, b = r(8192, 8192), r(8192, 8192)
    for _ in range(4): hip.gemm(a, b, tile=tile, splitk=1)
elif what == "attnbwd":
    B, H, T, DH = 16, 8, 4096, 40; d = H * DH
    qkv = r(B * T, 3 * d); q, k, v = qkv[:, :d], qkv[:, d:2*d], qkv[:, 2*d:]
    g = torch.empty_like(qkv); o, lse = hip.attention_fwd(q, k, v, B, H, T, T, DH, DH ** -0.5); do = r(B * T, d)
    for _ in range(3): hip.attention_bwd(q, k, v, o, do, lse, g[:, :d], g[:, d:2*d], g[:, 2*d:], B, H, T, T, DH, DH ** -0.5)
torch.cuda.synchronize()
