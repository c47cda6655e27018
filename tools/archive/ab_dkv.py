"""GPU: attention backward with the dK/dV kernel bounded for 2 (register prefetch) vs 3 (no prefetch) workgroups per CU.
The switch is read once per process: run as  E4T_ATTN_DKV_OCC=2 python tools/ab_dkv.py ; E4T_ATTN_DKV_OCC=3 python tools/ab_dkv.py
Prints ms per backward (delta + dK/dV + dQ kernels) and a checksum of dK, dV (must agree between the two runs)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd")]
import torch
from e4t import ops
hip = ops.HipBackend()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
r = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)
print("E4T_ATTN_DKV_OCC =", os.environ.get("E4T_ATTN_DKV_OCC"))
for (B, H, T, S, DH, causal) in [(16, 8, 4096, 4096, 40, False), (16, 8, 4096, 77, 40, False), (16, 12, 77, 77, 64, True), (16, 8, 1024, 1024, 80, False)]:
    d = H * DH
    q, k, v, do = r(B * T, d), r(B * S, d), r(B * S, d), r(B * T, d)
    o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5, causal=causal)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    run = lambda: hip.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, B, H, T, S, DH, DH ** -0.5, causal=causal)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"B{B} H{H} T{T} S{S} dh{DH} causal{int(causal)}: {e0.elapsed_time(e1) / 20:.4f} ms/bwd   checksum dk {dk.float().abs().sum().item():.6e} dv {dv.float().abs().sum().item():.6e}")
