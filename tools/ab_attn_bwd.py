"""Backward attention of the step's dominant shape only (B16 H8 T = S = 4096 dh 40), graph-replayed, for timing variants."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import ops  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("E4T_LIB", "default"))
dev = torch.device("cuda:0")
hip = ops.HipBackend()
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def graph_time(fn, iters=6):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return min(ts)


B, H, T, S, DH = 16, 8, 4096, 4096, 40
d = H * DH
qkv = r(B * T, 3 * d); q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
g = torch.empty_like(qkv); dq, dk, dv = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5)
do = r(B * T, d)
tb = graph_time(lambda: hip.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, B, H, T, S, DH, DH ** -0.5))
fl = 10.0 * B * H * T * S * DH
print(f"[{label}] B{B} H{H} T{T} S{S} dh{DH}: bwd (dQ + dK/dV) {tb:8.1f} us {fl / tb / 1e6:6.1f} TF", flush=True)
