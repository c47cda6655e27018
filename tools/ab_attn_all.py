"""Forward and backward attention over every attention shape of the B = 16 SD-1.4 step (and the CLIP-ViT's), graph-replayed, for timing
compile-time variants (E4T_LIB=<variant .so>).   python tools/ab_attn_all.py [label]"""
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


def graph_time(fn, iters=10):
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


# (B, H, T, S, dh, launches per step fwd, bwd)
SHAPES = [(16, 8, 4096, 4096, 40, 6, 6), (16, 8, 1024, 1024, 80, 7, 7), (16, 8, 256, 256, 160, 7, 7), (16, 8, 64, 64, 160, 2, 2),
          (16, 8, 4096, 77, 40, 7, 7), (16, 8, 1024, 77, 80, 7, 7), (16, 8, 256, 77, 160, 7, 7), (16, 16, 257, 257, 80, 32, 0), (16, 12, 77, 77, 64, 12, 12)]
tot = 0.0
for B, H, T, S, DH, nf, nb in SHAPES:
    d = H * DH
    causal = (T == 77)
    q, k, v = r(B * T, d), r(B * S, d), r(B * S, d)
    o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5, causal=causal)
    out = torch.empty_like(o)
    tf = graph_time(lambda: hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5, causal=causal, out=out))
    tb = 0.0
    if nb:
        dq, dk, dv, do = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), r(B * T, d)
        tb = graph_time(lambda: hip.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, B, H, T, S, DH, DH ** -0.5, causal=causal))
    tot += nf * tf + nb * tb
    print(f"[{label}] B{B} H{H} T{T} S{S} dh{DH}{' causal' if causal else ''}: fwd {tf:8.1f} us  bwd {tb:8.1f} us   per step {(nf * tf + nb * tb) / 1e3:6.2f} ms", flush=True)
print(f"[{label}] attention per step (launch-weighted): {tot / 1e3:.2f} ms", flush=True)
