"""HBM-bound kernels micro-benchmark for compile-time variants (E4T_LIB=<variant .so>): AdamW over the step's 374.6 M parameters, GEGLU
forward / backward at the 64 x 64 level, graph-replayed; GB/s = algorithmic bytes.   python tools/ab_stream.py [label]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import ops  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("E4T_LIB", "default"))
dev = torch.device("cuda:0")
hip = ops.HipBackend()


def graph_time(fn, iters=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


n = 374580608
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
t = graph_time(lambda: hip.adamw(p, g, m, v, 1e-4, 0.9, 0.999, 1e-8, 1e-2, 3), iters=5)
print(f"[{label}] adamw n={n}: {t:8.1f} us  {28.0 * n / t / 1e3:7.0f} GB/s", flush=True)
del p, g, m, v
for M, H in [(65536, 1280), (16384, 2560)]:
    u = (torch.randn(M, 2 * H, device=dev) * 0.5).to(torch.bfloat16)
    dh = (torch.randn(M, H, device=dev) * 0.5).to(torch.bfloat16)
    tf = graph_time(lambda: hip.geglu_fwd(u))
    tb = graph_time(lambda: hip.geglu_bwd(u, dh))
    print(f"[{label}] geglu M{M} H{H}: fwd {tf:7.1f} us {6.0 * M * H / tf / 1e3:6.0f} GB/s | bwd {tb:7.1f} us {10.0 * M * H / tb / 1e3:6.0f} GB/s", flush=True)
