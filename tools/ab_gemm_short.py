"""Short-K / mid-size GEMM micro-benchmark for library variants (E4T_LIB): the step's projection shapes, graph-replayed, operands cycled
through a pool larger than the caches.   python tools/ab_gemm_short.py [label] [tile]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import ops  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
hip = ops.HipBackend()
bf16 = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(bf16)


def graph_time(fns, iters):
    for f in fns:
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(iters):
            fns[i % len(fns)]()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M, N, K in [(65536, 320, 320), (65536, 960, 320), (65536, 1280, 320), (65536, 2560, 320), (16384, 640, 640), (4096, 1280, 1280), (16384, 1920, 640), (65536, 320, 1280)]:
    nset = max(3, min(8, int(500e6 // (M * (K + N) * 2))))
    As = [r(M, K) for _ in range(nset)]
    b = r(N, K) * K ** -0.5
    outs = [torch.empty((M, N), dtype=bf16, device=dev) for _ in range(nset)]
    fns = [(lambda a=a, o=o: hip.gemm(a, b, out=o, tile=tile)) for a, o in zip(As, outs)]
    t = graph_time(fns, 4 * nset)
    by = 2.0 * (M * K + N * K + M * N)
    print(f"[{label}] gemm M{M} N{N} K{K} tile{tile}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF {by / t / 1e3:6.0f} GB/s", flush=True)
    del As, outs
