"""The frozen CLIP-ViT's four GEMMs per layer at the headline batch (M = 16 x 257 = 4112 token rows) next to the same N, K at M = 4096, isolated:
graph-replayed, operands cycled through a pool larger than the caches, epilogues as the encoder calls them (fp32 residual stream for out-proj / fc2,
GELU for fc1).   python tools/ab_vit_gemm.py [label]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import ops  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda:0")
hip = ops.HipBackend()
bf16, f32 = torch.bfloat16, torch.float32
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


CASES = [("qkv", 3840, 1280, "bf16"), ("out-proj", 1280, 1280, "res"), ("fc1", 5120, 1280, "gelu"), ("fc2", 1280, 5120, "res")]
for name, N, K, kind in CASES:
    for M in (4096, 4112, 4128):
        nset = 6
        As = [r(M, K) for _ in range(nset)]
        b = r(N, K) * K ** -0.5
        bias = torch.randn(N, device=dev, generator=g)
        if kind == "res":
            res = [torch.randn(M, N, device=dev, generator=g) for _ in range(nset)]
            outs = [torch.empty((M, N), dtype=f32, device=dev) for _ in range(nset)]
            fns = [(lambda a=a, o=o, x=x: hip.gemm(a, b, bias=bias, residual=x, out=o)) for a, o, x in zip(As, outs, res)]
        else:
            outs = [torch.empty((M, N), dtype=bf16, device=dev) for _ in range(nset)]
            fns = [(lambda a=a, o=o: hip.gemm(a, b, bias=bias, out=o, gelu=(kind == "gelu"))) for a, o in zip(As, outs)]
        t = graph_time(fns, 4 * nset)
        print(f"[{label}] {name:9s} M{M} N{N} K{K}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF", flush=True)
        del As, outs
