"""GEMM / 3x3-conv micro-benchmark for compile-time variants of the ping-pong kernels (E4T_LIB=<variant .so>): the step's heaviest
shapes of the 256 x 256 and 256 x 320 tiles, graph-replayed with operands cycled through a pool larger than the caches, plus a bitwise
checksum of every output (variants that only re-schedule instructions must reproduce it).   python tools/ab_conv.py [label]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import ops  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("E4T_LIB", "default"))
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


def csum(t):
    return int(t.view(torch.int16).to(torch.int64).sum().item()) & 0xFFFFFFFF


convs = [(16, 128, 512, 512, 0), (16, 256, 256, 256, 0), (16, 64, 512, 512, 0), (16, 16, 1280, 1280, 0), (16, 64, 320, 320, 0), (16, 64, 640, 640, 0), (16, 32, 1280, 1280, 0),
         (16, 32, 640, 640, 0)]
for B, H, Cin, Cout, tile in convs:
    nset = max(2, min(6, int(600e6 // (B * H * H * Cin * 2))))
    xs = [r(B * H * H, Cin) for _ in range(nset)]
    w = r(Cout, 9 * Cin) * (9 * Cin) ** -0.5
    bias = torch.randn(Cout, device=dev, generator=g)
    outs = [torch.empty((B * H * H, Cout), dtype=bf16, device=dev) for _ in range(nset)]
    fns = [(lambda x=x, o=o: hip.conv3x3(x, w, B, H, H, H, H, 1, bias=bias, out=o, tile=tile)) for x, o in zip(xs, outs)]
    t = graph_time(fns, 3 * nset)
    pl = hip._plan(hip.lib.e4t_conv3x3_plan, __import__("e4t._C", fromlist=["ConvDesc"]).ConvDesc(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, Cout=Cout, mode=1), "plan")
    print(f"[{label}] conv B{B} {H}x{H} {Cin}->{Cout} tile{pl.tile} sk{pl.splitk}: {t:8.1f} us {2.0 * B * H * H * Cout * 9 * Cin / t / 1e6:7.1f} TF  csum {csum(outs[0]):08x}", flush=True)
    del xs, outs
gemms = [(4112, 3840, 1280, False), (65536, 320, 2560, False), (4096, 10240, 1280, False), (65536, 320, 320, False), (4112, 5120, 1280, True), (16384, 5120, 640, False),
         (65536, 320, 1280, False)]
for M, N, K, gelu in gemms:
    nset = max(2, min(6, int(400e6 // (M * K * 2))))
    As = [r(M, K) for _ in range(nset)]
    b = r(N, K) * K ** -0.5
    outs = [torch.empty((M, N), dtype=bf16, device=dev) for _ in range(nset)]
    fns = [(lambda a=a, o=o: hip.gemm(a, b, out=o, gelu=gelu)) for a, o in zip(As, outs)]
    t = graph_time(fns, 3 * nset)
    print(f"[{label}] gemm M{M} N{N} K{K}{' gelu' if gelu else ''}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF  csum {csum(outs[0]):08x}", flush=True)
    del As, outs
