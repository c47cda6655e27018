import os, sys
R = "/root/repo"
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
dev = torch.device("cuda:0"); hip = ops.HipBackend(); bf16 = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf16)
def graph_time(fns, iters):
    for f in fns: f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(iters): fns[i % len(fns)]()
    gr.replay(); torch.cuda.synchronize()
    ts=[]
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return min(ts)
for B, H, Cin, Cout in [(16, 256, 128, 256), (16, 256, 256, 256), (16, 512, 128, 128), (16, 128, 512, 512), (16, 128, 256, 512), (16, 64, 512, 512), (16, 64, 320, 320), (16, 64, 640, 640), (16, 32, 640, 640), (16, 32, 1280, 1280), (16, 16, 1280, 1280)]:
  for tile in ((0, 512) if os.environ.get('E4T_CONV_STRIP256') else (0, 5256)):
    xs = [r(B * H * H, Cin) for _ in range(2)]
    w = r(Cout, 9 * Cin) * (9 * Cin) ** -0.5
    outs = [torch.empty((B * H * H, Cout), dtype=bf16, device=dev) for _ in range(2)]
    fns = [(lambda x=x, o=o: hip.conv3x3(x, w, B, H, H, H, H, 1, out=o, tile=tile)) for x, o in zip(xs, outs)]
    t = graph_time(fns, 6)
    print(f"conv B{B} {H}x{H} {Cin}->{Cout} tile{tile}: {t:8.1f} us {2.0 * B * H * H * Cout * 9 * Cin / t / 1e6:7.1f} TF", flush=True)
