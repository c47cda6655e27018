"""GroupNorm apply (+ SiLU) alone, per shape of the step, graph-replayed over a pool of buffers larger than the caches: is the 2 TB/s of the
small shapes in the per-shape table the kernel or the company it keeps?   python tools/ab_gn_apply.py [label]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import _C, ops  # noqa: E402
from e4t.ops import _ptr, _stream  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda:0")
hip = ops.HipBackend()
lib = hip.lib
bf16, f32 = torch.bfloat16, torch.float32


def graph_time(fns, iters):
    for f in fns:
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(iters):
            fns[i % len(fns)]()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return min(ts)


for B, HW, C in [(16, 4096, 320), (16, 1024, 640), (16, 4096, 512), (16, 4096, 640), (16, 16384, 512), (16, 65536, 256), (16, 262144, 128), (16, 256, 1280)]:
    G = 32
    nbytes = B * HW * C * 2
    nset = max(2, min(12, int(600e6 // (2 * nbytes)) + 1))
    xs = [(torch.randn(B * HW, C, device=dev) * 0.5).to(bf16) for _ in range(nset)]
    ys = [torch.empty_like(x) for x in xs]
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    stats = torch.empty((B, G, 2), dtype=f32, device=dev)
    ws = hip.workspace(lib.e4t_groupnorm_workspace_bytes(B, HW, C, G, 0), dev)
    _C.check(lib.e4t_groupnorm_stats(_ptr(xs[0]), C, None, 0, B, HW, G, 1e-5, _ptr(stats), _ptr(ws), ws.numel(), _stream()), "stats")
    fns = [(lambda x=x, y=y: _C.check(lib.e4t_groupnorm_apply(_ptr(x), C, None, 0, _ptr(stats), _ptr(g), _ptr(b), _ptr(y), B, HW, G, 1, _stream()), "apply")) for x, y in zip(xs, ys)]
    t = graph_time(fns, 4 * nset)
    print(f"[{label}] gn_apply B{B} HW{HW} C{C}: {t:7.1f} us  {2 * nbytes / t / 1e3:6.0f} GB/s (read + write)", flush=True)
    del xs, ys
