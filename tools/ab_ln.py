"""LayerNorm forward / backward micro-benchmark at the step's shapes, graph-replayed over rotating buffers (E4T_LIB=<variant .so>)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import ops  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("E4T_LIB", "default"))
dev = torch.device("cuda:0")
hip = ops.HipBackend()


def graph_time(fns, reps=4):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns:
                f()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / (reps * len(fns)) * 1e3)
    return min(ts)


for M, D, dt in [(4112, 1280, torch.float32), (4096, 1280, torch.bfloat16), (16384, 640, torch.bfloat16), (65536, 320, torch.bfloat16), (1232, 768, torch.bfloat16)]:
    xs = [(torch.randn(M, D, device=dev)).to(dt) for _ in range(6)]
    ga, be = torch.randn(D, device=dev), torch.randn(D, device=dev)
    t = graph_time([lambda x=x: hip.layernorm_fwd(x, ga, be, 1e-5) for x in xs])
    nb = (xs[0].element_size() + 2.0) * M * D
    ys = [hip.layernorm_fwd(x, ga, be, 1e-5) for x in xs]
    line = f"[{label}] ln_fwd M{M} D{D} {str(dt)[6:]}: {t:6.1f} us {nb / t / 1e3:6.0f} GB/s"
    if dt == torch.bfloat16:
        dys = [torch.randn(M, D, device=dev).to(dt) for _ in range(6)]
        tb = graph_time([lambda x=x, dy=dy, st=y[1]: hip.layernorm_bwd(x, dy, ga, st) for x, dy, y in zip(xs, dys, ys)])
        line += f" | ln_bwd {tb:6.1f} us {6.0 * M * D / tb / 1e3:6.0f} GB/s"
    print(line, flush=True)
