"""GPU: split-K of the TN (weight-gradient) GEMM on the step's shapes (B = 16, and B = 1 / 4 of BASELINE configs[4]): launches graph-replayed,
operands cold (pool larger than the Infinity Cache).  dW[M, N] = dY[K, M]^T . X[K, N], fp32 out."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
hip = ops.HipBackend()
dev = torch.device("cuda:0")
bf16, f32 = torch.bfloat16, torch.float32
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf16)


def pool_time(make, run, nbytes, iters=16):
    n = max(3, min(32, int(700e6 / max(nbytes, 1)) + 1))
    bufs = [make() for _ in range(n)]
    for i in range(3):
        run(bufs[i % n])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            run(bufs[(i + 3) % n])
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e-3


shapes = [(3840, 1280, 4096), (960, 320, 65536), (1920, 640, 16384), (320, 320, 65536), (2560, 768, 1232), (640, 768, 1232), (1280, 768, 1232), (640, 640, 16384),
          (1280, 1280, 4096), (3840, 1280, 1024), (1280, 1280, 1024), (1280, 1280, 2064), (1280, 1280, 16),
          # BASELINE configs[4] at B = 1 (768 px: 9216 / 2304 / 576 / 144 positions, 77 text tokens, 257 ViT tokens) and B = 4
          (320, 320, 9216), (960, 320, 9216), (640, 640, 2304), (1920, 640, 2304), (1280, 1280, 576), (3840, 1280, 576), (1280, 1280, 144), (3840, 1280, 144),
          (640, 1024, 77), (1280, 1024, 77), (2560, 1024, 77), (1280, 1280, 257), (320, 320, 36864), (640, 640, 9216), (1280, 1280, 2304), (1280, 1024, 308)]
DUMP = []
for M, N, K in shapes:
    nkt = (K + 63) // 64
    mk = lambda: (r(K, M), r(K, N), torch.empty((M, N), dtype=f32, device=dev))
    nb = 2.0 * K * (M + N) + 4.0 * M * N
    res = []
    for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48):
        if sk > 1 and (sk > nkt or M * N * sk > 64e6):
            continue
        try:
            t = pool_time(mk, lambda b: hip.gemm_tn(b[0], b[1], out=b[2], splitk=sk), nb)
        except Exception as e:
            continue
        res.append((t, sk))
    auto = pool_time(mk, lambda b: hip.gemm_tn(b[0], b[1], out=b[2]), nb)
    DUMP.append(dict(M=M, N=N, K=K, auto=auto * 1e6, variants=[(sk, t * 1e6) for t, sk in res]))
    print(f"gemm_tn M{M} N{N} K{K} (tiles {((M+127)//128)*((N+127)//128)}, K-tiles {nkt}): auto {auto*1e6:6.1f}us | " + " ".join(f"k{sk}:{t*1e6:5.1f}" for t, sk in res)
          + f" | best k{min(res)[1]} ({auto/min(res)[0]:.2f}x)", flush=True)
json.dump(DUMP, open(os.path.join(R, "gpurun_out", "sweep_tn.json"), "w"))
