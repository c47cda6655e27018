"""GPU: GroupNorm forward (column-statistics path) / backward on the step's shapes with cold operands (pool > Infinity Cache)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
hip = ops.HipBackend()
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf16)


def with_cs(x):
    blk = x.float().view(x.shape[0] // 32, 32, x.shape[1])
    x._e4t_colstats = torch.stack([blk.sum(1), (blk * blk).sum(1)], dim=-1).contiguous()
    return x


def pool_time(bufs, run, iters=24):
    for i in range(3):
        run(bufs[i % len(bufs)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(bufs[i % len(bufs)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


tot_f = tot_b = 0.0
# (B, HW, C1, C2, launches per step fwd, bwd)  — UNet ResBlock / transformer GroupNorms of the SD-1.4 B=16 step
shapes = [(16, 4096, 320, 0, 19, 13), (16, 4096, 320, 320, 6, 4), (16, 4096, 640, 320, 2, 1), (16, 1024, 640, 0, 18, 12), (16, 1024, 640, 640, 3, 2),
          (16, 1024, 1280, 640, 2, 1), (16, 1024, 320, 0, 2, 2), (16, 256, 1280, 0, 18, 12), (16, 256, 1280, 1280, 3, 2), (16, 256, 640, 0, 2, 2),
          (16, 64, 1280, 0, 19, 12), (16, 64, 1280, 1280, 6, 4), (16, 262144, 128, 0, 4, 0), (16, 65536, 256, 0, 3, 0), (16, 16384, 512, 0, 3, 0)]
for B, HW, C1, C2, nf, nb in shapes:
    C = C1 + C2
    n = max(2, min(16, int(700e6 / (B * HW * C * 2)) + 1))
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    xs = [(with_cs(r(B * HW, C1)), with_cs(r(B * HW, C2)) if C2 else None) for _ in range(n)]
    tf = pool_time(xs, lambda b: hip.groupnorm_fwd(b[0], b[1], ga, be, B, HW, 32, 1e-5, True))
    y, st = hip.groupnorm_fwd(xs[0][0], xs[0][1], ga, be, B, HW, 32, 1e-5, True)
    dys = [r(B * HW, C) for _ in range(n)]
    adds = [r(B * HW, C1) for _ in range(n)]
    idx = [0]
    def bwd(b):
        i = idx[0] % n; idx[0] += 1
        hip.groupnorm_bwd(b[0], b[1], dys[i], st, ga, be, adds[i], B, HW, 32, True)
    tb = pool_time(xs, bwd) if nb else 0.0
    fb, bb = 4.0 * B * HW * C, 2.0 * B * HW * (5 * C + C1)
    tot_f += tf * nf; tot_b += tb * nb
    print(f"B{B} HW{HW} C{C1}+{C2}: fwd {tf*1e6:7.1f} us {fb/tf/1e9:6.0f} GB/s | bwd {tb*1e6:7.1f} us {bb/tb/1e9 if tb else 0:6.0f} GB/s")
print(f"weighted per step: fwd {tot_f*1e3:.2f} ms  bwd {tot_b*1e3:.2f} ms   (E4T_GN_BLOCKS={os.environ.get('E4T_GN_BLOCKS', '1024')})")
