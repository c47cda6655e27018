"""GPU: 2 vs 3 vs 4 LDS stages of the 64 / 128 / 160 DMA GEMM tiles on the step's shapes, cold operands (a pool of buffers larger
than the 256 MB Infinity Cache is cycled so that every launch reads from HBM like in the training step)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
hip = ops.HipBackend()
dev = torch.device("cuda:0")
bf16 = torch.bfloat16


def pool_time(make, run, nbytes, iters=20):
    n = max(2, min(24, int(600e6 / max(nbytes, 1)) + 1))
    bufs = [make() for _ in range(n)]
    for i in range(3):
        run(bufs[i % n])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(bufs[i % n])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf16)
# correctness of the deeper pipelines first: same accumulation order -> bit-identical to the 2-stage kernel
for M, N, K in [(4096, 1280, 1280), (1000, 640, 328), (4112, 1280, 64), (300, 320, 192)]:
    a, w = r(M, K), r(N, K)
    for tile in (160, 128, 64):
        if tile == 160 and N % 160:
            continue
        base = hip.gemm(a, w, tile=tile)
        for st in (3, 4):
            got = hip.gemm(a, w, tile=st * 1000 + tile)
            assert torch.equal(base, got), (M, N, K, tile, st, float((base.float() - got.float()).abs().max()))
x, w = r(2 * 16 * 16, 128), r(256, 9 * 128)
for tile in (128, 64):
    base = hip.conv3x3(x, w, 2, 16, 16, 16, 16, 1, tile=tile)
    for st in (3, 4):
        assert torch.equal(base, hip.conv3x3(x, w, 2, 16, 16, 16, 16, 1, tile=st * 1000 + tile)), (tile, st)
print("deeper pipelines: bit-identical to the 2-stage kernels")
gemms = [(65536, 320, 320), (16384, 640, 640), (4096, 1280, 1280), (4096, 1280, 5120), (4096, 5120, 1280), (4096, 10240, 1280), (4096, 1280, 10240),
         (65536, 2560, 320), (65536, 320, 1280), (16384, 5120, 640), (16384, 640, 2560), (4112, 5120, 1280), (4112, 1280, 5120), (4112, 3840, 1280),
         (4112, 1280, 1280), (1024, 1280, 1280), (1232, 768, 3072), (1232, 3072, 768), (1024, 10240, 1280), (1024, 1280, 5120)]
for M, N, K in gemms:
    w = r(N, K)
    res = []
    for tile in (160, 128, 64):
        if tile == 160 and N % 160:
            continue
        for st in (2, 3, 4):
            code = tile if st == 2 else st * 1000 + tile
            t = pool_time(lambda: (r(M, K), torch.empty((M, N), dtype=bf16, device=dev)), lambda b: hip.gemm(b[0], w, out=b[1], tile=code), 2.0 * M * (K + N))
            res.append((t, tile, st))
    auto = pool_time(lambda: (r(M, K), torch.empty((M, N), dtype=bf16, device=dev)), lambda b: hip.gemm(b[0], w, out=b[1]), 2.0 * M * (K + N))
    best = min(res)
    print(f"gemm M{M} N{N} K{K}: auto {auto*1e6:7.1f}us | " + " ".join(f"t{tl}s{st}:{t*1e6:6.1f}" for t, tl, st in res) + f" | best t{best[1]}s{best[2]} {2.0*M*N*K/best[0]/1e12:.0f}TF ({auto/best[0]:.2f}x)")
convs = [(16, 64, 320, 320), (16, 32, 640, 640), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 8, 1280, 1280), (16, 8, 2560, 1280), (16, 64, 640, 320), (16, 32, 1280, 640)]
for B, H, Cin, Cout in convs:
    w = r(Cout, 9 * Cin)
    res = []
    for tile in (160, 128):
        if tile == 160 and Cout % 160:
            continue
        for st in (2, 3, 4):
            for sk in (1, 2, 3, 4, 6):
                if sk > 1 and H > 16:
                    continue
                code = tile if st == 2 else st * 1000 + tile
                try:
                    t = pool_time(lambda: r(B * H * H, Cin), lambda x: hip.conv3x3(x, w, B, H, H, H, H, 1, tile=code, splitk=sk), 2.0 * B * H * H * (Cin + Cout), iters=10)
                except Exception as e:
                    continue
                res.append((t, tile, st, sk))
    auto = pool_time(lambda: r(B * H * H, Cin), lambda x: hip.conv3x3(x, w, B, H, H, H, H, 1), 2.0 * B * H * H * (Cin + Cout), iters=10)
    best = min(res)
    top = sorted(res)[:5]
    print(f"conv B{B} {H}x{H} {Cin}->{Cout}: auto {auto*1e6:7.1f}us | " + " ".join(f"t{tl}s{st}k{sk}:{t*1e6:6.1f}" for t, tl, st, sk in top) + f" | best {2.0*B*H*H*Cout*9*Cin/best[0]/1e12:.0f}TF ({auto/best[0]:.2f}x)")
