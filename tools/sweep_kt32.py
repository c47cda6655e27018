"""GPU: the 32-wide K-tile, 4-stage variant of the 128x128 / 64x64 DMA GEMM (tile codes 5128 / 5064) against the default, cold operands."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
hip = ops.HipBackend()
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf16)
# bit-exactness first: same accumulation order as the 64-wide kernel
for M, N, K in [(4096, 1280, 1280), (1000, 640, 328), (4112, 1280, 64), (300, 320, 192), (513, 200, 72), (8192, 1280, 512)]:
    a, w = r(M, K), r(N, K)
    for tile in (128, 64):
        base = hip.gemm(a, w, tile=tile, splitk=1)
        got = hip.gemm(a, w, tile=5000 + tile, splitk=1)
        assert torch.equal(base, got), (M, N, K, tile, float((base.float() - got.float()).abs().max()))
    b2 = hip.gemm(a, w, tile=128, splitk=2) if K >= 256 else None
    if b2 is not None:
        assert torch.equal(b2, hip.gemm(a, w, tile=5128, splitk=2)), ("splitk", M, N, K)
x, w = r(2 * 16 * 16, 128), r(256, 9 * 128)
for tile in (128, 64):
    assert torch.equal(hip.conv3x3(x, w, 2, 16, 16, 16, 16, 1, tile=tile), hip.conv3x3(x, w, 2, 16, 16, 16, 16, 1, tile=5000 + tile)), tile
a, a2, w = r(1000, 128), r(1000, 192), r(320, 320)
assert torch.equal(hip.gemm(a, w, a2=a2, tile=128), hip.gemm(a, w, a2=a2, tile=5128))
print("32-wide K-tile kernels: bit-identical to the 64-wide ones")


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


for M, N, K in [(65536, 320, 320), (16384, 640, 640), (4096, 1280, 1280), (4096, 1280, 5120), (4096, 5120, 1280), (4096, 10240, 1280), (65536, 2560, 320),
                (16384, 5120, 640), (4112, 5120, 1280), (4112, 1280, 5120), (4112, 3840, 1280), (4112, 1280, 1280), (1024, 1280, 1280), (1232, 3072, 768), (8192, 8192, 8192)]:
    w = r(N, K)
    mk = lambda: (r(M, K), torch.empty((M, N), dtype=bf16, device=dev))
    res = {}
    for code in (0, 128, 5128, 64, 5064):
        res[code] = pool_time(mk, lambda b: hip.gemm(b[0], w, out=b[1], tile=code), 2.0 * M * (K + N))
    print(f"gemm M{M} N{N} K{K}: auto {res[0]*1e6:7.1f} | t128 {res[128]*1e6:7.1f} kt32 {res[5128]*1e6:7.1f} ({res[128]/res[5128]:.2f}x) | t64 {res[64]*1e6:7.1f} kt32 {res[5064]*1e6:7.1f} ({res[64]/res[5064]:.2f}x) | best kt32 {2.0*M*N*K/min(res[5128],res[5064])/1e12:.0f} TF")
for B, H, Cin, Cout in [(16, 64, 320, 320), (16, 32, 640, 640), (16, 16, 1280, 1280), (16, 8, 1280, 1280), (16, 512, 128, 128), (16, 32, 1280, 640)]:
    w = r(Cout, 9 * Cin)
    mk = lambda: r(B * H * H, Cin)
    res = {}
    for code in (0, 128, 5128):
        res[code] = pool_time(mk, lambda x: hip.conv3x3(x, w, B, H, H, H, H, 1, tile=code), 2.0 * B * H * H * (Cin + Cout), iters=8)
    print(f"conv B{B} {H}x{H} {Cin}->{Cout}: auto {res[0]*1e6:7.1f} | t128 {res[128]*1e6:7.1f} kt32 {res[5128]*1e6:7.1f} ({res[128]/res[5128]:.2f}x) {2.0*B*H*H*Cout*9*Cin/res[5128]/1e12:.0f} TF")
