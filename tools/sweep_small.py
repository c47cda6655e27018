import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
from bench_kernels import timeit, r, hip
for B, H, Cin, Cout in [(16, 8, 1280, 1280), (16, 8, 2560, 1280), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 32, 640, 640)]:
    x, w = r(B * H * H, Cin), r(Cout, 9 * Cin)
    for tile, sk in [(64, 1), (64, 2), (64, 4), (128, 1), (128, 2), (128, 4), (128, 8)]:
        t = timeit(lambda: hip.conv3x3(x, w, B, H, H, H, H, 1, tile=tile, splitk=sk), iters=5)
        print(f"conv B{B} {H}x{H} {Cin}->{Cout} tile{tile} sk{sk}: {t*1e6:8.1f} us {2.0*B*H*H*Cout*9*Cin/t/1e12:7.1f} TF")
for M, N, K in [(1024, 1280, 1280), (1024, 10240, 1280), (1024, 1280, 5120), (4096, 1280, 1280), (4096, 3840, 1280), (1232, 1280, 768), (1232, 2560, 768), (16384, 640, 640), (16384, 1920, 640)]:
    a, b = r(M, K), r(N, K)
    for tile, sk in [(64, 1), (64, 2), (128, 1), (128, 2), (128, 4)]:
        t = timeit(lambda: hip.gemm(a, b, tile=tile, splitk=sk))
        print(f"gemm {M} {N} {K} tile{tile} sk{sk}: {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TF")
