"""split-K sweep for the under-filled 16x16 / 8x8 UNet convs and long-K GEMMs (128x128 tile, 2 workgroups/CU = 512 slots)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
from bench_kernels import timeit, r, hip
for B, H, Cin, Cout in [(16, 8, 1280, 1280), (16, 8, 2560, 1280), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 16, 1920, 1280), (16, 16, 1280, 640)]:
    x, w = r(B * H * H, Cin), r(Cout, 9 * Cin)
    for sk in (0, 1, 2, 3, 4, 5, 6, 8):
        t = timeit(lambda: hip.conv3x3(x, w, B, H, H, H, H, 1, tile=128 if sk else 0, splitk=sk), iters=8)
        print(f"conv B{B} {H}x{H} {Cin}->{Cout} sk{sk}: {t*1e6:8.1f} us {2.0*B*H*H*Cout*9*Cin/t/1e12:7.1f} TF")
for M, N, K in [(1024, 1280, 1280), (1024, 1280, 5120), (1024, 1280, 10240), (4096, 1280, 5120), (4096, 1280, 10240), (4112, 1280, 5120)]:
    a, b = r(M, K), r(N, K)
    for sk in (0, 1, 2, 3, 4, 6):
        t = timeit(lambda: hip.gemm(a, b, tile=128 if sk else 0, splitk=sk))
        print(f"gemm {M} {N} {K} sk{sk}: {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TF")
