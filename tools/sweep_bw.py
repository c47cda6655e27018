"""Is the short-K GEMM bandwidth- or latency-limited?  A-stream rate for thin N, several tiles."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from bench_kernels import timeit, r, hip
M = 65536
for N, K in [(64, 320), (128, 320), (160, 320), (320, 320), (128, 1280), (320, 1280), (128, 5120)]:
    a, b = r(M, K), r(N, K)
    for tile in (64, 128, 160):
        if tile == 160 and N % 160: continue
        t = timeit(lambda: hip.gemm(a, b, tile=tile, splitk=1))
        print(f"gemm M{M} N{N} K{K} tile{tile}: {t*1e6:6.1f} us  A-read {M*K*2/t/1e12:5.2f} TB/s  total {(M*K+M*N)*2/t/1e12:5.2f} TB/s  {2.0*M*N*K/t/1e12:6.1f} TF")
