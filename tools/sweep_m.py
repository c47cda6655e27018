import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from bench_kernels import timeit, r, hip
for N, K in [(320, 320), (640, 640), (1280, 1280)]:
    for M in (2048, 4096, 8192, 16384, 32768, 65536, 131072):
        a, b = r(M, K), r(N, K)
        t = timeit(lambda: hip.gemm(a, b))
        t2 = timeit(lambda: hip.gemm(a, b, residual=a[:, :N].contiguous() if K >= N else None)) if K >= N else 0
        print(f"gemm M{M} N{N} K{K}: {t*1e6:6.1f} us  {2.0*M*N*K/t/1e12:6.1f} TF  {(M*K+M*N)*2/t/1e12:5.2f} TB/s   (+residual {t2*1e6:6.1f} us)")
