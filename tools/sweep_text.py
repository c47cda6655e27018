"""tile / split-K sweep for the CLIP text encoder GEMMs (M = 16 x 77 = 1232 rows)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from bench_kernels import timeit, r, hip
for M, N, K in [(1232, 768, 3072), (1232, 3072, 768), (1232, 768, 768), (1232, 2304, 768), (1232, 768, 2304), (1024, 1280, 1280), (1232, 768, 2560), (1232, 2560, 768)]:
    a, b = r(M, K), r(N, K)
    res = []
    for tile, sk in [(0, 0), (64, 1), (64, 2), (64, 3), (64, 4), (128, 1), (128, 2), (128, 3), (128, 4), (128, 6)]:
        t = timeit(lambda: hip.gemm(a, b, tile=tile, splitk=sk))
        res.append(f"t{tile}s{sk}:{t*1e6:5.1f}")
    print(f"gemm {M} {N} {K}: " + "  ".join(res))
