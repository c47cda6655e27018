"""Tile / split-K sweep for the M = 4112 (16 x 257 tokens) ViT-H GEMMs and the 4096-row UNet ones."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
from bench_kernels import timeit, r, hip
for M, N, K in [(4112, 1280, 5120), (4112, 5120, 1280), (4112, 3840, 1280), (4112, 1280, 1280), (4096, 1280, 1280), (4096, 1280, 10240),
                (4096, 10240, 1280), (16384, 640, 2560), (16384, 5120, 640)]:
    a, b = r(M, K), r(N, K)
    for tile, sk in [(0, 0), (128, 1), (128, 2), (128, 3), (160, 1), (160, 2), (64, 1), (256, 1)]:
        try:
            t = timeit(lambda: hip.gemm(a, b, tile=tile, splitk=sk))
            print(f"gemm {M} {N} {K} tile{tile} sk{sk}: {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TF")
        except Exception as e:
            print("ERR", tile, sk, e)
