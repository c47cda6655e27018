import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from bench_kernels import timeit, r, hip
f32 = torch.float32
for K, M, N in [(65536, 960, 320), (65536, 320, 320), (16384, 1920, 640), (16384, 640, 640), (4096, 3840, 1280), (4096, 1280, 1280), (1232, 2560, 768),
                (65536, 2560, 320), (65536, 320, 1280), (4096, 10240, 1280)]:
    a, b = r(K, M), r(K, N)
    def old():
        return hip.gemm(hip.transpose(a, pad_to=K), hip.transpose(b, pad_to=K), out_dtype=f32)
    t1 = timeit(lambda: hip.gemm_tn(a, b))
    t2 = timeit(old)
    print(f"wgrad K{K} M{M} N{N}: tn {t1*1e6:7.1f} us ({2.0*M*N*K/t1/1e12:6.1f} TF)   transposes+nt {t2*1e6:7.1f} us")
