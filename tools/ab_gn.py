import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from bench_kernels import timeit, r, hip
f32 = torch.float32
for B, HW, C in [(16, 4096, 320), (16, 1024, 640), (16, 256, 1280), (16, 64, 1280), (16, 4096, 960), (16, 262144, 128), (16, 65536, 256)]:
    x = r(B * HW, C); g = torch.ones(C, device=x.device); b = torch.zeros(C, device=x.device)
    t1 = timeit(lambda: hip.groupnorm_fwd(x, None, g, b, B, HW, 32, 1e-5, True))
    t2 = timeit(lambda: hip.groupnorm_fwd_unfused(x, None, g, b, B, HW, 32, 1e-5, True))
    print(f"GN B{B} HW{HW} C{C}: fused {t1*1e6:7.1f} us  unfused {t2*1e6:7.1f} us")
