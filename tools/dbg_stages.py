"""GPU stress: every stage count of the 64 / 128 / 160 DMA GEMM tiles, many repetitions on grids of several waves — any
run-to-run difference is a race (the kernels are deterministic)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
hip = ops.HipBackend()
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
reps = int(os.environ.get("REPS", "60"))
bad_total = 0
for M, N, K in [(4096, 1280, 320), (8192, 1280, 512), (8192, 2560, 1280), (16384, 640, 640)]:
    a, w = r(M, K), r(N, K)
    ref = hip.gemm(a, w, tile=128)
    for code in (64, 3064, 4064, 128, 3128, 4128, 160, 3160, 4160):
        if code % 1000 == 160 and N % 160:
            continue
        bad = 0
        for rep in range(reps):
            y = hip.gemm(a, w, tile=code)
            bad += int((y != ref).view(M // 64, 64, N // 64, 64).any(3).any(1).sum())
        bad_total += bad
        print(f"M{M} N{N} K{K} tile code {code}: {bad} differing 64x64 tiles over {reps} launches")
x, w = r(16 * 32 * 32, 640), r(640, 9 * 640)
ref = hip.conv3x3(x, w, 16, 32, 32, 32, 32, 1, tile=128)
for code in (128, 3128, 4128, 160, 3160, 4160, 64, 3064):
    bad = 0
    for rep in range(20):
        bad += int((hip.conv3x3(x, w, 16, 32, 32, 32, 32, 1, tile=code) != ref).sum() > 0)
    bad_total += bad
    print(f"conv 32x32 640->640 tile code {code}: {bad} differing launches of 20")
print("TOTAL", bad_total)
