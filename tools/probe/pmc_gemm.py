"""Driver for tools/probe/pmc_gemm.sh: a few launches each of the step's dominant GEMM / conv shapes with explicit tile codes, so that
the SQ counters of one kernel symbol come from one shape.  usage: python tools/probe/pmc_gemm.py"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import ops  # noqa: E402

hip = ops.HipBackend()
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
# conv 128x128 Cin512 -> 512 (pp, code 512); conv 64x64 Cin320 -> 320 (pq, code 2320); gemm M65536 N320 K2560 (pq); gemm M4096 N1280 K1280 (128)
x = r(16 * 128 * 128, 512); w = r(512, 9 * 512)
for _ in range(6):
    hip.conv3x3(x, w, 16, 128, 128, 128, 128, 1, tile=512, splitk=1)
x = r(16 * 64 * 64, 320); w = r(320, 9 * 320)
for _ in range(6):
    hip.conv3x3(x, w, 16, 64, 64, 64, 64, 1, tile=2320, splitk=1)
a = r(4096, 1280); w = r(1280, 1280)
for _ in range(6):
    hip.gemm(a, w, tile=128)
torch.cuda.synchronize()
