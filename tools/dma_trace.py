"""Debug: where a workgroup of the LDS-DMA GEMM spends its life (needs a -DDMA_TRACE build: tools/dma_trace.sh M N K [tile])."""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd")]
import torch
from e4t import ops
hip = ops.HipBackend(); dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 320, 320)
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pool = [(r(M, K), torch.empty((M, N), dtype=torch.bfloat16, device=dev)) for _ in range(8)]
w = r(N, K)
for a, c in pool: hip.gemm(a, w, out=c, tile=tile)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hip.gemm(pool[0][0], w, out=pool[0][1], tile=tile); e1.record(); torch.cuda.synchronize()
print(f"M{M} N{N} K{K} tile{tile}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us (with stamps)")
buf = (C.c_ulonglong * (128 * 16))()
hip.lib.e4t_debug_dma_trace.argtypes = [C.c_void_p]
assert hip.lib.e4t_debug_dma_trace(buf) == 0
rows = [[buf[g * 16 + k] for k in range(16)] for g in range(128)]
rows = [x for x in rows if x[0] and x[12]]
t0 = min(x[0] for x in rows)
nk = (K + 63) // 64
print(f"{len(rows)} traced workgroups; cycles (100 MHz-ish constant clock? see ratio below) relative to the earliest start")
print("   start  | setup+issue | wait tile0 | per K-tile ... | loop->sync | epilogue | total")
import statistics
for x in rows[:8] + rows[-8:]:
    its = [x[2 + i] for i in range(min(nk, 8))]
    d_it = [its[0] - x[1]] + [its[i + 1] - its[i] for i in range(len(its) - 1)]
    print(f"  {x[0] - t0:7d} | {x[1] - x[0]:6d} | " + " ".join(f"{d:5d}" for d in d_it) + f" | last->10 {x[10] - its[-1]:5d} sync {x[11] - x[10]:5d} | epi {x[12] - x[11]:6d} = preload {x[13] - x[11]:5d} stage {x[14] - x[13]:5d} sync {x[15] - x[14]:5d} stores {x[12] - x[15]:5d} | {x[12] - x[0]:7d}")
tot = [x[12] - x[0] for x in rows]
print("median total", statistics.median(tot), "median epilogue", statistics.median([x[12] - x[11] for x in rows]),
      "median setup+issue", statistics.median([x[1] - x[0] for x in rows]), "median wait tile0", statistics.median([x[2] - x[1] for x in rows]),
      "span (last end - first start)", max(x[12] for x in rows) - t0)
