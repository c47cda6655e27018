"""Debug: per-phase timeline of the ping-pong GEMM (needs a -DPP_TRACE build: tools/pp_trace.sh)."""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd")]
import torch
from e4t import ops
hip = ops.HipBackend(); dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
a, b = r(4096, 4096), r(4096, 4096)
for _ in range(3): hip.gemm(a, b, tile=512, splitk=1)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 6 * 64))()
hip.lib.e4t_debug_pp_trace.argtypes = [C.c_void_p]
print("rc", hip.lib.e4t_debug_pp_trace(buf))
starts = [[buf[g * 384 + i * 6] for i in range(40)] for g in range(2)]
for g in range(2):
    print(f"group {g} phase periods:", [starts[g][i + 1] - starts[g][i] for i in range(39)])
if os.environ.get("PP_MIN"): sys.exit(0)
for g in range(2):
    base = buf[g * 384]
    print(f"group {g}: columns = L-start, issued, waited, after-barrier1, mfma-issued, after-barrier2 (cycles rel. to first stamp)")
    for i in range(24):
        row = [buf[g * 384 + i * 6 + k] - base for k in range(6)]
        d = [row[k + 1] - row[k] for k in range(5)]
        print(f"  ph{i:2d} start {row[0]:7d} | reads+issue {d[0]:4d} vmcnt {d[1]:4d} bar1 {d[2]:4d} mfma {d[3]:4d} bar2 {d[4]:4d}")
