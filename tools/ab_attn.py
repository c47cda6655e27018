"""Attention micro-benchmark for compile-time variants (E4T_LIB=<variant .so>): forward and backward of the step's attention shapes,
graph-replayed (the 15 us ctypes launch is not what is measured), with a numerics check of every output against the fp32 restatement
on a smaller problem.   python tools/ab_attn.py [label]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import ops  # noqa: E402
from emu_backend import EmuBackend  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("E4T_LIB", "default"))
dev = torch.device("cuda:0")
hip, emu = ops.HipBackend(), EmuBackend()
bf16 = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf16)


def graph_time(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-20))


# numerics (small): vs the fp32 restatement
B, H, T, S, DH = 2, 4, 320, 320, 40
d = H * DH
qkv = r(B * T, 3 * d); q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5)
o2, lse2 = emu.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5)
do = r(B * T, d)
g = torch.empty_like(qkv); g2 = torch.empty_like(qkv)
hip.attention_bwd(q, k, v, o, do, lse, g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, H, T, S, DH, DH ** -0.5)
emu.attention_bwd(q, k, v, o2, do, lse2, g2[:, :d], g2[:, d:2 * d], g2[:, 2 * d:], B, H, T, S, DH, DH ** -0.5)
errs = dict(o=rel(o, o2), dq=rel(g[:, :d], g2[:, :d]), dk=rel(g[:, d:2 * d], g2[:, d:2 * d]), dv=rel(g[:, 2 * d:], g2[:, 2 * d:]))
print(f"[{label}] numerics dh40 T320: " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()), "OK" if max(errs.values()) < 1.5e-2 else "FAIL")

for B, H, T, S, DH in [(16, 8, 4096, 4096, 40), (16, 8, 4096, 77, 40), (16, 8, 1024, 1024, 80)]:
    d = H * DH
    if T == S:
        qkv = r(B * T, 3 * d); q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        g = torch.empty_like(qkv); dq, dk, dv = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
    else:
        q = r(B * T, d); kv = r(B * S, 2 * d); k, v = kv[:, :d], kv[:, d:]
        dq = torch.empty_like(q); g = torch.empty_like(kv); dk, dv = g[:, :d], g[:, d:]
    o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5)
    do = r(B * T, d)
    out = torch.empty_like(o)
    tf = graph_time(lambda: hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5, out=out), iters=10)
    tb = graph_time(lambda: hip.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, B, H, T, S, DH, DH ** -0.5), iters=6)
    fl = 4.0 * B * H * T * S * DH
    print(f"[{label}] B{B} H{H} T{T} S{S} dh{DH}: fwd {tf:8.1f} us {fl / tf / 1e6:6.1f} TF | bwd {tb:8.1f} us {2.5 * fl / tb / 1e6:6.1f} TF", flush=True)
