"""Kernel micro-benchmarks on the GPU box (not a test): representative shapes of the SD-1.4 B=16 step.
    python tests/bench_kernels.py [gemm] [conv] [attn] [gn]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(os.path.dirname(HERE), "e4t-diffusion_amd"), HERE]

import torch  # noqa: E402
from e4t import ops  # noqa: E402

bf16 = torch.bfloat16
dev = torch.device("cuda:0")
hip = ops.HipBackend()


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def r(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(bf16)


def bench_gemm():
    print("== gemm (M, N, K)")
    for M, N, K in [(65536, 320, 320), (65536, 2560, 320), (65536, 320, 1280), (16384, 640, 640), (16384, 5120, 640), (16384, 640, 2560),
                    (4096, 1280, 1280), (4096, 10240, 1280), (4096, 1280, 5120), (4112, 3840, 1280), (4112, 5120, 1280), (4112, 1280, 5120),
                    (1232, 640, 768), (8192, 8192, 8192)]:
        a, b = r(M, K), r(N, K)
        for tile in (160, 128, 512):
            t = timeit(lambda: hip.gemm(a, b, tile=tile, splitk=1))
            print(f"  {M:6d} {N:6d} {K:6d} tile{tile:3d}: {t*1e6:9.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF")


def bench_conv():
    print("== conv3x3 (B, H, Cin, Cout)")
    for B, H, Cin, Cout in [(16, 64, 320, 320), (16, 64, 960, 320), (16, 64, 640, 320), (16, 32, 640, 640), (16, 32, 1920, 640), (16, 16, 1280, 1280),
                            (16, 16, 2560, 1280), (16, 8, 1280, 1280), (16, 8, 2560, 1280), (16, 512, 128, 128), (16, 256, 256, 256), (16, 128, 512, 512)]:
        x, w = r(B * H * H, Cin), r(Cout, 9 * Cin)
        for tile in (160, 128, 512):
            t = timeit(lambda: hip.conv3x3(x, w, B, H, H, H, H, 1, tile=tile, splitk=1), iters=5)
            print(f"  B{B} {H:3d}x{H:<3d} {Cin:5d}->{Cout:5d} tile{tile:3d}: {t*1e6:9.1f} us  {2.0*B*H*H*Cout*9*Cin/t/1e12:7.1f} TF")


def bench_attn():
    print("== attention (B, H, T, S, DH)")
    for B, H, T, S, DH in [(16, 8, 4096, 4096, 40), (16, 8, 4096, 77, 40), (16, 8, 1024, 1024, 80), (16, 8, 256, 256, 160), (16, 16, 257, 257, 80)]:
        d = H * DH
        if T == S:
            qkv = r(B * T, 3 * d); q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            g = torch.empty_like(qkv); dq, dk, dv = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
        else:
            q = r(B * T, d); kv = r(B * S, 2 * d); k, v = kv[:, :d], kv[:, d:]
            dq = torch.empty_like(q); g = torch.empty_like(kv); dk, dv = g[:, :d], g[:, d:]
        o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5)
        do = r(B * T, d)
        tf = timeit(lambda: hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5), iters=5)
        tb = timeit(lambda: hip.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, B, H, T, S, DH, DH ** -0.5), iters=5)
        fl = 4.0 * B * H * T * S * DH
        print(f"  B{B} H{H} T{T} S{S} dh{DH}: fwd {tf*1e6:9.1f} us {fl/tf/1e12:6.1f} TF | bwd {tb*1e6:9.1f} us {2.5*fl/tb/1e12:6.1f} TF")


def bench_gn():
    print("== groupnorm (B, HW, C)  [GB/s = algorithmic bytes: stats read + apply read+write]")
    for B, HW, C in [(16, 4096, 320), (16, 4096, 960), (16, 1024, 640), (16, 256, 1280), (16, 262144, 128), (16, 65536, 256)]:
        x = r(B * HW, C)
        ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        t = timeit(lambda: hip.groupnorm_fwd(x, None, ga, be, B, HW, 32, 1e-5, True), iters=5)
        print(f"  B{B} HW{HW} C{C}: fwd {t*1e6:9.1f} us  {3.0*x.numel()*2/t/1e9:8.1f} GB/s")
        y, st = hip.groupnorm_fwd(x, None, ga, be, B, HW, 32, 1e-5, True)
        dy = r(B * HW, C)
        t = timeit(lambda: hip.groupnorm_bwd(x, None, dy, st, ga, be, None, B, HW, 32, True), iters=5)
        print(f"  {'':22s} bwd {t*1e6:9.1f} us  {5.0*x.numel()*2/t/1e9:8.1f} GB/s")


if __name__ == "__main__":
    want = sys.argv[1:] or ["gemm", "conv", "attn", "gn"]
    for w in want:
        globals()["bench_" + w]()
