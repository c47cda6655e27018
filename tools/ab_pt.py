"""A/B of the 512x128 ping-pong tile against the 128x128 tile on the VAE's 128-channel conv shapes (B=16, 512 px)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd")]
import torch
from e4t import ops, _C
hip = ops.backend(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def run(B, H, W, Cin, Cout, mode, Ho, Wo, tile, colstats):
    x = (torch.randn((B * H * W, Cin), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn((Cout, 9 * Cin), generator=g, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    bias = torch.randn(Cout, generator=g, device=dev)
    res = (torch.randn((B * Ho * Wo, Cout), generator=g, device=dev)).to(torch.bfloat16)
    f = lambda: hip.conv3x3(x, w, B, H, W, Ho, Wo, mode, bias=bias, residual=res, tile=tile, colstats=colstats)
    y = f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return ms, 2.0 * B * Ho * Wo * Cout * 9 * Cin / ms / 1e9, y
for name, c in [("512^2 128->128 s1", (16, 512, 512, 128, 128, _C.CONV_S1, 512, 512)), ("512^2 128->128 s2a", (16, 512, 512, 128, 128, _C.CONV_S2A, 256, 256)),
                ("256^2 128->128 s1 (B=4 slice)", (4, 512, 512, 128, 128, _C.CONV_S1, 512, 512)), ("768^2 128->128 s1 B=8", (8, 768, 768, 128, 128, _C.CONV_S1, 768, 768))]:
    a = run(*c, 128, True); b = run(*c, 640, True); d = run(*c, 0, True)
    err = float((a[2].float() - b[2].float()).abs().max())
    print(f"{name:32s} t128 {a[0]:7.3f} ms {a[1]:7.1f} TF | t640 {b[0]:7.3f} ms {b[1]:7.1f} TF | auto {d[0]:7.3f} ms | max|diff| {err:.3e}", flush=True)
