"""conv_strip_kernel against torch's fp32 convolution and against the tile kernel (E4T_CONV_NOSTRIP=1 in a second process gives the A/B timing)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
import torch.nn.functional as F
from e4t import ops
dev = torch.device("cuda:0"); hip = ops.HipBackend(); bf16 = torch.bfloat16
torch.manual_seed(0)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
bad = 0
for B, H, W, Cin, Cout, tile in ([(2, 512, 512, 128, 128, 0), (1, 9, 256, 64, 128, 5256), (2, 5, 512, 64, 256, 5256), (3, 3, 768, 192, 128, 5256), (1, 1, 256, 64, 128, 5256), (2, 64, 256, 128, 256, 5256), (3, 16, 16, 64, 128, 5256), (2, 32, 32, 128, 256, 5256), (2, 64, 64, 64, 320, 5256), (1, 8, 128, 64, 128, 5256), (5, 16, 64, 128, 128, 5256)]
         + ([(2, 64, 256, 128, 256, 512), (3, 16, 16, 64, 256, 512), (2, 32, 32, 128, 512, 512), (2, 64, 64, 64, 320, 512), (1, 8, 128, 64, 128, 512), (1, 4, 512, 128, 384, 512)]
            if True else [])):      # (experimental build: the 16-wave 256 x 256 variant in place of the ping-pong kernel)
    x = (torch.randn(B * H * W, Cin, device=dev) * 0.5).to(bf16)
    w = (torch.randn(Cout, 9 * Cin, device=dev) * (9 * Cin) ** -0.5).to(bf16)
    bias = torch.randn(Cout, device=dev)
    res = (torch.randn(B * H * W, Cout, device=dev) * 0.5).to(bf16)
    y = hip.conv3x3(x, w, B, H, W, H, W, 1, bias=bias, residual=res, tile=tile, colstats=True)
    y = y[0] if isinstance(y, tuple) else y
    xr = x.float().view(B, H, W, Cin).permute(0, 3, 1, 2)
    wr = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xr, wr, bias=bias, padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Cout) + res.float()
    e = rel(y.float(), ref)
    # borders separately: first / last column and row of every image
    yv, rv = y.float().view(B, H, W, Cout), ref.view(B, H, W, Cout)
    eb = max(rel(yv[:, :, 0], rv[:, :, 0]), rel(yv[:, :, -1], rv[:, :, -1]), rel(yv[:, 0], rv[:, 0]), rel(yv[:, -1], rv[:, -1]))
    cs = getattr(y, "_e4t_colstats", None)
    ec = 0.0
    if cs is not None:
        blk = y.float().view(-1, 32, Cout)
        ec = max(rel(cs[..., 0].reshape(-1, Cout), blk.sum(1)), rel(cs[..., 1].reshape(-1, Cout), (blk * blk).sum(1))) if cs.dim() >= 3 else -1.0
    ok = e < 4e-3 and eb < 4e-3
    bad += not ok
    print(f"conv B{B} {H}x{W} {Cin}->{Cout} tile{tile}: rel {e:.2e} borders {eb:.2e} colstats {ec:.2e} {'ok' if ok else 'BAD'}", flush=True)
# bitwise repeatability
x = (torch.randn(16 * 512 * 512, 128, device=dev) * 0.5).to(bf16); w = (torch.randn(128, 9 * 128, device=dev) / 34).to(bf16)
y0 = hip.conv3x3(x, w, 16, 512, 512, 512, 512, 1); y0 = y0[0] if isinstance(y0, tuple) else y0
y0 = y0.clone()
for _ in range(3):
    y1 = hip.conv3x3(x, w, 16, 512, 512, 512, 512, 1); y1 = y1[0] if isinstance(y1, tuple) else y1
    bad += not torch.equal(y0, y1)
print("repeat bitwise", "ok" if not bad else "BAD")
sys.exit(1 if bad else 0)
