"""Per-kernel parity checks: HIP op (through the C ABI) vs the torch restatement in emu_backend.py,
on the same seeded inputs.  Each check returns a list of (label, rel_l2_error, tolerance).
Used by tests/test_kernels_gpu.py (pytest, -m gpu) and tests/gpu_report.py (prints everything).

Tolerances (stated per check): outputs are bf16 (8 mantissa bits, eps = 2^-8 = 3.9e-3) computed from
bf16 inputs with fp32 accumulation, so the relative L2 error against an fp32 evaluation of the same
bf16 inputs is bounded by ~eps/sqrt(3) per output rounding (~2.3e-3) plus accumulation-order noise;
we allow 6e-3 for single-rounding kernels, 1.5e-2 where an intermediate (P, dS) is also rounded to
bf16 inside the kernel, 1e-5 for fp32-only kernels.
"""
from __future__ import annotations

import math

import torch

from emu_backend import EmuBackend, CONV_S1, CONV_S2, CONV_UP2, CONV_S2T

bf16, f32 = torch.bfloat16, torch.float32
TOL1, TOL2, TOLF = 6e-3, 1.5e-2, 2e-5


def rel(a, b):
    a, b = a.float(), b.float()
    d = (a - b).norm()
    n = b.norm()
    if not torch.isfinite(d):
        return float("inf")
    return float(d / (n + 1e-20))


def rnd(g, *shape, scale=1.0, dtype=bf16, dev="cuda"):
    return (torch.randn(*shape, generator=g, device=dev, dtype=f32) * scale).to(dtype)


def gen(seed, dev="cuda"):
    return torch.Generator(device=dev).manual_seed(seed)


def check_probe(hip, emu, dev):
    rows, cols = hip.probe_mfma(dev)
    l = torch.arange(64, device=dev)[:, None]
    r = torch.arange(16, device=dev)[None, :]
    exp_rows = ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5) + 1).float()
    exp_cols = ((l & 31) + 1).float().expand(64, 16)
    return [("mfma32 row map", rel(rows, exp_rows), 0.0), ("mfma32 col map", rel(cols, exp_cols), 0.0)]


def _product_tile(hip, code):
    """False for the tile codes of the measured-and-rejected GEMM variants, which only an E4T_EXPERIMENTAL=1 build of the library
    carries (32-wide-K 64 / 128 tiles, 512 x 128 ping-pong, persistent streaming kernels): the default library maps them to product
    tiles, so checking them there would re-check those tiles under another label.  (The 3 / 4-stage 64 / 128 / 160 tiles are product
    code since round 4: the planner picks them for grids under one round.)"""
    if hip.lib.e4t_build_flags() & 1:
        return True
    return code not in (640, 1128, 1160, 5064, 5128)


def check_gemm(hip, emu, dev):
    out = []
    cases = [  # M, N, K, tile, splitk
        (256, 256, 128, 0, 0), (200, 72, 64, 0, 0), (1000, 320, 320, 128, 1), (130, 200, 1032, 64, 3),
        (16, 1280, 1280, 0, 0), (4096, 640, 2560, 0, 0), (64, 64, 4096, 0, 0), (1000, 200, 328, 256, 1), (2048, 256, 64, 256, 1),
        (700, 320, 1280, 256, 2), (900, 320, 384, 160, 1), (300, 480, 128, 160, 2), (513, 200, 72, 160, 1),
        (512, 512, 512, 512, 1), (700, 520, 256, 512, 1), (300, 256, 64, 512, 1), (1000, 300, 128, 512, 1), (513, 1000, 1152, 512, 2),
        (2048, 1280, 1920, 512, 0), (257, 64, 192, 512, 1),
        (4096, 1280, 10240, 0, 0), (4096, 2560, 8192, 0, 0),       # auto: K-deep, 64-255 tiles of 256 x 256 -> ping-pong + split-K (3 | 1)
        (1024, 128, 512, 640, 1), (700, 128, 1152, 640, 1), (1300, 384, 256, 640, 1), (513, 100, 64, 640, 1), (2048, 128, 2048, 640, 0), (4096, 128, 4096, 640, 3),
        # persistent streaming tiles (gemm_ps.hip): 256 x 160 / 256 x 128; ragged M / N, one K-tile, several units per workgroup
        (1000, 320, 320, 1160, 1), (513, 200, 128, 1128, 1), (4096, 640, 2560, 1160, 0), (70000, 320, 320, 1160, 1), (33000, 256, 192, 1128, 1),
        (300, 480, 128, 1160, 1), (2048, 128, 64, 1128, 1), (256, 160, 64, 1160, 1), (66000, 100, 64, 1128, 1), (4112, 1280, 1280, 1160, 1),
        (1000, 256, 328, 1128, 1),                                   # K % 64 != 0: falls back to the 128 x 128 tile
        (900, 384, 512, 5256, 1), (5000, 640, 320, 5256, 1),         # experimental: 256 x 128, 32-wide K-tiles
        # k-step-phased ping-pong tile (gemm_pq_kernel): 256 x 320 (wave tile 64 x 160); ragged M / N, one K-tile, split-K
        (1000, 320, 320, 2320, 1), (700, 640, 256, 2320, 1), (300, 200, 64, 2320, 1), (4096, 1280, 2560, 2320, 0), (513, 1000, 1152, 2320, 2),
        (65536, 320, 320, 2320, 1), (16384, 640, 640, 2320, 0),
    ]
    for i, (M, N, K, tile, sk) in enumerate(cases):
        if not _product_tile(hip, tile):
            continue
        g = gen(10 + i, dev)
        a, b = rnd(g, M, K, dev=dev), rnd(g, N, K, scale=K ** -0.5, dev=dev)
        bias = rnd(g, N, dtype=f32, dev=dev)
        res = rnd(g, M, N, dev=dev)
        y = hip.gemm(a, b, bias=bias, residual=res, tile=tile, splitk=sk)
        yr = emu.gemm(a, b, bias=bias, residual=res)
        out.append((f"gemm {M}x{N}x{K} t{tile} s{sk} bias+res", rel(y, yr), TOL1))
    # TN: contraction over the rows (weight gradients)
    for i, (K, M, N, sk) in enumerate([(64, 128, 128, 1), (1000, 320, 320, 0), (4096, 960, 320, 0), (777, 200, 72, 3), (65536, 320, 320, 0),
                                       (1232, 2560, 768, 0), (130, 8, 1280, 1)]):
        g = gen(40 + i, dev)
        a, b = rnd(g, K, M + 8, dev=dev)[:, :M], rnd(g, K, N, scale=K ** -0.5, dev=dev)
        out.append((f"gemm_tn K{K} M{M} N{N} s{sk}", rel(hip.gemm_tn(a, b, splitk=sk), emu.gemm_tn(a, b)), TOLF * 50))
    g = gen(48, dev)
    a, b = rnd(g, 500, 320, dev=dev), rnd(g, 500, 328, scale=0.05, dev=dev)
    c1 = rnd(g, 320, 328, dtype=f32, dev=dev); c2 = c1.clone()
    hip.gemm_tn(a, b, out=c1, accum=True); emu.gemm_tn(a, b, out=c2, accum=True)
    out.append(("gemm_tn fp32 accumulate", rel(c1, c2), TOLF * 50))
    wide = torch.zeros(320, 700, dtype=f32, device=dev); wide2 = wide.clone()
    hip.gemm_tn(a, b, out=wide[:, 100:428]); emu.gemm_tn(a, b, out=wide2[:, 100:428])
    out.append(("gemm_tn into a column slice", rel(wide, wide2), TOLF * 50))
    # column statistics left by the epilogue for the consuming GroupNorm (all tile variants; with / without residual)
    for i, (M, N, K, tile) in enumerate([(256, 128, 128, 0), (4096, 320, 320, 160), (1024, 640, 1280, 128), (2048, 512, 2304, 512), (192, 72, 64, 64),
                                         (4096, 320, 320, 1160), (8192, 256, 128, 1128), (66560, 640, 192, 1160), (34816, 128, 64, 1128),
                                         (4096, 320, 320, 2320), (8192, 640, 640, 2320)]):
        g = gen(60 + i, dev)
        a, b = rnd(g, M, K, dev=dev), rnd(g, N, K, scale=K ** -0.5, dev=dev)
        res = rnd(g, M, N, dev=dev) if i % 2 == 0 else None
        y = hip.gemm(a, b, bias=rnd(g, N, dtype=f32, dev=dev), residual=res, tile=tile, splitk=1, colstats=True)
        cs = getattr(y, "_e4t_colstats", None)
        blk = y.float().reshape(M // 32, 32, N)
        ref = torch.stack([blk.sum(1), (blk * blk).sum(1)], dim=-1)
        out.append((f"gemm {M}x{N}x{K} t{tile} colstats", rel(cs, ref) if cs is not None else 1.0, 1e-5))
    g = gen(30, dev)
    # two-source A, gelu, fp32 out, accumulate, rowbias
    M, N, K1, K2 = 384, 192, 128, 64
    a1, a2, b = rnd(g, M, K1, dev=dev), rnd(g, M, K2, dev=dev), rnd(g, N, K1 + K2, scale=0.07, dev=dev)
    out.append(("gemm two-source A", rel(hip.gemm(a1, b, a2=a2), emu.gemm(a1, b, a2=a2)), TOL1))
    out.append(("gemm two-source A t512", rel(hip.gemm(a1, b, a2=a2, tile=512), emu.gemm(a1, b, a2=a2)), TOL1))
    out.append(("gemm gelu fp32-out t512", rel(hip.gemm(a1, b[:, :K1].contiguous(), gelu=True, out_dtype=f32, tile=512),
                                                emu.gemm(a1, b[:, :K1].contiguous(), gelu=True, out_dtype=f32)), TOLF * 50))
    out.append(("gemm gelu", rel(hip.gemm(a1, b[:, :K1].contiguous(), gelu=True), emu.gemm(a1, b[:, :K1].contiguous(), gelu=True)), TOL1))
    for t in [t for t in (3064, 512, 2320, 1128, 1160) if _product_tile(hip, t)]:      # the GENERAL epilogue instantiations of the 3-stage 64 tile, the ping-pong (/ persistent) kernels
        out.append((f"gemm gelu t{t}", rel(hip.gemm(a1, b[:, :K1].contiguous(), gelu=True, tile=t), emu.gemm(a1, b[:, :K1].contiguous(), gelu=True)), TOL1))
        out.append((f"gemm two-source A t{t}", rel(hip.gemm(a1, b, a2=a2, tile=t), emu.gemm(a1, b, a2=a2)), TOL1))
    c0 = rnd(g, M, N, dtype=f32, dev=dev)
    c1, c2 = c0.clone(), c0.clone()
    hip.gemm(a1, b[:, :K1].contiguous(), out=c1, accum=True, alpha=0.5)
    emu.gemm(a1, b[:, :K1].contiguous(), out=c2, accum=True, alpha=0.5)
    out.append(("gemm fp32 out + accumulate + alpha", rel(c1, c2), TOLF * 50))
    # tail rows (round 5): M = 128 k + r, r <= 32 — the tile grid covers M - r rows, gemm_tail() computes the rest at the end of the launch
    # (the CLIP-ViT's 16 x 257 = 4112 token rows).  The tail rows are ALSO compared on their own: 16 of 4112 rows move the whole-matrix
    # rel-L2 by 6 % when they are garbage but a subtler error would drown in it.
    gt = gen(35, dev)
    import ctypes as _ct
    from e4t import _C as _Cm
    for (Mt, Nt, Kt, kw) in [(4112, 1280, 1280, dict(f32=True)), (4112, 5120, 1280, dict(gelu=True)), (4112, 1280, 5120, dict(f32=True)), (4112, 3840, 1280, {}),
                             (8224, 1280, 1280, dict(res=True)), (8224, 5120, 1280, dict(gelu=True)), (4096 + 1, 1280, 1280, dict(res=True)), (4096 + 31, 3840, 1280, {}),
                             (4112, 1280, 1296, dict(f32=True)), (2056, 1280, 5120, {})]:
        at, bt = rnd(gt, Mt, Kt, dev=dev), rnd(gt, Nt, Kt, scale=Kt ** -0.5, dev=dev)
        bi = rnd(gt, Nt, dtype=f32, dev=dev)
        res = rnd(gt, Mt, Nt, dtype=f32 if kw.get("f32") else bf16, dev=dev) if (kw.get("f32") or kw.get("res")) else None
        args = dict(bias=bi, residual=res, gelu=bool(kw.get("gelu")), out_dtype=f32 if kw.get("f32") else bf16)
        d = _Cm.GemmDesc(M=Mt, N=Nt, K=Kt, K1=Kt, lda=Kt, ldb=Kt, ldc=Nt, batch=1, alpha=1.0,
                         flags=(_Cm.OUT_F32 | _Cm.RES_F32 if kw.get("f32") else 0) | (_Cm.ACT_GELU if kw.get("gelu") else 0), residual=(1 << 20) if res is not None else None)
        pl = _Cm.GemmPlan()
        hip.lib.e4t_gemm_plan(_ct.byref(d), _ct.byref(pl))
        y, yr = hip.gemm(at, bt, **args), emu.gemm(at, bt, **args)
        tol = TOLF * 50 if kw.get("f32") else TOL1
        r = pl.tail_rows
        out.append((f"gemm {Mt}x{Nt}x{Kt} tail{r} t{pl.tile} {'gelu ' if kw.get('gelu') else ''}{'f32' if kw.get('f32') else 'bf16'}", rel(y, yr), tol))
        if r > 0:      # (r == 0: the planner ran the shape without a tail stage — e.g. M = 2056, whose 2048 main rows want split-K)
            out.append((f"gemm {Mt}x{Nt}x{Kt} tail{r}: the tail rows alone", rel(y[Mt - r:], yr[Mt - r:]), tol))
            out.append((f"gemm {Mt}x{Nt}x{Kt} tail{r}: the last tile rows in front of the tail", rel(y[Mt - r - 64:Mt - r], yr[Mt - r - 64:Mt - r]), tol))
    # fp32 C + fp32 residual (the CLIP-ViT's fp32 residual stream): the line-wide direct-store epilogue, every kernel family
    gv = gen(33, dev)
    for (Mv, Nv, Kv, t) in [(4112, 1280, 1280, 0), (4112, 1280, 5120, 0), (1000, 1280, 320, 160), (700, 512, 256, 512), (513, 200, 64, 64), (2048, 256, 128, 1128), (900, 384, 96, 5256),
                             (1000, 640, 320, 2320)]:
        if not _product_tile(hip, t):
            continue
        av, bw = rnd(gv, Mv, Kv, dev=dev), rnd(gv, Nv, Kv, scale=Kv ** -0.5, dev=dev)
        r32, bi = rnd(gv, Mv, Nv, dtype=f32, dev=dev), rnd(gv, Nv, dtype=f32, dev=dev)
        out.append((f"gemm {Mv}x{Nv}x{Kv} t{t} fp32 out + fp32 residual", rel(hip.gemm(av, bw, bias=bi, residual=r32, out_dtype=f32, tile=t),
                                                                            emu.gemm(av, bw, bias=bi, residual=r32, out_dtype=f32)), TOLF * 50))
    # the GENERAL (GELU / row-lookup) epilogue of the 256 x 320 ping-pong tile (GEMM only) and its row panels (e4t_gemm_desc.panel_*):
    # ragged M, several K depths, bf16 and fp32 output; panels leave the rows between them (the class-token rows) untouched
    gp = gen(34, dev)
    for (Mg, Ng, Kg) in [(1000, 640, 320), (4096, 320, 1280), (513, 960, 64)]:
        ag, bg, big = rnd(gp, Mg, Kg, dev=dev), rnd(gp, Ng, Kg, scale=Kg ** -0.5, dev=dev), rnd(gp, Ng, dtype=f32, dev=dev)
        out.append((f"gemm {Mg}x{Ng}x{Kg} t2320 gelu", rel(hip.gemm(ag, bg, bias=big, gelu=True, tile=2320), emu.gemm(ag, bg, bias=big, gelu=True)), TOL1))
        Mr = (Mg // 97) * 97                      # rows_per_batch = 97 (not a multiple of 32): the per-row row-bias lookup of the GENERAL epilogue
        rbg = rnd(gp, Mr // 97, Ng, dtype=f32, dev=dev)
        out.append((f"gemm {Mr}x{Ng}x{Kg} t2320 row bias, rows_per_batch 97", rel(hip.gemm(ag[:Mr], bg, rowbias=rbg, rows_per_batch=97, tile=2320),
                                                                                 emu.gemm(ag[:Mr], bg, rowbias=rbg, rows_per_batch=97)), TOL1))
    for (Bp, Tp, Np, Kp, pr, dt) in [(3, 257, 640, 128, 256, bf16), (16, 257, 5120, 1280, 256, bf16), (2, 520, 320, 64, 512, f32), (5, 257, 960, 192, 256, bf16)]:
        ap, bp, bip = rnd(gp, Bp * Tp, Kp, dev=dev), rnd(gp, Np, Kp, scale=Kp ** -0.5, dev=dev), rnd(gp, Np, dtype=f32, dev=dev)
        resid = rnd(gp, Bp * Tp, Np, dtype=dt, dev=dev) if dt == f32 else None
        yh = torch.full((Bp * Tp, Np), 7.0, dtype=dt, device=dev)
        ye = yh.clone()
        pan = (pr, Tp, Tp - pr, Bp)
        hip.gemm(ap, bp, bias=bip, gelu=(dt == bf16), residual=resid, out=yh, panels=pan)
        emu.gemm(ap, bp, bias=bip, gelu=(dt == bf16), residual=resid, out=ye, panels=pan)
        out.append((f"gemm row panels B{Bp} T{Tp} N{Np} K{Kp} ({pr} rows, offset {Tp - pr}) {'bf16 gelu' if dt == bf16 else 'fp32 out + residual'}", rel(yh, ye), TOL1 if dt == bf16 else TOLF * 50))
        skipped = torch.arange(Bp * Tp, device=dev).remainder(Tp) < Tp - pr
        out.append((f"gemm row panels B{Bp} T{Tp}: rows outside the panels untouched", float((yh[skipped] != 7.0).sum()), 0.0))
    rb = rnd(g, M // 96, N, dtype=f32, dev=dev)
    out.append(("gemm rowbias", rel(hip.gemm(a1, b[:, :K1].contiguous(), rowbias=rb, rows_per_batch=96),
                                    emu.gemm(a1, b[:, :K1].contiguous(), rowbias=rb, rows_per_batch=96)), TOL1))
    for t in [t for t in (1128, 1160) if _product_tile(hip, t)]:
        out.append((f"gemm rowbias t{t}", rel(hip.gemm(a1, b[:, :K1].contiguous(), rowbias=rb, rows_per_batch=96, tile=t),
                                              emu.gemm(a1, b[:, :K1].contiguous(), rowbias=rb, rows_per_batch=96)), TOL1))
        out.append((f"gemm strided A view t{t}", rel(hip.gemm(a1[:, 64:], b[:, :64].contiguous(), tile=t), emu.gemm(a1[:, 64:], b[:, :64].contiguous())), TOL1))
    # strided views (column slices of a wider buffer)
    wide = rnd(g, M, 3 * K1, dev=dev)
    out.append(("gemm strided A view", rel(hip.gemm(wide[:, K1:2 * K1], b[:, :K1].contiguous()), emu.gemm(wide[:, K1:2 * K1], b[:, :K1].contiguous())), TOL1))
    # batched + reduce-batch (E4T head shape, small)
    nb, M, N, K = 9, 16, 128, 128
    A, Bw = rnd(g, nb, M, K, dev=dev), rnd(g, nb, N, K, scale=0.09, dev=dev)
    bias = rnd(g, nb, N, dtype=f32, dev=dev)
    out.append(("gemm batched", rel(hip.gemm(A, Bw, bias=bias), emu.gemm(A, Bw, bias=bias)), TOL1))
    out.append(("gemm batched reduce (mean over slots)", rel(hip.gemm(A, Bw, reduce_batch=True, alpha=1.0 / nb, out_dtype=f32),
                                                            emu.gemm(A, Bw, reduce_batch=True, alpha=1.0 / nb, out_dtype=f32)), TOLF * 50))
    Ab = A[0].unsqueeze(0).expand(nb, M, K)   # broadcast operand (stride 0)
    out.append(("gemm batched broadcast A", rel(hip.gemm(Ab, Bw), emu.gemm(Ab, Bw)), TOL1))
    return out


def check_conv(hip, emu, dev):
    out = []
    cases = [  # B, Hin, Win, Cin, Cout, mode, Hout, Wout, tile, splitk
        (2, 16, 16, 64, 64, CONV_S1, 16, 16, 0, 0), (2, 8, 8, 128, 192, CONV_S1, 8, 8, 64, 3),
        (3, 16, 16, 64, 128, CONV_S2, 8, 8, 0, 0), (2, 9, 9, 64, 64, CONV_S2, 5, 5, 0, 0),
        (2, 8, 8, 64, 64, CONV_UP2, 16, 16, 0, 0), (2, 8, 8, 128, 64, CONV_S2T, 16, 16, 0, 0),
        (2, 5, 5, 64, 64, CONV_S2T, 9, 9, 0, 0), (4, 32, 32, 320, 320, CONV_S1, 32, 32, 128, 1),
        (2, 16, 16, 64, 4, CONV_S1, 16, 16, 0, 0), (3, 24, 24, 64, 192, CONV_S1, 24, 24, 256, 1), (3, 24, 24, 64, 320, CONV_S1, 24, 24, 160, 1), (2, 16, 16, 128, 128, 5, 8, 8, 0, 0), (1, 64, 64, 128, 128, 5, 32, 32, 0, 0),
        (3, 24, 24, 64, 320, CONV_S1, 24, 24, 512, 1), (2, 32, 32, 128, 256, CONV_S1, 32, 32, 512, 1), (2, 16, 16, 256, 512, CONV_S1, 16, 16, 512, 2),
        (3, 16, 16, 64, 128, CONV_S2, 8, 8, 512, 1), (2, 8, 8, 64, 64, CONV_UP2, 16, 16, 512, 1), (2, 8, 8, 128, 64, CONV_S2T, 16, 16, 512, 1),
        (1, 64, 64, 128, 128, 5, 32, 32, 512, 1),
        (2, 32, 32, 128, 128, CONV_S1, 32, 32, 640, 1), (1, 48, 40, 64, 128, CONV_S1, 48, 40, 640, 1), (3, 16, 16, 128, 384, CONV_S1, 16, 16, 640, 2),
        (1, 64, 64, 128, 128, 5, 32, 32, 640, 1), (2, 16, 16, 128, 128, CONV_UP2, 32, 32, 640, 1), (2, 32, 32, 128, 128, CONV_S2, 16, 16, 640, 1),
        # persistent streaming tiles: every gather mode; B16 64x64 = 256 row panels x 2 column tiles = two units per workgroup
        (4, 32, 32, 320, 320, CONV_S1, 32, 32, 1160, 1), (2, 64, 64, 128, 128, CONV_S1, 64, 64, 1128, 1), (16, 64, 64, 64, 320, CONV_S1, 64, 64, 1160, 1),
        (3, 24, 24, 64, 192, CONV_S1, 24, 24, 1128, 1), (3, 16, 16, 64, 128, CONV_S2, 8, 8, 1128, 1), (2, 8, 8, 64, 160, CONV_UP2, 16, 16, 1160, 1),
        (2, 8, 8, 128, 64, CONV_S2T, 16, 16, 1128, 1), (1, 64, 64, 128, 128, 5, 32, 32, 1128, 1), (18, 64, 64, 128, 128, CONV_S1, 64, 64, 1128, 1),
        (2, 48, 40, 64, 320, CONV_S1, 48, 40, 5256, 1),
        (2, 20, 12, 192, 128, CONV_S1, 20, 12, 5256, 1), (2, 20, 12, 192, 128, CONV_S1, 20, 12, 5128, 2), (1, 7, 9, 192, 64, CONV_S1, 7, 9, 5064, 1),   # channel-chunk-major K order with 32-wide chunks
        (4, 32, 32, 320, 320, CONV_S1, 32, 32, 2320, 1), (2, 16, 16, 256, 640, CONV_S1, 16, 16, 2320, 2), (2, 8, 8, 128, 320, CONV_S2T, 16, 16, 2320, 1),
        (3, 16, 16, 64, 320, CONV_S2, 8, 8, 2320, 1), (2, 8, 8, 64, 320, CONV_UP2, 16, 16, 2320, 1),
        (16, 64, 64, 64, 320, CONV_S1, 64, 64, 2320, 1),
        # round 6: conv_strip_kernel (stride 1, W % 256 == 0, the 256 x 128 x 32 tile): image borders in both directions, several 256-pixel
        # segments per row, one-row images, Cout beyond one column tile, batch crossing
        (2, 6, 256, 64, 128, CONV_S1, 6, 256, 5256, 1), (1, 3, 768, 192, 256, CONV_S1, 3, 768, 5256, 1), (3, 1, 256, 64, 128, CONV_S1, 1, 256, 5256, 1),
        (2, 5, 512, 128, 128, CONV_S1, 5, 512, 0, 0),
        # ... and gemm_pps_kernel (the ping-pong kernel on half-strips): whole-row tiles W = 16 / 64, row segments W = 256 / 512, split-K on kernel-row boundaries
        (2, 64, 64, 64, 256, CONV_S1, 64, 64, 512, 1), (1, 4, 256, 128, 320, CONV_S1, 4, 256, 512, 1), (1, 2, 512, 64, 256, CONV_S1, 2, 512, 512, 1),
        (2, 16, 16, 192, 256, CONV_S1, 16, 16, 512, 3), (3, 128, 128, 64, 128, CONV_S1, 128, 128, 512, 1),
    ]
    for i, (B, Hin, Win, Cin, Cout, mode, Hout, Wout, tile, sk) in enumerate(cases):
        if not _product_tile(hip, tile):
            continue
        g = gen(50 + i, dev)
        x = rnd(g, B * Hin * Win, Cin, dev=dev)
        w = rnd(g, Cout, 9 * Cin, scale=(9 * Cin) ** -0.5, dev=dev)
        bias = rnd(g, Cout, dtype=f32, dev=dev)
        rb = rnd(g, B, Cout, dtype=f32, dev=dev)
        res = rnd(g, B * Hout * Wout, Cout, dev=dev)
        y = hip.conv3x3(x, w, B, Hin, Win, Hout, Wout, mode, bias=bias, rowbias=rb, residual=res, tile=tile, splitk=sk)
        yr = emu.conv3x3(x, w, B, Hin, Win, Hout, Wout, mode, bias=bias, rowbias=rb, residual=res)
        out.append((f"conv mode{mode} B{B} {Hin}x{Win} {Cin}->{Cout} t{tile} s{sk}", rel(y, yr), TOL1))
    # weight relayout + dgrad identity: conv dgrad == autograd of conv
    g = gen(70, dev)
    O, I = 96, 64
    w4 = rnd(g, O, I, 3, 3, scale=0.05, dtype=f32, dev=dev)
    wf, wd = hip.conv_weight_prepare(w4)
    wf2, wd2 = emu.conv_weight_prepare(w4)
    out.append(("conv_weight_prepare fwd layout", rel(wf, wf2), 0.0))
    out.append(("conv_weight_prepare dgrad layout", rel(wd, wd2), 0.0))
    return out


def check_attention(hip, emu, dev):
    out = []
    cases = [  # B, H, T, S, DH (, causal)
        (2, 2, 64, 64, 32), (2, 3, 200, 200, 40), (1, 2, 128, 77, 40), (2, 2, 96, 77, 80), (1, 2, 64, 64, 160),
        (1, 2, 257, 257, 80), (2, 2, 130, 33, 64), (1, 8, 1024, 1024, 40),
        (3, 5, 300, 300, 40), (2, 8, 4096, 4096, 40),        # 45 workgroups (XCD re-deal with a remainder); the step's own 64 x 64 self-attention
        (1, 2, 300, 2100, 40), (1, 1, 2050, 2050, 64),       # S >= 2048: the dK/dV kernel's three-workgroups-per-CU instantiation, ragged tiles
        (3, 12, 77, 77, 64, True), (2, 3, 200, 200, 40, True), (1, 2, 128, 128, 80, True),      # causal: CLIP text encoder
        # few key blocks, long query range: the dK/dV kernel cuts T into chunks + fp32 partial reduce (round 4) — the step's own
        # cross-attention shape at a smaller batch, a ragged T (3 chunks of 384 / 384 / 232), dh 80 / 64, a short self-attention
        (2, 8, 4096, 77, 40), (1, 2, 1000, 77, 40), (2, 2, 1024, 77, 80), (1, 2, 600, 33, 64), (1, 4, 1024, 1024, 40),
        # round 6: the 64-queries-per-wave forward (dh 40, S >= 512): last tile of 8 keys (its second sub-tile fully masked), of 33
        # keys, ragged T inside a 256-query block, exactly one tile pair
        (1, 2, 520, 520, 40), (2, 3, 700, 545, 40), (1, 1, 40, 512, 40), (2, 2, 256, 640, 40),
        # ... and the 64-keys-per-wave dK/dV kernel + 64-queries-per-wave dQ kernel (dh 40, enough key blocks that the query range is
        # not split): ragged T (last tile of 24 / 12 queries) and ragged S (last workgroup with 208 / 42 keys)
        (4, 8, 600, 2000, 40), (2, 16, 1100, 2090, 40),
    ]
    for i, case in enumerate(cases):
        (B, H, T, S, DH), causal = case[:5], (len(case) > 5 and case[5])
        g = gen(90 + i, dev)
        d = H * DH
        # q, k, v as column slices of fused buffers (self-attn layout) when T == S, separate otherwise
        if T == S:
            qkv = rnd(g, B * T, 3 * d, dev=dev)
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            dqkv_h = torch.zeros_like(qkv); dqkv_e = torch.zeros_like(qkv)
            gh = (dqkv_h[:, :d], dqkv_h[:, d:2 * d], dqkv_h[:, 2 * d:])
            ge = (dqkv_e[:, :d], dqkv_e[:, d:2 * d], dqkv_e[:, 2 * d:])
        else:
            q = rnd(g, B * T, d, dev=dev)
            kv = rnd(g, B * S, 2 * d, dev=dev)
            k, v = kv[:, :d], kv[:, d:]
            dq_h, dq_e = torch.zeros_like(q), torch.zeros_like(q)
            dkv_h, dkv_e = torch.zeros_like(kv), torch.zeros_like(kv)
            gh = (dq_h, dkv_h[:, :d], dkv_h[:, d:]); ge = (dq_e, dkv_e[:, :d], dkv_e[:, d:])
        scale = DH ** -0.5
        o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, scale, causal=causal)
        o_r, lse_r = emu.attention_fwd(q, k, v, B, H, T, S, DH, scale, causal=causal)
        tag = f"attn B{B} H{H} T{T} S{S} dh{DH}" + (" causal" if causal else "")
        out.append((tag + " fwd O", rel(o, o_r), TOL2))
        out.append((tag + " fwd LSE", rel(lse, lse_r), 1e-3))
        do = rnd(g, B * T, d, dev=dev)
        hip.attention_bwd(q, k, v, o, do, lse, gh[0], gh[1], gh[2], B, H, T, S, DH, scale, causal=causal)
        emu.attention_bwd(q, k, v, o_r, do, lse_r, ge[0], ge[1], ge[2], B, H, T, S, DH, scale, causal=causal)
        for nm, a, b in zip(("dQ", "dK", "dV"), gh, ge):
            out.append((tag + " bwd " + nm, rel(a, b), TOL2))
    # The step's own dh-40 self-attention at the bench batch (B16 H8 T = S = 4096: 2048-4096 workgroups = several rounds of two / three
    # resident workgroups per CU — the regime in which round 6's LDS-DMA kernels first showed a race: pad columns written into rows
    # whose DMA piece another wave still had in flight; every smaller case above ran one round and passed): outputs must be BITWISE equal
    # run to run, and equal to a second evaluation in four batch chunks of B = 4 (other grid, other residency) to rounding.
    g = gen(119, dev)
    B, H, T, S, DH = 16, 8, 4096, 4096, 40
    d = H * DH
    qkv = rnd(g, B * T, 3 * d, scale=0.7, dev=dev)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    do = rnd(g, B * T, d, dev=dev)
    runs = []
    for rep in range(3):
        o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5)
        gq = torch.zeros_like(qkv)
        hip.attention_bwd(q, k, v, o, do, lse, gq[:, :d], gq[:, d:2 * d], gq[:, 2 * d:], B, H, T, S, DH, DH ** -0.5)
        runs.append((o.clone(), lse.clone(), gq))
    for rep in (1, 2):
        out.append((f"attn B16 H8 T4096 dh40 run {rep} == run 0 bitwise (O, LSE, dQ|dK|dV)",
                    float(sum((a != b).sum() for a, b in zip(runs[rep], runs[0]))), 0.0))
    o4, g4 = torch.empty_like(runs[0][0]), torch.zeros_like(qkv)
    for c in range(4):
        sl = slice(c * 4 * T, (c + 1) * 4 * T)
        oc, lc = hip.attention_fwd(q[sl], k[sl], v[sl], 4, H, T, S, DH, DH ** -0.5)
        o4[sl] = oc
        hip.attention_bwd(q[sl], k[sl], v[sl], oc, do[sl], lc, g4[sl, :d], g4[sl, d:2 * d], g4[sl, 2 * d:], 4, H, T, S, DH, DH ** -0.5)
    out.append(("attn B16 == 4 x B4 fwd O", rel(runs[0][0], o4), 1e-5))
    for nm, sl_ in (("dQ", slice(0, d)), ("dK", slice(d, 2 * d)), ("dV", slice(2 * d, 3 * d))):
        out.append((f"attn B16 == 4 x B4 bwd {nm}", rel(runs[0][2][:, sl_], g4[:, sl_]), 1e-5))
    del runs, o4, g4, qkv, do
    # peaked scores: one key dominates (exercises the online-softmax rescale with large max jumps)
    g = gen(120, dev)
    B, H, T, S, DH = 1, 1, 64, 160, 64
    q, k, v = rnd(g, T, DH, dev=dev), rnd(g, S, DH, dev=dev), rnd(g, S, DH, dev=dev)
    k[130] = (q[5].float() * 6).to(bf16)
    o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, 1.0)
    o_r, lse_r = emu.attention_fwd(q, k, v, B, H, T, S, DH, 1.0)
    out.append(("attn peaked-score fwd", rel(o, o_r), TOL2))
    # the same on the dh-40 long-key kernel: keys that beat the running max by far in the first, a middle and the last tile (its slow
    # path: P recomputed after the row max is raised), a row whose best key comes first, and a huge negative score
    g = gen(121, dev)
    B, H, T, S, DH = 1, 1, 96, 840, 40
    q, k, v = rnd(g, T, DH, dev=dev), rnd(g, S, DH, dev=dev), rnd(g, S, DH, dev=dev)
    for key, row, mul in ((3, 7, 8.0), (333, 40, 12.0), (834, 5, 20.0), (0, 70, 30.0), (500, 9, -30.0)):
        k[key] = (q[row].float() * mul).to(bf16)
    o, lse = hip.attention_fwd(q, k, v, B, H, T, S, DH, 1.0)
    o_r, lse_r = emu.attention_fwd(q, k, v, B, H, T, S, DH, 1.0)
    out.append(("attn peaked-score dh40 long-key fwd O", rel(o, o_r), TOL2))
    out.append(("attn peaked-score dh40 long-key fwd LSE", rel(lse, lse_r), 1e-3))
    return out


def check_norms(hip, emu, dev):
    out = []
    # the last six take the one-launch slab kernels (HW x C/G <= 10240, C/G % 4 == 0, no group straddling the concat): 3 / 5 / 10
    # quads per thread, two sources, ragged item counts, with and without SiLU / shortcut gradients
    for i, (B, HW, C1, C2, G, silu) in enumerate([(2, 64, 64, 0, 32, True), (3, 256, 320, 0, 32, True), (2, 100, 320, 640, 32, True),
                                                  (2, 64, 1280, 1280, 32, False), (16, 4096, 320, 0, 32, True),
                                                  (2, 64, 1280, 0, 32, True), (3, 256, 1280, 0, 32, True), (2, 64, 1280, 1280, 32, True),
                                                  (2, 60, 640, 0, 32, True), (1, 256, 640, 640, 32, False), (2, 64, 640, 640, 32, True)]):
        g = gen(140 + i, dev)
        x1 = rnd(g, B * HW, C1, dev=dev) + 0.3
        x2 = rnd(g, B * HW, C2, scale=2.0, dev=dev) if C2 else None
        Cn = C1 + C2
        gamma, beta = rnd(g, Cn, dtype=f32, dev=dev) * 0.3 + 1.0, rnd(g, Cn, dtype=f32, dev=dev) * 0.2
        y, st = hip.groupnorm_fwd(x1, x2, gamma, beta, B, HW, G, 1e-5, silu)
        yr, str_ = emu.groupnorm_fwd(x1, x2, gamma, beta, B, HW, G, 1e-5, silu)
        tag = f"groupnorm B{B} HW{HW} C{C1}+{C2} silu{int(silu)}"
        out.append((tag + " fwd", rel(y, yr), TOL1))
        out.append((tag + " stats", rel(st, str_), 1e-4))
        if HW % 32 == 0:       # the same GroupNorm fed with the column statistics a producing GEMM would have left behind
            def with_cs(t):
                blk = t.float().reshape(t.shape[0] // 32, 32, t.shape[1])
                t._e4t_colstats = torch.stack([blk.sum(1), (blk * blk).sum(1)], dim=-1).contiguous()
                return t
            x1c, x2c = with_cs(x1.clone()), (with_cs(x2.clone()) if x2 is not None else None)
            y3, st3 = hip.groupnorm_fwd(x1c, x2c, gamma, beta, B, HW, G, 1e-5, silu)
            out.append((tag + " via column statistics: y", rel(y3, y), 2e-3))
            out.append((tag + " via column statistics: stats", rel(st3, st), 1e-4))
        y2, st2 = hip.groupnorm_fwd_unfused(x1, x2, gamma, beta, B, HW, G, 1e-5, silu)
        cpg = Cn // G
        slab = cpg % 4 == 0 and HW * cpg <= 10240 and (C2 == 0 or C1 % cpg == 0)
        if slab:     # one-launch kernel: same arithmetic, different summation tree
            out.append((tag + " slab kernel ~ stats/finalize/apply entry points: y", rel(y, y2), 2e-3))
            out.append((tag + " slab kernel ~ stats/finalize/apply entry points: stats", rel(st, st2), 1e-5))
        else:
            out.append((tag + " fused == stats/finalize/apply entry points", float((y2 != y).sum() + (st2 != st).sum()), 0.0))
        dy = rnd(g, B * HW, Cn, dev=dev)
        add = rnd(g, B * HW, C1, dev=dev) if i % 2 == 0 else None            # gradient through the block's shortcut
        add2 = rnd(g, B * HW, C2, dev=dev) if (C2 and i != 3) else None
        dx1, dx2, dga, dbe = hip.groupnorm_bwd(x1, x2, dy, str_, gamma, beta, add, B, HW, G, silu, want_param_grads=True, add2=add2)
        ex1, ex2, ega, ebe = emu.groupnorm_bwd(x1, x2, dy, str_, gamma, beta, add, B, HW, G, silu, want_param_grads=True, add2=add2)
        out.append((tag + " bwd dx1", rel(dx1, ex1), TOL1))
        if C2:
            out.append((tag + " bwd dx2", rel(dx2, ex2), TOL1))
        out.append((tag + " bwd dgamma", rel(dga, ega), 1e-3))
        out.append((tag + " bwd dbeta", rel(dbe, ebe), 1e-3))
        # frozen gamma / beta (pre-training): no parameter-gradient partials -> slab kernel where eligible
        fx1, fx2, _, _ = hip.groupnorm_bwd(x1, x2, dy, str_, gamma, beta, add, B, HW, G, silu, want_param_grads=False, add2=add2)
        out.append((tag + " bwd dx1 (no param grads)", rel(fx1, ex1), TOL1))
        if C2:
            out.append((tag + " bwd dx2 (no param grads)", rel(fx2, ex2), TOL1))
    # the last three take the several-rows-per-wave forward (4 / 2 / 2 rows for 1 / 2 / 3 chunks per lane) with a ragged last wave
    for i, (M, D) in enumerate([(37, 64), (1000, 320), (4112, 1280), (300, 768), (128, 1024), (8195, 320), (4101, 640), (4111, 1280)]):
        g = gen(160 + i, dev)
        x = rnd(g, M, D, dev=dev) + 0.5
        gamma, beta = rnd(g, D, dtype=f32, dev=dev) * 0.3 + 1.0, rnd(g, D, dtype=f32, dev=dev) * 0.2
        y, st = hip.layernorm_fwd(x, gamma, beta, 1e-5)
        yr, str_ = emu.layernorm_fwd(x, gamma, beta, 1e-5)
        out.append((f"layernorm {M}x{D} fwd", rel(y, yr), TOL1))
        out.append((f"layernorm {M}x{D} stats", rel(st, str_), 1e-4))
        dy = rnd(g, M, D, dev=dev)
        dx, dga, dbe = hip.layernorm_bwd(x, dy, gamma, str_, want_param_grads=True)
        ex, ega, ebe = emu.layernorm_bwd(x, dy, gamma, str_, want_param_grads=True)
        out.append((f"layernorm {M}x{D} bwd dx", rel(dx, ex), TOL1))
        out.append((f"layernorm {M}x{D} bwd dgamma", rel(dga, ega), 1e-3))
        out.append((f"layernorm {M}x{D} bwd dbeta", rel(dbe, ebe), 1e-3))
        x32 = (rnd(g, M, D, dtype=f32, dev=dev) + 0.5) * 3.0            # fp32 rows (ViT residual stream) -> bf16 y
        y32, st32 = hip.layernorm_fwd(x32, gamma, beta, 1e-5)
        yr32, str32 = emu.layernorm_fwd(x32, gamma, beta, 1e-5)
        out.append((f"layernorm {M}x{D} fwd, fp32 input", rel(y32, yr32), TOL1))
        out.append((f"layernorm {M}x{D} stats, fp32 input", rel(st32, str32), 1e-4))
        skip = rnd(g, M, D, dev=dev)
        out.append((f"layernorm {M}x{D} bwd dx + residual grad", rel(hip.layernorm_bwd(x, dy, gamma, str_, add=skip)[0],
                                                                     emu.layernorm_bwd(x, dy, gamma, str_, add=skip)[0]), TOL1))
    return out


def check_streaming(hip, emu, dev):
    out = []
    g = gen(179, dev)
    for M, Cn in [(37, 64), (5000, 320), (65536, 320), (1232, 2560), (300, 8)]:
        xs = rnd(g, M, Cn + 8, dev=dev)[:, :Cn]                      # strided view
        out.append((f"colsum {M}x{Cn}", rel(hip.colsum(xs), emu.colsum(xs)), 1e-3))
    for B, Cn, HW, cfg, nhwc, noise in [(2, 4, 4096, True, True, False), (3, 4, 100, True, False, True), (1, 4, 9216, False, True, True), (2, 4, 37, False, False, False)]:
        pred = torch.randn(((2 if cfg else 1) * B * Cn * HW,), generator=g, device=dev)
        x = torch.randn((B, Cn, HW), generator=g, device=dev)
        nz = torch.randn((B, Cn, HW), generator=g, device=dev) if noise else None
        coef = torch.tensor([7.5, 1.0123, -0.0456, 0.3], dtype=f32, device=dev)
        got = hip.guided_step(pred, x, coef, noise=nz, cfg=cfg, pred_nhwc=nhwc)
        out.append((f"guided_step B{B} HW{HW} cfg={cfg} nhwc={nhwc}", rel(got, emu.guided_step(pred, x, coef, noise=nz, cfg=cfg, pred_nhwc=nhwc)), 1e-6))
    x2 = x.clone()
    hip.guided_step(pred, x2, coef, cfg=False, pred_nhwc=False, out=x2)          # in place on the sample, as the pipeline uses it
    out.append(("guided_step in place", rel(x2, emu.guided_step(pred, x, coef, cfg=False)), 1e-6))
    acc = torch.ones(320, dtype=f32, device=dev)
    xs = rnd(g, 700, 320, dev=dev)
    hip.colsum(xs, out=acc, accumulate=True)
    out.append(("colsum accumulate", rel(acc, 1.0 + emu.colsum(xs)), 1e-3))
    g = gen(180, dev)
    u = rnd(g, 300, 2 * 640, dev=dev)
    dh = rnd(g, 300, 640, dev=dev)
    out.append(("geglu fwd", rel(hip.geglu_fwd(u), emu.geglu_fwd(u)), TOL1))
    out.append(("geglu bwd", rel(hip.geglu_bwd(u, dh), emu.geglu_bwd(u, dh)), TOL1))
    x, dy = rnd(g, 64, 1280, dev=dev), rnd(g, 64, 1280, dev=dev)
    for op in range(8):
        out.append((f"unary op{op}", rel(hip.unary(x, op, dy if op & 1 else None), emu.unary(x, op, dy if op & 1 else None)), TOL1))
    out.append(("add", rel(hip.add(x, dy), emu.add(x, dy)), TOL1))
    t = rnd(g, 130, 72, dev=dev)
    out.append(("transpose", rel(hip.transpose(t), emu.transpose(t)), 0.0))
    out.append(("transpose padded + strided in", rel(hip.transpose(u[:, 8:80], pad_to=320), emu.transpose(u[:, 8:80], pad_to=320)), 0.0))
    xx = rnd(g, 2 * 16 * 16, 64, dev=dev)
    out.append(("sumpool2", rel(hip.sumpool2(xx, 2, 8, 8), emu.sumpool2(xx, 2, 8, 8)), TOL1))
    o1 = torch.zeros(2, 200, dtype=f32, device=dev); o2 = torch.zeros_like(o1)
    hip.spatial_mean(xx, 2, 256, o1, 100); emu.spatial_mean(xx, 2, 256, o2, 100)
    out.append(("spatial_mean", rel(o1, o2), TOLF))
    gg = rnd(g, 2, 200, dtype=f32, dev=dev)
    out.append(("spatial_mean_bwd", rel(hip.spatial_mean_bwd(gg, xx, 2, 256, 64, 100), emu.spatial_mean_bwd(gg, xx, 2, 256, 64, 100)), TOL1))
    ts = torch.tensor([0, 1, 17, 500, 999], device=dev)
    out.append(("timestep_embedding", rel(hip.timestep_embedding(ts, 320), emu.timestep_embedding(ts, 320)), TOL1))
    px = torch.rand(2, 3, 512, 512, generator=g, device=dev) * 2 - 1
    out.append(("clip_preprocess 512->224 p14", rel(hip.clip_preprocess(px, 224, 14, 640), emu.clip_preprocess(px, 224, 14, 640)), TOL1))
    px = torch.rand(2, 3, 64, 64, generator=g, device=dev) * 2 - 1
    out.append(("clip_preprocess 64->28 p14", rel(hip.clip_preprocess(px, 28, 14, 640), emu.clip_preprocess(px, 28, 14, 640)), TOL1))
    n = 100003
    p = rnd(g, n, dtype=f32, dev=dev); gr = rnd(g, n, dtype=f32, dev=dev); m = rnd(g, n, dtype=f32, dev=dev) * 0.1; v = rnd(g, n, dtype=f32, dev=dev).abs() * 0.01
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    hip.adamw(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 3, 0.5)
    emu.adamw(p2, gr, m2, v2, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 3, 0.5)
    out += [("adamw p", rel(p, p2), TOLF), ("adamw m", rel(m, m2), TOLF), ("adamw v", rel(v, v2), TOLF)]
    out.append(("sumsq", rel(hip.sumsq(gr), emu.sumsq(gr)), 1e-4))
    # AdamW of a weight stack under the never-materialised rank-K gradient G^T Z_i (round 6): K = 16 (one k chunk), 40 (three, ragged),
    # rows not a multiple of the row block, columns that are / are not a multiple of the 64-lane quad count, device-resident scalars
    for i, (n, rows, cols, K) in enumerate(((5, 64, 128, 16), (3, 100, 320, 40), (2, 1280, 1280, 16), (4, 24, 72, 3))):
        gg = gen(470 + i, dev)
        P = rnd(gg, n, rows, cols, dtype=f32, dev=dev)
        M = rnd(gg, n, rows, cols, dtype=f32, dev=dev) * 0.1
        Vv = rnd(gg, n, rows, cols, dtype=f32, dev=dev).abs() * 0.01
        Gf, Zf = rnd(gg, K, rows, dev=dev), rnd(gg, K, n * cols, dev=dev)
        P2, M2, V2 = P.clone(), M.clone(), Vv.clone()
        hyper = torch.tensor([2e-3, 1 - 0.9 ** 4, (1 - 0.999 ** 4) ** 0.5, 0.25], dtype=f32, device=dev) if i == 1 else None
        hip.adamw_rank(P, M, Vv, Gf, Zf, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 3, 0.5, hyper=hyper)
        emu.adamw_rank(P2, M2, V2, Gf, Zf, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 3, 0.5, hyper=hyper)
        tag = f"adamw_rank n{n} {rows}x{cols} K{K}" + (" hyper" if hyper is not None else "")
        out += [(tag + " p", rel(P, P2), TOLF), (tag + " m", rel(M, M2), TOLF), (tag + " v", rel(Vv, V2), TOLF)]
    for mode, (Bn, Hin, Hout) in ((1, (2, 12, 12)), (2, (3, 9, 5)), (3, (2, 6, 12))):
        xi = rnd(g, Bn * Hin * Hin, 72, dev=dev)
        out.append((f"im2col_T mode{mode}", rel(hip.im2col_T(xi, Bn, Hin, Hin, Hout, Hout, mode), emu.im2col_T(xi, Bn, Hin, Hin, Hout, Hout, mode)), 0.0))
        out.append((f"im2col mode{mode}", rel(hip.im2col(xi, Bn, Hin, Hin, Hout, Hout, mode), emu.im2col(xi, Bn, Hin, Hin, Hout, Hout, mode)), 0.0))
    sc = rnd(g, 3, 100, 4096, scale=2.0, dev=dev)
    out.append(("softmax_rows 4096", rel(hip.softmax_rows_(sc.clone()), emu.softmax_rows_(sc.clone())), TOL1))
    sc = rnd(g, 77, 64, scale=3.0, dev=dev)
    out.append(("softmax_rows 64", rel(hip.softmax_rows_(sc.clone()), emu.softmax_rows_(sc.clone())), TOL1))
    px = torch.rand(2, 3, 40, 24, generator=g, device=dev) * 2 - 1
    out.append(("im2col3_rgb", rel(hip.im2col3_rgb(px), emu.im2col3_rgb(px)), 0.0))
    big = rnd(g, 2 * 4096, 320, dev=dev)
    o1 = torch.zeros(2, 320, dtype=f32, device=dev); o2 = torch.zeros_like(o1)
    hip.spatial_mean(big, 2, 4096, o1, 0); emu.spatial_mean(big, 2, 4096, o2, 0)
    out.append(("spatial_mean 4096x320", rel(o1, o2), TOLF * 5))
    return out


def make_wo_entry(g, row, col, dev, ops_mod, with_gW=False, ld_pad=0):
    WOEntry = ops_mod.WOEntry
    s = lambda *sh, sc=1.0: rnd(g, *sh, scale=sc, dtype=f32, dev=dev)
    params = dict(v=torch.full((1,), 0.8, device=dev), w1=s(row, 1, sc=0.5), b1=s(row, sc=0.5), w2=s(col, 1, sc=0.5), b2=s(col, sc=0.5),
                  wc=s(row, row, sc=row ** -0.5), bc=s(row, sc=0.1), wr=s(col, col, sc=col ** -0.5), br=s(col, sc=0.1))
    grads = {"g_" + k: torch.zeros_like(v) for k, v in params.items()}
    W = s(col, row, sc=row ** -0.5)
    weff = torch.zeros(col, row + ld_pad, dtype=bf16, device=dev)
    weffT = torch.zeros(row, col + ld_pad, dtype=bf16, device=dev)
    dweff = s(col, row + ld_pad)
    return WOEntry(row=row, col=col, W=W, params=params, weff=weff, weffT=weffT, dweff=dweff, grads=grads,
                   g_W=torch.zeros_like(W) if with_gW else None)


def check_wo(hip, emu, dev, ops_mod):
    import copy
    out = []
    g = gen(200, dev)
    dims = [(64, 64), (96, 160), (320, 320), (768, 320), (1280, 1280)]
    ents = [make_wo_entry(g, r, c, dev, ops_mod, with_gW=(i == 1), ld_pad=(8 if i == 2 else 0)) for i, (r, c) in enumerate(dims)]
    ents2 = copy.deepcopy(ents)
    th, te = ops_mod.WOTable(ents), ops_mod.WOTable(ents2)
    hip.wo_forward(th); emu.wo_forward(te)
    for (r, c), a, b in zip(dims, ents, ents2):
        out.append((f"wo fwd W_eff {r}->{c}", rel(a.weff, b.weff), TOL1))
        out.append((f"wo fwd W_eff^T {r}->{c}", rel(a.weffT, b.weffT), TOL1))
    hip.wo_backward(th, False); emu.wo_backward(te, False)
    for (r, c), a, b in zip(dims, ents, ents2):
        for k in a.grads:
            out.append((f"wo bwd {k} {r}->{c}", rel(a.grads[k], b.grads[k]), 2e-4))
        if a.g_W is not None:
            out.append((f"wo bwd g_W {r}->{c}", rel(a.g_W, b.g_W), 2e-4))
    hip.wo_backward(th, True); emu.wo_backward(te, True)
    out.append(("wo bwd accumulate g_wc", rel(ents[2].grads["g_wc"], ents2[2].grads["g_wc"]), 2e-4))
    # plain weights (cast + transpose only)
    W = rnd(g, 200, 136, dtype=f32, dev=dev)
    e1 = ops_mod.WOEntry(row=136, col=200, W=W, weff=torch.zeros(200, 136, dtype=bf16, device=dev), weffT=torch.zeros(136, 200, dtype=bf16, device=dev))
    e2 = copy.deepcopy(e1)
    hip.weight_prepare(ops_mod.WOTable([e1])); emu.weight_prepare(ops_mod.WOTable([e2]))
    out.append(("weight_prepare cast", rel(e1.weff, e2.weff), 0.0))
    out.append(("weight_prepare transpose", rel(e1.weffT, e2.weffT), 0.0))
    return out


def check_image_prep(hip, emu, dev):
    """data path (N2): byte-exact against the numpy restatement of SmallestMaxSize(INTER_AREA)+crop+flip+normalise,
    every resize branch (untouched / integer box / 2x2 / general area / enlarging fixed point), ragged sizes in one batch"""
    import numpy as np
    import image_prep_oracle as ipo
    sys_rng = np.random.default_rng(7)
    res = []
    for S, dims in ((64, [(128, 192), (192, 192), (97, 131), (40, 55), (64, 80), (65, 64), (64, 64), (201, 77)]),
                    (512, [(1024, 1536), (1536, 1536), (700, 933), (300, 400), (512, 640), (513, 700), (2048, 2731), (1200, 512)])):
        samples = []
        for i, (H, W) in enumerate(dims):
            img = sys_rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            nh, nw = ipo.smallest_max_size_dims(H, W, S)
            y0, x0 = ipo.random_crop_origin(nh, nw, S, sys_rng.random(), sys_rng.random())
            samples.append(dict(image=img, plan=(nh, nw, y0, x0, i & 1)))
        offs, total = [], 0
        for smp in samples:
            offs.append(total)
            total += (smp["image"].size + 15) // 16 * 16
        pool = np.zeros(total, np.uint8)
        table = []
        for smp, off in zip(samples, offs):
            im = smp["image"]
            pool[off:off + im.size] = im.reshape(-1)
            table.append([off, im.shape[0], im.shape[1], *smp["plan"]])
        d_pool = torch.from_numpy(pool).to(dev)
        d_table = torch.tensor(table, dtype=torch.int64, device=dev)
        got = hip.image_prep(d_pool, d_table, len(samples), S).cpu()
        for i, smp in enumerate(samples):
            nh, nw, y0, x0, flip = smp["plan"]
            want = torch.from_numpy(ipo.image_prep(smp["image"], S, y0, x0, bool(flip)))
            H, W = smp["image"].shape[:2]
            res.append((f"image_prep S={S} {H}x{W}->{nh}x{nw} flip={flip}", float((got[i] - want).abs().max()), 0.0))
    return res


def check_gemm_races(hip, emu, dev):
    """The DMA GEMM kernels hand an LDS stage back to the loader right after the K-loop barrier.  Until round 2 that barrier
    did not wait for the fragment reads of the stage to complete (gemm.hip, loop_barrier): with several workgroups per CU a
    tile in a thousand came out wrong, not reproducibly.  The kernels are deterministic, so ANY difference between
    repeated launches — and between the 2 / 3 / 4-stage variants, which accumulate in the same order — is a race.
    Grids of several waves, 2-5 workgroups per CU, operands larger than one XCD's L2."""
    out = []
    g = gen(400, dev)
    for M, N, K in [(4096, 1280, 320), (8192, 1280, 512), (16384, 640, 640), (131072, 320, 320)]:
        a, w = rnd(g, M, K, dev=dev), rnd(g, N, K, dev=dev)
        want = emu.gemm(a, w)
        ref = hip.gemm(a, w, tile=128)
        out.append((f"race-check reference {M}x{N}x{K}", rel(ref, want), TOL1))
        for code in (64, 3064, 4064, 128, 3128, 4128, 160, 3160, 4160, 1128, 1160, 5256, 512, 2320):
            if not _product_tile(hip, code):
                continue
            if code % 1000 == 160 and N % 160:
                continue
            if code == 512 and N % 256:
                continue
            if code == 2320 and N % 320:
                continue
            differing = 0
            for _ in range(12):
                differing += int((hip.gemm(a, w, tile=code) != ref).sum() > 0)
            out.append((f"gemm {M}x{N}x{K} tile code {code}: launches (of 12) differing from the reference", float(differing), 0.0))
    x, w = rnd(g, 16 * 32 * 32, 640, dev=dev), rnd(g, 640, 9 * 640, dev=dev)
    ref = hip.conv3x3(x, w, 16, 32, 32, 32, 32, CONV_S1, tile=128)
    for code in (128, 3128, 160, 4160, 64, 3064, 1128, 1160, 2320, 512, 5256):
        if not _product_tile(hip, code):
            continue
        sk = 1 if code in (2320, 512) else 0          # (its automatic split-K would change the summation order, not a race)
        run = lambda: hip.conv3x3(x, w, 16, 32, 32, 32, 32, CONV_S1, tile=code, splitk=sk)
        # the persistent kernels walk K tap-major, every other DMA kernel channel-chunk-major (gemm_common.h, cm_step): a different
        # fp32 summation order, so their bitwise reference is their own first launch (checked against `ref` to tolerance)
        own = code in (1128, 1160, 5256)        # (5256: 32-wide channel chunks = another fp32 summation order than the 64-wide tiles)
        r = run() if own else ref
        if own:
            out.append((f"conv 32x32 640->640 tile code {code} vs the channel-major kernels", rel(r, ref), TOL2))
        differing = sum(int((run() != r).sum() > 0) for _ in range(6))
        out.append((f"conv 32x32 640->640 tile code {code}: launches (of 6) differing", float(differing), 0.0))
    # tail rows (gemm_tail: cross-wave reduction through LDS in wave order, behind the tile epilogue's staging): repeated launches of the
    # ViT's shapes on the 128 x 160 / 256 x 320 / 256 x 256 tiles must be bitwise reproducible
    for (Mt, Nt, Kt, gelu) in [(4112, 1280, 5120, False), (4112, 5120, 1280, True), (4112, 3840, 1280, False)]:
        at, bt = rnd(g, Mt, Kt, dev=dev), rnd(g, Nt, Kt, scale=Kt ** -0.5, dev=dev)
        ref = hip.gemm(at, bt, gelu=gelu)
        differing = sum(int((hip.gemm(at, bt, gelu=gelu) != ref).sum() > 0) for _ in range(12))
        out.append((f"gemm {Mt}x{Nt}x{Kt} with tail rows: launches (of 12) differing", float(differing), 0.0))
    dy, xx = rnd(g, 16384, 640, dev=dev), rnd(g, 16384, 1280, dev=dev)
    ref = hip.gemm_tn(dy, xx)
    differing = sum(int((hip.gemm_tn(dy, xx) != ref).sum() > 0) for _ in range(12))
    out.append(("gemm_tn 640x1280 K=16384: launches (of 12) differing", float(differing), 0.0))
    return out


def all_checks(hip, emu, dev, ops_mod):
    yield "gemm_races", lambda: check_gemm_races(hip, emu, dev)
    yield "probe", lambda: check_probe(hip, emu, dev)
    yield "gemm", lambda: check_gemm(hip, emu, dev)
    yield "conv", lambda: check_conv(hip, emu, dev)
    yield "attention", lambda: check_attention(hip, emu, dev)
    yield "norms", lambda: check_norms(hip, emu, dev)
    yield "streaming", lambda: check_streaming(hip, emu, dev)
    yield "wo", lambda: check_wo(hip, emu, dev, ops_mod)
    yield "image_prep", lambda: check_image_prep(hip, emu, dev)
