"""GPU: the small-M GEMMs of the step — the B = 1 shapes of BASELINE configs[4] (ViT
at 257 rows, the 12 x 12 / 24 x 24 UNet levels, the 77-token text encoder) and the small shapes of the B = 16 step — over tile, LDS stages
and split-K, with COLD operands (activations AND weights cycle through a pool larger than the 256 MB Infinity Cache, as in the step where
1.3 GB of ViT weights pass between two uses of one matrix)."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch
from e4t import ops
hip = ops.HipBackend()
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf16)
EXP = bool(hip.lib.e4t_build_flags() & 1)


def pool_time(make, run, nbytes, iters=24):
    """the launches are captured in a graph and replayed: at 10-20 us per kernel an eager loop measures the host (ctypes launch ~15 us)"""
    n = max(3, min(48, int(700e6 / max(nbytes, 1)) + 1))
    bufs = [make() for _ in range(n)]
    for i in range(3):
        run(bufs[i % n])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            run(bufs[(i + 3) % n])
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay(); g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e-3


shapes = [(576, 1280, 1280), (2304, 640, 640), (9216, 320, 320), (257, 1280, 5120), (257, 5120, 1280), (257, 3840, 1280), (257, 1280, 1280),
          (144, 1280, 1280), (77, 2560, 1024), (77, 1280, 1024), (77, 1024, 2560), (77, 640, 1024), (576, 1280, 5120), (2304, 640, 2560),
          (144, 1280, 2560), (576, 3840, 1280), (9216, 320, 1280), (1232, 768, 768), (1232, 3072, 768), (1232, 768, 3072), (1232, 2304, 768),
          (1024, 1280, 1280), (16, 1280, 1280), (1232, 1280, 768), (1232, 320, 768)]
b16 = [(4096, 1280, 1280), (16384, 640, 640), (4112, 1280, 1280), (1024, 1280, 1280), (1232, 768, 3072), (1232, 3072, 768), (1232, 768, 768), (16384, 640, 1920),
       (1232, 768, 2304), (1232, 2560, 768), (1232, 2304, 768), (1232, 1280, 768), (1232, 768, 2560), (1024, 1280, 10240), (4096, 1280, 2560), (1024, 1280, 2560),
       (1024, 1280, 5120), (1024, 3840, 1280), (4096, 640, 1280), (1024, 5120, 1280), (1024, 1280, 3840), (65536, 640, 320), (4096, 1280, 5120), (4096, 5120, 1280),
       (16384, 640, 2560), (65536, 320, 320), (4096, 3840, 1280), (16384, 1920, 640), (1024, 10240, 1280), (36864, 320, 320), (9216, 640, 640), (2304, 1280, 1280), (1028, 1280, 1280), (1028, 5120, 1280),
       (1028, 3840, 1280), (2304, 1280, 5120), (2304, 10240, 1280), (9216, 640, 2560), (9216, 5120, 640), (36864, 320, 1280), (36864, 2560, 320)]
DUMP = []
only = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:] if "x" in s]
if "b16" in sys.argv[1:]:
    shapes, only = b16, []
for M, N, K in (only or shapes):
    nkt = (K + 63) // 64
    res = []
    mk = lambda: (r(M, K), r(N, K), torch.empty((M, N), dtype=bf16, device=dev))
    nb = 2.0 * (M * K + N * K + M * N)
    for tile in (64, 128, 160, 5256, 512, 2320):
        if (tile == 160 and N % 160) or (tile == 512 and N % 256) or (tile == 2320 and N % 320) or (tile == 5256 and N % 128):
            continue
        for st in ((2, 3, 4) + ((5,) if EXP else ())) if tile < 1000 else (2,):       # 5 = 32-wide K-tiles, 4 stages (experimental build only)
            code = tile if st == 2 else st * 1000 + tile
            for sk in (1, 2, 3, 4, 6, 8):
                if sk > 1 and (nkt // sk < 2 or M * N * sk > 64e6):
                    continue
                try:
                    t = pool_time(mk, lambda b: hip.gemm(b[0], b[1], out=b[2], tile=code, splitk=sk), nb)
                except Exception as e:
                    continue
                res.append((t, tile, st, sk))
    auto = pool_time(mk, lambda b: hip.gemm(b[0], b[1], out=b[2]), nb)
    DUMP.append(dict(kind="gemm", M=M, N=N, K=K, auto=auto * 1e6, variants=[(tl, st, sk, t * 1e6) for t, tl, st, sk in res]))
    top = sorted(res)[:6]
    base = {(tl, st, sk): t for t, tl, st, sk in res}
    print(f"gemm M{M} N{N} K{K}: auto {auto*1e6:6.1f}us t64s2k1 {base.get((64, 2, 1), 0)*1e6:6.1f} t64s4k1 {base.get((64, 4, 1), 0)*1e6:6.1f} t128s2k1 {base.get((128, 2, 1), 0)*1e6:6.1f} | "
          + " ".join(f"t{tl}s{st}k{sk}:{t*1e6:5.1f}" for t, tl, st, sk in top) + f" | HBM floor {nb/5e12*1e6:.1f}us ({auto/top[0][0]:.2f}x)", flush=True)

convs16 = [(16, 8, 1280, 1280), (16, 8, 2560, 1280), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 16, 1920, 1280), (16, 16, 640, 1280), (16, 32, 640, 640),
           (16, 32, 1280, 640), (16, 32, 960, 640), (16, 32, 320, 640), (4, 12, 1280, 1280), (4, 24, 1280, 1280), (4, 48, 640, 640), (4, 96, 320, 320), (4, 24, 2560, 1280)]
convs = [(1, 12, 1280, 1280), (1, 24, 1280, 1280), (1, 24, 2560, 1280), (1, 12, 2560, 1280), (1, 48, 640, 640), (1, 96, 320, 320), (1, 48, 1280, 640), (1, 24, 640, 1280)]
if "b16" in sys.argv[1:]:
    convs = convs16
if not only:
    for B, H, Cin, Cout in convs:
        nkt = 9 * Cin // 64
        res = []
        mk = lambda: (r(B * H * H, Cin), r(Cout, 9 * Cin))
        nb = 2.0 * (B * H * H * (Cin + Cout) + Cout * 9 * Cin)
        for tile in (64, 128, 160, 512, 2320):
            if (tile == 160 and Cout % 160) or (tile == 512 and Cout % 256) or (tile == 2320 and Cout % 320):
                continue
            for st in ((2, 3, 4) if tile < 500 else (2,)):
                for sk in (1, 2, 3, 4, 6, 9, 12):
                    if sk > 1 and (nkt // sk < 4 or B * H * H * Cout * sk > 64e6):
                        continue
                    code = tile if st == 2 else st * 1000 + tile
                    try:
                        t = pool_time(mk, lambda b: hip.conv3x3(b[0], b[1], B, H, H, H, H, 1, tile=code, splitk=sk), nb, iters=12)
                    except Exception as e:
                        continue
                    res.append((t, tile, st, sk))
        auto = pool_time(mk, lambda b: hip.conv3x3(b[0], b[1], B, H, H, H, H, 1), nb, iters=12)
        DUMP.append(dict(kind="conv", B=B, H=H, Cin=Cin, Cout=Cout, M=B * H * H, N=Cout, K=9 * Cin, auto=auto * 1e6, variants=[(tl, st, sk, t * 1e6) for t, tl, st, sk in res]))
        top = sorted(res)[:7]
        print(f"conv B{B} {H}x{H} {Cin}->{Cout}: auto {auto*1e6:6.1f}us | " + " ".join(f"t{tl}s{st}k{sk}:{t*1e6:5.1f}" for t, tl, st, sk in top)
              + f" | HBM floor {nb/5e12*1e6:.1f}us ({auto/top[0][0]:.2f}x)", flush=True)

json.dump(DUMP, open(os.path.join(R, "gpurun_out", "sweep_small_m_%s.json" % ("b16" if "b16" in sys.argv[1:] else "b1")), "w"))
