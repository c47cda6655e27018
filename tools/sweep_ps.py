"""GPU: the persistent 256 x BN streaming kernel (gemm_ps.hip, tile codes 1128 / 1160) against the tiles launch_gemm() picks
today, on the training step's own GEMM / conv shapes with cold operands (a pool larger than the Infinity Cache is cycled).
Reads the shape strings of profiles/r02_roofline_per_shape.csv; prints one line per shape + a weighted total.
usage: python tools/sweep_ps.py [csv] [top_n]"""
import csv
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import ops  # noqa: E402

hip = ops.HipBackend()
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "profiles", "r02_roofline_per_shape.csv")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 48


def pool_time(make, run, nbytes, iters=10):
    n = max(2, min(12, int(400e6 / max(nbytes, 1)) + 1))
    bufs = [make() for _ in range(n)]
    try:
        for i in range(2):
            run(bufs[i % n])
    except Exception as e:          # tile code not applicable to the shape
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(bufs[i % n])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(bf16)
rows = [x for x in csv.reader(l for l in open(path) if not l.startswith("#"))][1:]
rows = [x for x in rows if (x[1].startswith("gemm M") and " batch1 " in x[1]) or x[1].startswith("conv mode")]
rows.sort(key=lambda x: -float(x[4]))
seen, todo = set(), []
for x in rows:
    if x[1] not in seen and "flags9" not in x[1]:
        seen.add(x[1]); todo.append(x)
todo = todo[:top]
CODES = ((0, 0), (128, 0), (160, 0), (512, 0), (512, 1), (5256, 0), (2320, 0), (2320, 1))
tot = {c: 0.0 for c in CODES}
tot_best = tot_bestps = 0.0
for x in todo:
    shape, launches = x[1], int(x[2]) // 8
    m = re.match(r"gemm M(\d+) N(\d+) K(\d+) batch1 splitk(\d+) flags(\d+)", shape)
    if m:
        M, N, K, fl = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(5))
        w = r(N, K)
        make = lambda: r(M, K)
        call = lambda a, tile, sk: hip.gemm(a, w, tile=tile, splitk=sk, gelu=bool(fl & 4), out_dtype=torch.float32 if fl & 1 else bf16)
        nbytes = 2 * M * K + 2 * M * N
        flops = 2.0 * M * N * K
    else:
        m = re.match(r"conv mode(\d+) (\d+)x(\d+)->(\d+)x(\d+) Cin(\d+) Cout(\d+) M(\d+) splitk(\d+)", shape)
        if not m:
            continue
        mode, Hi, Wi, Ho, Wo, Ci, Co, M = (int(m.group(i)) for i in range(1, 9))
        B = M // (Ho * Wo)
        w = r(Co, 9 * Ci)
        make = lambda: r(B * Hi * Wi, Ci)
        call = lambda a, tile, sk: hip.conv3x3(a, w, B, Hi, Wi, Ho, Wo, mode, tile=tile, splitk=sk)
        nbytes = 2 * B * Hi * Wi * Ci + 2 * M * Co
        N, K = Co, 9 * Ci
        flops = 2.0 * M * N * K
    res = {}
    for tile, sk in CODES:
        if tile % 1000 == 160 and N % 160:
            continue
        if tile == 2320 and (N % 320 or K % 64):
            continue
        if tile == 512 and (N % 256 or K % 64):
            continue
        if tile in (1128, 5256) and N % 128 and N > 128:
            continue
        if 1000 <= tile < 5000 and K % 64:
            continue
        t = pool_time(make, lambda a: call(a, tile, sk), nbytes)
        if t is not None:
            res[(tile, sk)] = t
    auto = res.get((0, 0))
    if auto is None:
        continue
    best = min(res, key=res.get)
    tot_best += res[best] * launches
    tot[(0, 0)] += auto * launches
    alts = " ".join(f"{t}/{s}:{v:.1f}" for (t, s), v in sorted(res.items(), key=lambda kv: kv[1]))
    print(f"{shape:68s} x{launches:3d} auto {auto:7.1f} us {flops / auto / 1e6:6.0f} TF | best {str(best):10s} {res[best]:7.1f} us {flops / res[best] / 1e6:6.0f} TF "
          f"{nbytes / res[best] / 1e3:6.0f} GB/s {100 * (1 - res[best] / auto):5.1f}% | {alts}", flush=True)
print(f"per step over listed shapes: auto {tot[(0, 0)] / 1e3:.2f} ms   best-of-all {tot_best / 1e3:.2f} ms")
