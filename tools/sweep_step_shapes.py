"""GPU: tile / split-K choice of launch_gemm() re-checked on the training step's own shapes.

Reads the GEMM and 3x3-conv rows of profiles/r02_roofline_per_shape.csv (shape strings written by the launch log), times every
tile in {64, 128, 160, 256 x 256 ping-pong} x split-K in {auto, 1, 2, 3, 4, 6, 8} that the launcher accepts on cold operands (a
pool larger than the 256 MB Infinity Cache is cycled), and prints, per shape, what the heuristic picks against the best found —
weighted by the shape's share of the step.  usage: python tools/sweep_step_shapes.py [csv] [top_n]"""
import csv
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "e4t-diffusion_amd"), os.path.join(R, "tests")]
import torch  # noqa: E402
from e4t import _C, ops  # noqa: E402

hip = ops.HipBackend()
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "profiles", "r02_roofline_per_shape.csv")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 28


def pool_time(make, run, nbytes, iters=12):
    n = max(2, min(16, int(500e6 / max(nbytes, 1)) + 1))
    bufs = [make() for _ in range(n)]
    try:
        for i in range(2):
            run(bufs[i % n])
    except Exception:
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
tot_auto = tot_best = 0.0
for x in todo:
    shape, launches, in_step_us = x[1], int(x[2]), float(x[3])
    m = re.match(r"gemm M(\d+) N(\d+) K(\d+) batch1 splitk(\d+) flags(\d+)", shape)
    if m:
        M, N, K = int(m.group(1)), int(m.group(2)), int(m.group(3))
        f32out = int(m.group(5)) & 1
        w = r(N, K)
        make = lambda: r(M, K)
        call = lambda a, tile, sk: hip.gemm(a, w, tile=tile, splitk=sk, out_dtype=torch.float32 if f32out else bf16)
        nbytes = 2 * M * K + 2 * M * N
        tiles = (0, 64, 128, 160, 256, 512)
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
        tiles = (0, 64, 128, 160, 256, 512)
    res = {}
    for tile in tiles:
        if tile == 160 and N % 160:
            continue
        if tile == 512 and (N % 256 or K % 64):
            continue
        for sk in (0, 1, 2, 3, 4, 6, 8):
            if tile == 0 and sk:
                continue
            if sk > 1 and (K // 64) // sk < 8:
                continue
            t = pool_time(make, lambda a: call(a, tile, sk), nbytes)
            if t is not None:
                res[(tile, sk)] = t
    auto = res.get((0, 0))
    if auto is None:
        continue
    best = min(res, key=res.get)
    tot_auto += auto * launches; tot_best += res[best] * launches
    flag = "" if res[best] > 0.95 * auto else "   <== %.0f%%" % (100 * (1 - res[best] / auto))
    alts = " ".join(f"{t}/{s}:{v:.0f}" for (t, s), v in sorted(res.items(), key=lambda kv: kv[1])[:4])
    print(f"{shape:70s} x{launches:4d} in-step {in_step_us:7.1f} us  auto {auto:7.1f}  best {best} {res[best]:7.1f}{flag}   [{alts}]", flush=True)
print(f"sum over listed launches: auto {tot_auto / 1e3:.1f} ms   best {tot_best / 1e3:.1f} ms  ({100 * (1 - tot_best / tot_auto):.1f}% less)")
