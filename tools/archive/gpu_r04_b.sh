#!/bin/bash
# round 4, GPU call B: race check re-run, prefetch placement A/B after the event fix, ViT panels off, dK/dV and TN micro A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python tests/gpu_report.py gemm_races > $O/r04b_races.txt 2>&1; stamp "races rc=$?"; grep "FAIL\|TOTAL" $O/r04b_races.txt | head
timeout 300 python -m pytest tests/test_model_gpu.py -q -x -k "prefetch" > $O/r04b_prefetch_tests.txt 2>&1; stamp "prefetch tests rc=$?"; tail -3 $O/r04b_prefetch_tests.txt
bench() { local name=$1; shift
  env "$@" timeout 500 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary > $O/r04b_bench_$name.json 2> $O/r04b_bench_$name.err; stamp "bench $name rc=$?"; }
bench off E4T_PREFETCH=0
bench vit_bwd E4T_PREFETCH=vit
bench vitvae_bwd E4T_PREFETCH=vit+vae
bench vitvae_start E4T_PREFETCH=vit+vae E4T_PREFETCH_AT=start
bench vit_start E4T_PREFETCH=vit E4T_PREFETCH_AT=start
bench vitvae_bwd_again E4T_PREFETCH=vit+vae
bench off_notsplit E4T_PREFETCH=0 E4T_ATTN_NOTSPLIT=1
bench off_noxcd3 E4T_PREFETCH=0 E4T_TN_NOXCD3=1
python - <<'PY'
import json
for n in ("off", "vit_bwd", "vitvae_bwd", "vitvae_start", "vit_start", "vitvae_bwd_again", "off_notsplit", "off_noxcd3"):
    try:
        j = json.loads(open(f"gpurun_out/r04b_bench_{n}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "no result", e); continue
    pk = j["roofline"]["per_kernel"]
    print("%-20s ms/step %7.2f  img/s %6.1f  attn_bwd40 %.2f gemm_tn %.2f gemm128 %.2f gemm2320 %.2f gemm64 %.2f" % (n, j["ms_per_step"], j["value"], pk["attn_bwd40"]["ms_per_step"], pk["gemm_tn"]["ms_per_step"], pk["gemm128"]["ms_per_step"], pk["gemm2320"]["ms_per_step"], pk["gemm64"]["ms_per_step"]))
PY
# micro A/B: cross-attention backward with / without query chunks; TN GEMM with / without the 3-D re-deal
timeout 300 python - <<'PY'
import os, sys, time, subprocess
code = r'''
import sys, torch, time
sys.path[:0] = ["e4t-diffusion_amd", "tests"]
from e4t import ops
be = ops.HipBackend(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
B, H, T, S, DH = 16, 8, 4096, 77, 40
d = H * DH
q = torch.randn(B * T, d, device=dev, generator=g).bfloat16(); kv = torch.randn(B * S, 2 * d, device=dev, generator=g).bfloat16()
k, v = kv[:, :d], kv[:, d:]
o, lse = be.attention_fwd(q, k, v, B, H, T, S, DH, DH ** -0.5)
do = torch.randn_like(o); dq = torch.empty_like(q); dkv = torch.empty_like(kv)
print("cross-attn bwd B16 H8 T4096 S77 dh40: %.1f us" % bench(lambda: be.attention_bwd(q, k, v, o, do, lse, dq, dkv[:, :d], dkv[:, d:], B, H, T, S, DH, DH ** -0.5)))
for (M, N, K) in [(960, 320, 65536), (320, 320, 65536), (3840, 1280, 4096), (1920, 640, 16384)]:
    dy = torch.randn(K, M, device=dev, generator=g).bfloat16(); x = torch.randn(K, N, device=dev, generator=g).bfloat16()
    out = torch.zeros(M, N, device=dev)
    print("gemm_tn M%d N%d K%d: %.1f us" % (M, N, K, bench(lambda: be.gemm_tn(dy, x, out=out, accum=True))))
'''
for env in ({}, {"E4T_ATTN_NOTSPLIT": "1", "E4T_TN_NOXCD3": "1"}):
    print("== env", env, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env))
PY
stamp done
