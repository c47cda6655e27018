#!/bin/bash
# round 3, GPU call A: the persistent GEMM kernel — tiny smoke first (a hang must not eat the box), kernel checks, shape sweep, step A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=e4t-diffusion_amd:tests
timeout 120 python - > gpurun_out/r03a_smoke.txt 2>&1 <<'PY' || { echo "ps smoke failed"; tail -5 gpurun_out/r03a_smoke.txt; exit 1; }
import torch
from e4t import ops
from emu_backend import EmuBackend
hip, emu = ops.HipBackend(), EmuBackend()
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K, t) in [(256, 128, 128, 1128), (256, 160, 128, 1160), (700, 320, 320, 1160), (70000, 320, 320, 1160), (70000, 256, 64, 1128)]:
    a = torch.randn(M, K, generator=g, device="cuda").bfloat16(); b = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, generator=g, device="cuda")
    y = hip.gemm(a, b, bias=bias, tile=t); torch.cuda.synchronize()
    r = emu.gemm(a, b, bias=bias)
    print(M, N, K, t, float((y.float() - r.float()).norm() / r.float().norm()), flush=True)
PY
cat gpurun_out/r03a_smoke.txt
timeout 600 python tests/gpu_report.py gemm conv gemm_races > gpurun_out/r03a_kernel_checks.txt 2>&1; echo "kernel checks rc=$?"; grep -c "\[ok\]" gpurun_out/r03a_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error" gpurun_out/r03a_kernel_checks.txt | head -30
timeout 500 python tools/sweep_ps.py > gpurun_out/r03a_sweep_pre1.txt 2>&1; echo "sweep rc=$?"; tail -3 gpurun_out/r03a_sweep_pre1.txt
E4T_PS_PRE=0 timeout 300 python tools/sweep_ps.py profiles/r02_roofline_per_shape.csv 24 > gpurun_out/r03a_sweep_pre0.txt 2>&1; tail -1 gpurun_out/r03a_sweep_pre0.txt
E4T_GEMM_PS=0 timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03a_bench_ps0.json 2> gpurun_out/r03a_bench_ps0.err; echo "bench ps0 rc=$?"
E4T_GEMM_PS=1 timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03a_bench_ps1.json 2> gpurun_out/r03a_bench_ps1.err; echo "bench ps1 rc=$?"
python - <<'PY'
import json
for n in ("ps0", "ps1"):
    try:
        j = json.loads(open(f"gpurun_out/r03a_bench_{n}.json").read().strip().splitlines()[-1])
        pk = j["roofline"]["per_kernel"]
        print(n, "ms/step %.2f" % j["ms_per_step"], "img/s %.1f" % j["value"])
        for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["ms_per_step"])[:14]:
            print("   %-18s %7.2f ms %5d launches %7.1f TF %7.0f GB/s" % (k, v["ms_per_step"], v["launches"], v["tflops"], v["gbps"]))
    except Exception as e:
        print(n, "no result", e)
PY
