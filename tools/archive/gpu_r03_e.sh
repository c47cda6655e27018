#!/bin/bash
# round 3, GPU call E: fp32 ViT residual stream (kernel checks, parity effect), B=16 batch-consistency test, capture guard
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python tests/gpu_report.py gemm norms > gpurun_out/r03e_kernel_checks.txt 2>&1; echo "kernel checks rc=$?"; grep -c "\[ok\]" gpurun_out/r03e_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error" gpurun_out/r03e_kernel_checks.txt | head -30
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py -m gpu -x -q -s > gpurun_out/r03e_fullsize.txt 2>&1; echo "fullsize+model rc=$?"; grep -E "parity\[|batch-consistency|passed|failed|Error|assert" gpurun_out/r03e_fullsize.txt | head -30
E4T_VIT_F32_RESIDUAL=0 timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03e_bench_bf16res.json 2> gpurun_out/r03e_bench_bf16res.err; echo "bench bf16res rc=$?"
timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03e_bench_f32res.json 2> gpurun_out/r03e_bench_f32res.err; echo "bench f32res rc=$?"
python - <<'PY'
import json
for n in ("bf16res", "f32res"):
    try:
        j = json.loads(open(f"gpurun_out/r03e_bench_{n}.json").read().strip().splitlines()[-1])
        pk = j["roofline"]["per_kernel"]
        print(n, "ms/step %.2f" % j["ms_per_step"], "img/s %.1f" % j["value"])
        for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["ms_per_step"])[:22]:
            print("   %-18s %7.2f ms %5d launches %7.1f TF %7.0f GB/s" % (k, v["ms_per_step"], v["launches"], v["tflops"], v["gbps"]))
    except Exception as e:
        print(n, "no result", e)
PY
