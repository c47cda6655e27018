#!/bin/bash
# round 3, GPU call C: new tile rules — kernel checks, step A/B (round-2 rules vs round-3 rules), full GPU test suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python tests/gpu_report.py gemm conv gemm_races > gpurun_out/r03c_kernel_checks.txt 2>&1; echo "kernel checks rc=$?"; grep -c "\[ok\]" gpurun_out/r03c_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error" gpurun_out/r03c_kernel_checks.txt | head -30
for v in R2 R3; do
  if [ $v = R2 ]; then export E4T_GEMM_R2RULES=1; else unset E4T_GEMM_R2RULES; fi
  timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r03c_bench_$v.json 2> gpurun_out/r03c_bench_$v.err; echo "bench $v rc=$?"
done
unset E4T_GEMM_R2RULES
python - <<'PY'
import json
for n in ("R2", "R3"):
    try:
        j = json.loads(open(f"gpurun_out/r03c_bench_{n}.json").read().strip().splitlines()[-1])
        pk = j["roofline"]["per_kernel"]
        print(n, "ms/step %.2f" % j["ms_per_step"], "img/s %.1f" % j["value"])
        for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["ms_per_step"])[:16]:
            print("   %-18s %7.2f ms %5d launches %7.1f TF %7.0f GB/s" % (k, v["ms_per_step"], v["launches"], v["tflops"], v["gbps"]))
    except Exception as e:
        print(n, "no result", e)
PY
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03c_pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"; tail -5 gpurun_out/r03c_pytest_gpu.txt
