#!/bin/bash
# round 3, GPU call D: GELU paths, the new model-level tests (B=16 consistency, full_sd21, full_sd14 with e_hat / AdamW / kink counts), default bench incl. secondary + CPU leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python tests/gpu_report.py gemm streaming > gpurun_out/r03d_kernel_checks.txt 2>&1; echo "kernel checks rc=$?"; grep -c "\[ok\]" gpurun_out/r03d_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error" gpurun_out/r03d_kernel_checks.txt | head -30
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -s > gpurun_out/r03d_fullsize.txt 2>&1; echo "fullsize rc=$?"; grep -E "parity\[|batch-consistency|passed|failed|Error|assert" gpurun_out/r03d_fullsize.txt | head -30
timeout 1800 python -m pytest "tests/test_configs_gpu.py::test_sd2_config_step_matches_oracle[full_sd21]" -m gpu -x -q -s > gpurun_out/r03d_sd21.txt 2>&1; echo "sd21 rc=$?"; grep -E "parity\[|passed|failed|Error|assert" gpurun_out/r03d_sd21.txt | head -30
timeout 1500 python bench.py > gpurun_out/r03d_bench_default.json 2> gpurun_out/r03d_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r03d_bench_default.json").read().strip().splitlines()[-1])
    print("ms/step %.2f img/s %.1f" % (j["ms_per_step"], j["value"]), "dominant", j["roofline"]["kernel"], "frac %.3f" % j["roofline"]["frac"])
    print("secondary", json.dumps(j.get("secondary"), indent=1))
    print("parity", json.dumps(j.get("parity"))[:1500])
    print("cpu", j.get("cpu_baseline"))
except Exception as e:
    print("no bench result", e)
PY
tail -3 gpurun_out/r03d_bench_default.err
