#!/bin/bash
# round 3, GPU call H: 256x320 tile rule: kernel checks + step A/B (E4T_GEMM_NOPQ)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python tests/gpu_report.py gemm conv gemm_races > gpurun_out/r03h_kernel_checks.txt 2>&1; echo "kernel checks rc=$?"; grep -c "\[ok\]" gpurun_out/r03h_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error" gpurun_out/r03h_kernel_checks.txt | head -30
for v in nopq pq; do
  if [ $v = nopq ]; then export E4T_GEMM_NOPQ=1; else unset E4T_GEMM_NOPQ; fi
  timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03h_bench_$v.json 2> gpurun_out/r03h_bench_$v.err; echo "bench $v rc=$?"
done
unset E4T_GEMM_NOPQ
python - <<'PY'
import json
for n in ("nopq", "pq"):
    try:
        j = json.loads(open(f"gpurun_out/r03h_bench_{n}.json").read().strip().splitlines()[-1])
        pk = j["roofline"]["per_kernel"]
        print(n, "ms/step %.2f" % j["ms_per_step"], "img/s %.1f" % j["value"])
        for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["ms_per_step"])[:14]:
            print("   %-18s %7.2f ms %5d launches %7.1f TF %7.0f GB/s" % (k, v["ms_per_step"], v["launches"], v["tflops"], v["gbps"]))
    except Exception as e:
        print(n, "no result", e)
PY
