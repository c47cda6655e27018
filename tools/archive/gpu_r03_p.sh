#!/bin/bash
# round 3, GPU call O (incremental walker): channel-chunk-major K order of the stride-1 3x3 convs: kernel checks + step A/B (E4T_CONV_TAPMAJOR)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python tests/gpu_report.py conv gemm gemm_races > gpurun_out/r03p_kernel_checks.txt 2>&1; echo "kernel checks rc=$?"; grep -c "\[ok\]" gpurun_out/r03p_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error" gpurun_out/r03p_kernel_checks.txt | head -30
for v in tapmajor chanmajor; do
  if [ $v = tapmajor ]; then export E4T_CONV_TAPMAJOR=1; else unset E4T_CONV_TAPMAJOR; fi
  timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03p_bench_$v.json 2> gpurun_out/r03p_bench_$v.err; echo "bench $v rc=$?"
done
unset E4T_CONV_TAPMAJOR
python - <<'PY'
import json
for n in ("tapmajor", "chanmajor"):
    try:
        j = json.loads(open(f"gpurun_out/r03p_bench_{n}.json").read().strip().splitlines()[-1])
        pk = j["roofline"]["per_kernel"]
        print(n, "ms/step %.2f" % j["ms_per_step"], "img/s %.1f" % j["value"])
        for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["ms_per_step"])[:30]:
            if "conv" in k: print("   %-18s %7.2f ms %5d launches %7.1f TF %7.0f GB/s" % (k, v["ms_per_step"], v["launches"], v["tflops"], v["gbps"]))
    except Exception as e:
        print(n, "no result", e)
PY
