#!/bin/bash
# round 3, GPU call I: attention XCD-aware block raster: kernel checks + step A/B (E4T_ATTN_NOXCD)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python tests/gpu_report.py attention > gpurun_out/r03i_kernel_checks.txt 2>&1; echo "kernel checks rc=$?"; grep -c "\[ok\]" gpurun_out/r03i_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error" gpurun_out/r03i_kernel_checks.txt | head -30
for v in noxcd xcd; do
  if [ $v = noxcd ]; then export E4T_ATTN_NOXCD=1; else unset E4T_ATTN_NOXCD; fi
  timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03i_bench_$v.json 2> gpurun_out/r03i_bench_$v.err; echo "bench $v rc=$?"
done
unset E4T_ATTN_NOXCD
python - <<'PY'
import json
for n in ("noxcd", "xcd"):
    try:
        j = json.loads(open(f"gpurun_out/r03i_bench_{n}.json").read().strip().splitlines()[-1])
        pk = j["roofline"]["per_kernel"]
        print(n, "ms/step %.2f" % j["ms_per_step"], "img/s %.1f" % j["value"])
        for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["ms_per_step"])[:30]:
            if "attn" in k: print("   %-18s %7.2f ms %5d launches %7.1f TF %7.0f GB/s" % (k, v["ms_per_step"], v["launches"], v["tflops"], v["gbps"]))
    except Exception as e:
        print(n, "no result", e)
PY
