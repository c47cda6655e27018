#!/bin/bash
# round 3, final GPU call: the whole -m gpu suite, smoke(), the probes' outputs and the default bench (secondary configs + CPU leg)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_final_gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_final_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_final_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03_final_smoke.txt
tools/probe/bin/mfma_rate > gpurun_out/r03_probe_mfma_rate.txt 2>&1
for v in 0 1 2 3 4 5 6 7 8; do [ -x tools/probe/bin/attn_probe_$v ] && timeout 60 tools/probe/bin/attn_probe_$v; done > gpurun_out/r03_probe_attn_variants.txt 2>&1
timeout 900 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03_bench_default.json").read().strip().splitlines()[-1])
print("ms/step %.2f img/s %.1f" % (j["ms_per_step"], j["value"]), "roofline", j["roofline"]["kernel"], "%.3f" % j["roofline"]["frac"], "traffic/alg", j["roofline"].get("traffic_over_algorithmic"))
print("cpu_baseline", j.get("cpu_baseline")); print("parity", {k: v for k, v in (j.get("parity") or {}).items() if k in ("n_bad", "n_compared", "case")})
print("secondary", json.dumps(j.get("secondary"))[:600])
PY
