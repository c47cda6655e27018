#!/bin/bash
# round 4, final GPU call: the whole -m gpu suite, smoke(), idle report of the default configuration, the default bench (secondary configs + CPU leg)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1400 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r04_final_gpu_tests.txt 2>&1; stamp "pytest rc=$?"; tail -22 gpurun_out/r04_final_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_final_smoke.txt 2>&1; stamp "smoke rc=$?"; tail -2 gpurun_out/r04_final_smoke.txt
timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; stamp "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_bench_default.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("ms/step %.2f img/s %.1f" % (j["ms_per_step"], j["value"]), "roofline", r["kernel"], "%.3f" % r["frac"], "traffic/alg", r.get("traffic_over_algorithmic"), "without side stream", r.get("without_side_stream"))
print("cpu_baseline", j.get("cpu_baseline")); print("parity", {k: v for k, v in (j.get("parity") or {}).items() if k in ("n_bad", "n_quantities", "case", "kink_elements_aligned")})
print("secondary", json.dumps(j.get("secondary"))[:1500])
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/prof_idle; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_idle -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline > $R/gpurun_out/r04_idle_rocprof.log 2>&1
python $R/tools/idle_report.py /tmp/prof_idle 4 > $R/gpurun_out/r04_idle_report.txt 2>&1; head -5 $R/gpurun_out/r04_idle_report.txt
stamp done
