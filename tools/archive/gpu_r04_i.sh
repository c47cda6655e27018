#!/bin/bash
# round 4, GPU call I: default bench (new roofline block), idle report with the prefetch off, the offline B = 16 oracle comparison
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; stamp "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_bench_default.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("ms/step %.2f img/s %.1f" % (j["ms_per_step"], j["value"]), "| roofline", r["kernel"], "frac %.3f avg %.4f ms x %d" % (r["frac"], r["avg_launch_ms"], r["launches_per_step"]), "traffic/alg", r.get("traffic_over_algorithmic"))
print("in timed configuration:", {k: v for k, v in r.get("in_timed_configuration", {}).items() if k != "configuration"})
print("hbm:", {k: j["roofline_hbm"][k] for k in ("kernel", "frac", "ms_per_step")})
print("parity n_bad", j["parity"]["n_bad"], "cpu", j["cpu_baseline"]["value"])
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/prof_off; E4T_PREFETCH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_off -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline > $R/gpurun_out/r04_idle_rocprof_off.log 2>&1
python $R/tools/idle_report.py /tmp/prof_off 4 > $R/gpurun_out/r04_idle_report_prefetch_off.txt 2>&1; head -4 $R/gpurun_out/r04_idle_report_prefetch_off.txt
cd $R
timeout 1500 python tools/parity_b16_oracle.py --out gpurun_out/parity_full_sd14_b16_oracle.json > gpurun_out/r04_parity_b16_oracle.txt 2>&1; stamp "b16 oracle rc=$?"; tail -30 gpurun_out/r04_parity_b16_oracle.txt
stamp done
