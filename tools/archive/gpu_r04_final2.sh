#!/bin/bash
# round 4, final GPU call after the planner change: the whole -m gpu suite, smoke(), the default bench, the round's profile evidence regenerated
# at this commit (default command and E4T_PREFETCH=0), idle reports (B = 16 default / prefetch off, C5 at B = 1 replayed from the step graph)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
COMMIT=${1:-unknown}
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1300 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r04_final_gpu_tests.txt 2>&1; stamp "pytest rc=$?"; tail -20 gpurun_out/r04_final_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_final_smoke.txt 2>&1; stamp "smoke rc=$?"; tail -2 gpurun_out/r04_final_smoke.txt
timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; stamp "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_bench_default.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("ms/step %.2f img/s %.1f" % (j["ms_per_step"], j["value"]), "| roofline", r["kernel"], "frac %.3f avg %.4f ms x %d" % (r["frac"], r["avg_launch_ms"], r["launches_per_step"]), "traffic/alg", r.get("traffic_over_algorithmic"))
print("in timed configuration:", {k: v for k, v in r.get("in_timed_configuration", {}).items() if k != "configuration"})
print("parity", {k: v for k, v in (j.get("parity") or {}).items() if k in ("n_bad", "n_quantities", "case", "kink_elements_aligned")}, "cpu", j["cpu_baseline"]["value"])
print("secondary", json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "images_per_s", "ms_per_step_eager_launches")} for k, v in (j.get("secondary") or {}).items()}))
PY
bash tools/profile_round.sh r04 $COMMIT > gpurun_out/r04_profile_round.log 2>&1; stamp "profile_round rc=$?"
python tools/idle_report.py /tmp/prof_stats 4 > gpurun_out/r04_idle_report.txt 2>&1; head -4 gpurun_out/r04_idle_report.txt
E4T_PREFETCH=0 bash tools/profile_round.sh r04_prefetch_off $COMMIT > gpurun_out/r04_profile_round_prefetch_off.log 2>&1; stamp "profile_round prefetch off rc=$?"
python tools/idle_report.py /tmp/prof_stats 4 > gpurun_out/r04_idle_report_prefetch_off.txt 2>&1; head -4 gpurun_out/r04_idle_report_prefetch_off.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5; ( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python tools/c5_step.py 1 graph 8 ) > $R/gpurun_out/r04_c5_b1_rocprof.log 2>&1
python $R/tools/idle_report.py /tmp/prof_c5 6 > $R/gpurun_out/r04_c5_b1_idle_report.txt 2>&1; head -3 $R/gpurun_out/r04_c5_b1_idle_report.txt
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r04_c5_b1_kernel_stats.csv
stamp done
