#!/bin/bash
# round 6, evidence call: profile_round (kernel stats + PMC traffic + per-shape roofline) for the default command and for E4T_PREFETCH=0,
# idle report + main-stream gap analysis, the whole -m gpu suite, smoke(), the default bench (secondary configs + CPU leg)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
C=$1
if [ "$2" != "notests" ]; then
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r06_final_gpu_tests.txt 2>&1; stamp "pytest rc=$?"; tail -22 gpurun_out/r06_final_gpu_tests.txt
mkdir -p gpurun_out/r06_parity; cp gpurun_out/parity_*.json gpurun_out/r06_parity/ 2>/dev/null
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_final_smoke.txt 2>&1; stamp "smoke rc=$?"; tail -2 gpurun_out/r06_final_smoke.txt
bash tools/profile_round.sh r06 $C > gpurun_out/r06_profile_round.log 2>&1; stamp "profile default"
E4T_PREFETCH=0 bash tools/profile_round.sh r06_prefetch_off $C > gpurun_out/r06_prefetch_off_profile_round.log 2>&1; stamp "profile prefetch off"
R=$PWD
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_idle && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_idle -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline > $R/gpurun_out/r06_idle_rocprof.log 2>&1; python $R/tools/idle_report.py /tmp/prof_idle 4 > $R/gpurun_out/r06_idle_report.txt 2>&1; python $R/tools/stream_gaps.py /tmp/prof_idle 4 > $R/gpurun_out/r06_stream_gaps.txt 2>&1); head -4 gpurun_out/r06_idle_report.txt; stamp idle
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; stamp "bench rc=$?"; cat gpurun_out/r06_bench_default.json | tail -1 | cut -c1-3000
cp gpurun_out/bench_details.json gpurun_out/r06_bench_default_details.json
stamp done
