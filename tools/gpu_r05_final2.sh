#!/bin/bash
# round 5, closing evidence call at the final commit: the whole -m gpu suite, smoke(), the default bench (kernels unchanged since the
# profile passes of gpu_r05_final.sh at 84dc52a; only host-side changes after it)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r05_final_gpu_tests.txt 2>&1; stamp "pytest rc=$?"; tail -22 gpurun_out/r05_final_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_final_smoke.txt 2>&1; stamp "smoke rc=$?"; tail -2 gpurun_out/r05_final_smoke.txt
timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; stamp "bench rc=$?"; tail -1 gpurun_out/r05_bench_default.json | cut -c1-3000
cp gpurun_out/bench_details.json gpurun_out/r05_bench_default_details.json
stamp done
