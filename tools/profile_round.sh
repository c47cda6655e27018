#!/bin/bash
# One round's measured evidence for bench.py's numbers, all from the same command (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats      -> gpurun_out/rNN_step_kernel_stats.csv  (+ the kernel trace, launch log)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (own runs, --kernel-trace only)
#   3. tools/roofline_report.py joins them per kernel symbol and per shape -> gpurun_out/rNN_roofline_per_shape.csv, rNN_pmc_traffic.csv
# usage: tools/profile_round.sh r02 [commit]
R=${1:-r02}; COMMIT=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline"
rm -rf /tmp/prof_stats; E4T_LAUNCH_LOG=/tmp/launch_stats.log timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > $OUT/${R}_step_rocprof_bench.log 2>&1 || tail -5 $OUT/${R}_step_rocprof_bench.log
f=$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${R}_step_kernel_stats.csv
CMD2="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-roofline"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  E4T_LAUNCH_LOG=/tmp/launch_$c.log timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -- $CMD2 > /tmp/pm_$c.log 2>&1 || tail -3 /tmp/pm_$c.log
done
python $ROOT/tools/roofline_report.py --trace /tmp/prof_stats --log /tmp/launch_stats.log --fetch /tmp/pm_FETCH_SIZE --fetch-log /tmp/launch_FETCH_SIZE.log \
  --write /tmp/pm_WRITE_SIZE --write-log /tmp/launch_WRITE_SIZE.log --out $OUT/${R} --commit $COMMIT > $OUT/${R}_roofline_report.log 2>&1
tail -3 $OUT/${R}_step_rocprof_bench.log; head -40 $OUT/${R}_roofline_per_shape.csv
