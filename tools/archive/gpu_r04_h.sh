#!/bin/bash
# round 4, GPU call H: the round's profile evidence (rocprofv3 kernel stats, TCC traffic passes, per-shape roofline), idle / overlap report
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
COMMIT=$(cat gpurun_out/.commit 2>/dev/null || echo unknown)
bash tools/profile_round.sh r04 "$1"; stamp "profile_round rc=$?"
python tools/idle_report.py /tmp/prof_stats 4 > gpurun_out/r04_idle_report.txt 2>&1; head -6 gpurun_out/r04_idle_report.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/prof_off; E4T_PREFETCH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_off -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline > $R/gpurun_out/r04_rocprof_prefetch_off.log 2>&1
f=$(find /tmp/prof_off -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r04_step_kernel_stats_prefetch_off.csv
python $R/tools/idle_report.py /tmp/prof_off 4 > $R/gpurun_out/r04_idle_report_prefetch_off.txt 2>&1; head -4 $R/gpurun_out/r04_idle_report_prefetch_off.txt
stamp done
