#!/bin/bash
# round 6: kernel trace of the default step -> idle report + main-stream gap analysis
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R=$PWD
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_idle && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_idle -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline > $R/gpurun_out/r06f_rocprof.log 2>&1; python $R/tools/idle_report.py /tmp/prof_idle 4 > $R/gpurun_out/r06f_idle_report.txt 2>&1; python $R/tools/stream_gaps.py /tmp/prof_idle 4 > $R/gpurun_out/r06f_stream_gaps.txt 2>&1)
head -4 gpurun_out/r06f_idle_report.txt; cat gpurun_out/r06f_stream_gaps.txt
