#!/bin/bash
# round 4, GPU call K: per-kernel statistics of the C5 step at B = 1 (step graph) and B = 4
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
for B in 1 4; do
  rm -rf /tmp/prof_c5s
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5s -- python tools/c5_step.py $B graph 8 ) > $O/r04k_c5_b${B}_rocprof.log 2>&1
  f=$(find /tmp/prof_c5s -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04k_c5_b${B}_kernel_stats.csv; head -30 $O/r04k_c5_b${B}_kernel_stats.csv | cut -c1-160
done
