#!/bin/bash
# one GPU, the collective path forced on (1-rank RCCL group): where do the extra ~4.7 ms per step go?
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for fc in 1 0; do
  rm -rf /tmp/prof_c$fc
  E4T_FORCE_COMM=$fc timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$fc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline > $R/gpurun_out/comm$fc.log 2>&1
  python $R/tools/idle_report.py /tmp/prof_c$fc 4 > $R/gpurun_out/comm${fc}_idle.txt 2>&1
  cp $(ls -S /tmp/prof_c$fc/*/*kernel_stats.csv | head -1) $R/gpurun_out/comm${fc}_kernel_stats.csv
  head -3 $R/gpurun_out/comm${fc}_idle.txt
done
