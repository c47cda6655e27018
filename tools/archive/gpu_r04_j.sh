#!/bin/bash
# round 4, GPU call J: (1) the prefetch tests after the VAE-only stream-ordering fix; (2) kernel traces of the C5 step at B = 1, launched from the
# host and replayed from the step graph -> per-stream busy / idle report (is the B = 1 step host-bound or device-bound?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 400 python -m pytest tests/test_model_gpu.py -q -k "prefetch or step_graph" > $O/r04j_prefetch_tests.txt 2>&1; stamp "prefetch tests rc=$?"; tail -3 $O/r04j_prefetch_tests.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in eager graph; do
  ( cd $R && timeout 200 python tools/c5_step.py 1 $mode 8 ) > $O/r04j_c5_b1_$mode.txt 2>&1; grep "C5 B" $O/r04j_c5_b1_$mode.txt
  rm -rf /tmp/prof_c5_$mode
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c5_$mode -- python tools/c5_step.py 1 $mode 8 ) > $O/r04j_c5_b1_${mode}_rocprof.log 2>&1
  python $R/tools/idle_report.py /tmp/prof_c5_$mode 6 > $O/r04j_c5_b1_${mode}_idle.txt 2>&1; head -40 $O/r04j_c5_b1_${mode}_idle.txt
  stamp "trace $mode"
done
stamp done
