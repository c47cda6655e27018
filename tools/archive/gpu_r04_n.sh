#!/bin/bash
# round 4, GPU call N: 3 / 4-stage small-grid GEMM rule — kernel checks (gemm, conv, races), small-M sweep (auto vs variants), C5 step at B = 1 / 4,
# the B = 16 headline with and without the rule
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python tests/gpu_report.py gemm conv gemm_races > $O/r04n_checks.txt 2>&1; stamp "checks rc=$?"; grep -c "\[ok\]" $O/r04n_checks.txt; grep "FAIL\|TOTAL\|Error\|error" $O/r04n_checks.txt | head
timeout 600 python tools/sweep_small_m.py > $O/r04n_sweep_small_m.txt 2>&1; stamp "sweep rc=$?"; cut -c1-260 $O/r04n_sweep_small_m.txt
for v in "" 1; do
  for B in 1 4; do E4T_GEMM_NOSTAGES=$v timeout 300 python tools/c5_step.py $B graph 8 2>&1 | grep "C5 B" | sed "s/^/nostages='$v' /"; done
  E4T_GEMM_NOSTAGES=$v timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/r04n_bench_nostages$v.json 2> $O/r04n_bench_nostages$v.err
  python -c "
import json,sys
j=json.loads(open('$O/r04n_bench_nostages$v.json').read().strip().splitlines()[-1]); print('nostages=$v', j['ms_per_step'], j['value'])"
  stamp "bench nostages='$v'"
done
