#!/bin/bash
# round 3, GPU call B: 4-wave 256-row tiles (6128 / 6160): checks + sweep
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python tests/gpu_report.py gemm conv gemm_races > gpurun_out/r03b_kernel_checks.txt 2>&1; echo "kernel checks rc=$?"; grep -c "\[ok\]" gpurun_out/r03b_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error" gpurun_out/r03b_kernel_checks.txt | head -30
timeout 600 python tools/sweep_ps.py > gpurun_out/r03b_sweep.txt 2>&1; echo "sweep rc=$?"; tail -2 gpurun_out/r03b_sweep.txt
