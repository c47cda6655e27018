#!/bin/bash
# round 3, GPU call F: k-step-phased ping-pong kernel (2256 / 2320): checks + sweep; batch-consistency with the corrected bound; kink band 1e-2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python tests/gpu_report.py gemm conv gemm_races > gpurun_out/r03f_kernel_checks.txt 2>&1; echo "kernel checks rc=$?"; grep -c "\[ok\]" gpurun_out/r03f_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error" gpurun_out/r03f_kernel_checks.txt | head -30
timeout 600 python tools/sweep_ps.py > gpurun_out/r03f_sweep.txt 2>&1; echo "sweep rc=$?"; tail -2 gpurun_out/r03f_sweep.txt
E4T_KINK_TOL=1e-2 timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -s > gpurun_out/r03f_fullsize.txt 2>&1; echo "fullsize rc=$?"; grep -E "parity\[|batch-consistency|passed|failed|Error|assert" gpurun_out/r03f_fullsize.txt | head -30
