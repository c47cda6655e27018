#!/bin/bash
# round 4, GPU call O: kernel statistics of the B = 16 step with the small-grid stage rule (which instantiations run, what they sum to)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=$PWD/gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b16
( cd $R && E4T_PREFETCH=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b16 -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline ) > $O/r04o_rocprof.log 2>&1
f=$(find /tmp/prof_b16 -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04o_b16_kernel_stats.csv; grep "gemm_dma_kernel<64\|gemm_dma_kernel<128, 128, 4, 2, ., 4" $O/r04o_b16_kernel_stats.csv | cut -c1-170
