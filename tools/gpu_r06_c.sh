#!/bin/bash
# round 6, GPU call C: attention kernel checks + ab_attn for the given variants ("default" = in-tree build)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=e4t-diffusion_amd/e4t/variants
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | grep -v amdgpu.ids | tail -12
for v in "$@"; do
  if [ $v = default ]; then timeout 300 python tools/ab_attn.py default; else E4T_LIB=$V/libe4t_hip_$v.so timeout 300 python tools/ab_attn.py $v; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c_ab_attn.txt
