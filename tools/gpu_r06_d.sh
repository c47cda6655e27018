#!/bin/bash
# round 6: timing variants of the backward (ATTN64_PROBE builds)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=e4t-diffusion_amd/e4t/variants
for v in "$@"; do
  if [ $v = default ]; then timeout 120 python tools/ab_attn_bwd.py default; else E4T_LIB=$V/libe4t_hip_$v.so timeout 120 python tools/ab_attn_bwd.py $v; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d_probe.txt
