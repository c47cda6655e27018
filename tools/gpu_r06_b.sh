#!/bin/bash
# round 6, GPU call B: timing variants of attn_fwd64_kernel (ATTN64_PROBE builds)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=e4t-diffusion_amd/e4t/variants
for v in "$@"; do E4T_LIB=$V/libe4t_hip_$v.so timeout 120 python tools/ab_attn_fwd.py $v 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06b_probe.txt
