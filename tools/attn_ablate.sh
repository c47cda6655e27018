#!/bin/bash
# timing ablation: how much of the attention kernels is the transposed LDS staging (store_T)?
set -e
cd $GRAFT_REPO_ROOT/e4t-diffusion_amd/csrc
mkdir -p /tmp/abobj
cp obj/*.o /tmp/abobj/
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $ABFLAGS -c attention.hip -o /tmp/abobj/attention.o
cp ../e4t/libe4t_hip.so /tmp/libe4t_hip.so.bak
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abobj/*.o -o ../e4t/libe4t_hip.so
python $GRAFT_REPO_ROOT/tests/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids | head -4
cp /tmp/libe4t_hip.so.bak ../e4t/libe4t_hip.so
