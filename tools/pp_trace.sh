#!/bin/bash
# builds a PP_TRACE copy of the library into /tmp and runs the timeline dump (GPU box)
set -e
cd $GRAFT_REPO_ROOT/e4t-diffusion_amd/csrc
mkdir -p /tmp/ppobj
for f in core gemm attention norm wo elementwise; do
  if [ $f = gemm ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DPP_TRACE $PPFLAGS -c $f.hip -o /tmp/ppobj/$f.o; else cp obj/$f.o /tmp/ppobj/$f.o; fi
done
cp ../e4t/libe4t_hip.so /tmp/libe4t_hip.so.bak
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ppobj/*.o -o ../e4t/libe4t_hip.so
python $GRAFT_REPO_ROOT/tools/pp_trace.py
cp /tmp/libe4t_hip.so.bak ../e4t/libe4t_hip.so
