#!/bin/bash
# builds a DMA_TRACE copy of the library and dumps the per-workgroup timeline of the LDS-DMA GEMM (GPU box)
set -e
cd $GRAFT_REPO_ROOT/e4t-diffusion_amd/csrc
mkdir -p /tmp/dtobj
for f in core gemm attention norm wo elementwise image; do
  if [ $f = gemm ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DDMA_TRACE -c $f.hip -o /tmp/dtobj/$f.o; else cp obj/$f.o /tmp/dtobj/$f.o; fi
done
cp ../e4t/libe4t_hip.so /tmp/libe4t_hip.so.bak
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/dtobj/*.o -o ../e4t/libe4t_hip.so
python $GRAFT_REPO_ROOT/tools/dma_trace.py "$@"; if [ -n "$TRACE2" ]; then python $GRAFT_REPO_ROOT/tools/dma_trace.py $TRACE2; fi; if [ -n "$TRACE3" ]; then python $GRAFT_REPO_ROOT/tools/dma_trace.py $TRACE3; fi
cp /tmp/libe4t_hip.so.bak ../e4t/libe4t_hip.so
