#!/bin/bash
# Build libe4t_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [-j]
set -e
cd "$(dirname "$0")"
OUT=../e4t/libe4t_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
mkdir -p obj
pids=()
for f in core gemm gemm_ps attention norm wo elementwise image; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ common.h -nt obj/$f.o ] || { [[ $f == gemm* ]] && [ gemm_common.h -nt obj/$f.o ]; } || [ ../../include/e4t_hip.h -nt obj/$f.o ]; then
    EXTRA=""
    [ $f = image ] && EXTRA="-ffp-contract=off"      # byte-exact INTER_AREA: float ops must not be fused (see image.hip)
    hipcc $FLAGS $EXTRA -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC obj/*.o -o $OUT
echo "built $OUT"
