#!/bin/bash
# Build libe4t_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh
# E4T_EXPERIMENTAL=1 csrc/build.sh additionally builds the measured-and-rejected GEMM variants (3 / 4-stage and 32-wide-K 64 / 128 tiles,
# 64-wide 256 x 128, 512 x 128 ping-pong, the persistent streaming kernels of gemm_ps.hip) for tools/sweep_*.py — not part of the product.
set -e
cd "$(dirname "$0")"
OUT=../e4t/libe4t_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
SRCS="core gemm attention norm wo elementwise image comm"
STAMP=obj/.experimental
mkdir -p obj
if [ -n "$E4T_EXPERIMENTAL" ]; then
  FLAGS="$FLAGS -DE4T_EXPERIMENTAL"; SRCS="$SRCS gemm_ps"
  [ -f $STAMP ] || { rm -f obj/gemm.o obj/core.o; touch $STAMP; }
else
  if [ -f $STAMP ]; then rm -f obj/gemm.o obj/core.o $STAMP; fi
  rm -f obj/gemm_ps.o
fi
pids=()
for f in $SRCS; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ common.h -nt obj/$f.o ] || { [[ $f == gemm* ]] && [ gemm_common.h -nt obj/$f.o ]; } || [ ../../include/e4t_hip.h -nt obj/$f.o ]; then
    EXTRA=""
    [ $f = image ] && EXTRA="-ffp-contract=off"      # byte-exact INTER_AREA: float ops must not be fused (see image.hip)
    [ $f = attention ] && EXTRA="-fno-slp-vectorize" # packed fp32 VALU beside MFMAs is slower than the scalar pair (attention.hip: ATTN_PK)
    hipcc $FLAGS $EXTRA -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC obj/*.o -ldl -o $OUT
echo "built $OUT"
