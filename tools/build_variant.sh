#!/bin/bash
# Build a VARIANT of libe4t_hip.so with extra compile flags for ONE translation unit (compile-time A/B of kernel variants):
#   tools/build_variant.sh NAME UNIT "FLAGS"      ->  e4t-diffusion_amd/e4t/variants/libe4t_hip_NAME.so   (travels with gpurun; git-ignored)
# e.g. tools/build_variant.sh pair1 attention "-DATTN_FWD_PAIR=1".  Load it with E4T_LIB=<path> (e4t/_C.py).
set -e
cd "$(dirname "$0")/../e4t-diffusion_amd/csrc"
NAME=$1; UNIT=$2; EXTRA=$3
mkdir -p ../e4t/variants obj
[ -f obj/core.o ] || bash build.sh
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result $EXTRA -c $UNIT.hip -o /tmp/variant_${NAME}_$UNIT.o
OBJS=""
for f in core gemm attention norm wo elementwise image comm; do
  if [ $f = $UNIT ]; then OBJS="$OBJS /tmp/variant_${NAME}_$UNIT.o"; else OBJS="$OBJS obj/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o ../e4t/variants/libe4t_hip_$NAME.so
echo "built e4t/variants/libe4t_hip_$NAME.so"
