#!/bin/bash
# Whole-step A/B of library variants on ONE box (box-to-box spread is +-3 %): the headline step with the in-tree library and with each
# named variant (tools/build_variant.sh NAME ...), alternating, twice.   bash tools/ab_step_variants.sh NAME [NAME ...]
cd "$(dirname "$0")/.."
for i in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then unset E4T_LIB; else export E4T_LIB=$PWD/e4t-diffusion_amd/e4t/variants/libe4t_hip_$v.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', round(d['ms_per_step'],2))"
done; done
