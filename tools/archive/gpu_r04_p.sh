#!/bin/bash
# round 4, GPU call P: A/B of the small-grid stage rule (E4T_GEMM_NOSTAGES=1 = two stages everywhere), B = 16 headline and C5 at B = 1 / 4
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
for rep in 1 2; do
for v in on off; do
  if [ $v = off ]; then export E4T_GEMM_NOSTAGES=1; else unset E4T_GEMM_NOSTAGES; fi
  for B in 1 4; do timeout 300 python tools/c5_step.py $B graph 8 2>&1 | grep "C5 B" | sed "s/^/stages $v: /"; done
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/r04p_bench_stages_${v}_$rep.json 2> $O/r04p_bench_stages_${v}_$rep.err
  python -c "
import json
j=json.loads(open('$O/r04p_bench_stages_${v}_$rep.json').read().strip().splitlines()[-1]); print('stages $v: B16', j['ms_per_step'], j['value'])"
done
done
