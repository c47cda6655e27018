#!/bin/bash
# step time under runtime environment knobs (one box, back to back): bash tools/ab_env.sh "VAR=val" "VAR2=val" ...   ("" = baseline)
for kv in "$@"; do
  for rep in 1 2; do
    env $kv timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-roofline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$kv]', round(d['ms_per_step'],2), 'ms/step', round(d['value'],1), 'img/s')"
  done
done
