#!/bin/bash
# round 4, GPU call S: the small-grid cost model in the planner — kernel checks, sweeps (auto vs every variant), A/B against the round-3 rules
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python tests/gpu_report.py gemm conv gemm_races > $O/r04s_checks.txt 2>&1; stamp "checks rc=$?"; grep -c "\[ok\]" $O/r04s_checks.txt; grep "FAIL\|TOTAL\|Error\|error" $O/r04s_checks.txt | head
timeout 450 python tools/sweep_small_m.py > $O/r04s_sweep_b1.txt 2>&1; timeout 500 python tools/sweep_small_m.py b16 > $O/r04s_sweep_b16.txt 2>&1; stamp sweeps
python - <<'PY'
import json
for n in ("b1", "b16"):
    tot = [0, 0, 0]
    for e in json.load(open("gpurun_out/sweep_small_m_%s.json" % n)):
        best = min(v[3] for v in e["variants"])
        small = min(v[3] for v in e["variants"] if v[0] in (64, 128, 160))
        tot[0] += e["auto"]; tot[1] += best; tot[2] += small
        if e["auto"] > 1.07 * small:
            print("  %s M%d N%d K%d: auto %.1f, best small tile %.1f, best %.1f" % (e["kind"], e["M"], e["N"], e["K"], e["auto"], small, best))
    print(n, "sum auto %.0f  best small-tile variant %.0f  best %.0f us" % tuple(tot))
PY
for rep in 1 2; do
for v in on off; do
  if [ $v = off ]; then export E4T_GEMM_NOMODEL=1; else unset E4T_GEMM_NOMODEL; fi
  for B in 1 4; do timeout 300 python tools/c5_step.py $B graph 8 2>&1 | grep "C5 B" | sed "s/^/model $v: /"; done
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/r04s_bench_model_${v}_$rep.json 2> $O/r04s_bench_model_${v}_$rep.err
  python -c "
import json
j=json.loads(open('$O/r04s_bench_model_${v}_$rep.json').read().strip().splitlines()[-1]); print('model $v: B16', j['ms_per_step'], j['value'], 'parity bad', (j.get('parity') or {}).get('n_bad'))"
done
done
stamp done
