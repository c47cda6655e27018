#!/bin/bash
# round 4, GPU call E: prepare-ahead of the next step's weights, VAE-only prefetch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "step_graph or prefetch or deterministic or smoke" > $O/r04e_tests.txt 2>&1; stamp "tests rc=$?"; tail -4 $O/r04e_tests.txt
bench() { local name=$1; shift
  env "$@" timeout 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-roofline > $O/r04e_bench_$name.json 2> $O/r04e_bench_$name.err; stamp "bench $name rc=$?"; }
for rep in 1 2; do
bench off_$rep E4T_PREFETCH=0 E4T_PREPARE_AHEAD=0
bench vitvae_noahead_$rep E4T_PREPARE_AHEAD=0
bench vitvae_$rep E4T_PREPARE_AHEAD=1
bench vae_$rep E4T_PREFETCH=vae
bench off_ahead_$rep E4T_PREFETCH=0
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04e_bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s ms/step %7.2f  img/s %6.1f" % (f.split("bench_")[1][:-5], j["ms_per_step"], j["value"]))
    except Exception as e:
        print(f, "no result", e)
PY
stamp done
