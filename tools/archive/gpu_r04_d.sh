#!/bin/bash
# round 4, GPU call D: prefetch placement (start vs bwd) with keyed slots
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "step_graph or prefetch or deterministic" > $O/r04d_tests.txt 2>&1; stamp "tests rc=$?"; tail -4 $O/r04d_tests.txt
bench() { local name=$1; shift
  env "$@" timeout 500 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary > $O/r04d_bench_$name.json 2> $O/r04d_bench_$name.err; stamp "bench $name rc=$?"; }
bench off E4T_PREFETCH=0
bench vitvae_bwd E4T_PREFETCH_AT=bwd
bench vitvae_start E4T_PREFETCH_AT=start
bench vit_start E4T_PREFETCH=vit E4T_PREFETCH_AT=start
bench vitvae_start_sidelo E4T_PREFETCH_AT=start E4T_SIDE_PRIORITY=1
bench vitvae_start_mainhi E4T_PREFETCH_AT=start E4T_MAIN_PRIORITY=-1
bench vitvae_start_again E4T_PREFETCH_AT=start
python - <<'PY'
import json
for n in ("off", "vitvae_bwd", "vitvae_start", "vit_start", "vitvae_start_sidelo", "vitvae_start_mainhi", "vitvae_start_again"):
    try:
        j = json.loads(open(f"gpurun_out/r04d_bench_{n}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "no result", e); continue
    print("%-22s ms/step %7.2f  img/s %6.1f" % (n, j["ms_per_step"], j["value"]))
PY
python - <<'PY'
import torch
print("stream priority range:", [torch.cuda.Stream(priority=p).priority for p in (-3, -2, -1, 0, 1, 2, 3)])
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/prof_idle; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_idle -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline > $R/$O/r04d_rocprof.log 2>&1
stamp "rocprof rc=$?"
python $R/tools/idle_report.py /tmp/prof_idle 4 2>&1 | head -4
stamp done
