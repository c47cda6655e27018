#!/bin/bash
# round 6, GPU call E: adamw_rank kernel checks, the step with the factored head update against the materialised stack
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "streaming" 2>&1 | grep -v amdgpu.ids | tail -8
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
run() { name=$1; shift; timeout 400 $B "$@" > gpurun_out/r06e_$name.json 2> gpurun_out/r06e_$name.err; python - "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(f"gpurun_out/r06e_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print(sys.argv[1], "ms/step %.2f" % j["ms_per_step"], "loss", j["config"].get("last_loss"), "by op", json.dumps(r.get("ms_per_step_by_op"))[:900])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/r06e_{sys.argv[1]}.err").read()[-1500:])
PY
}
run factored
run materialised --materialise-head-grad
run factored_2
