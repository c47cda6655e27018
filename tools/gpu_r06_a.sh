#!/bin/bash
# round 6, GPU call A: the 64-queries-per-wave dh-40 forward (attn_fwd64_kernel): kernel checks, micro-benchmark against the old
# kernel and two build variants (one wave per SIMD; no scheduling fences), then the step with and without it on the same box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
V=e4t-diffusion_amd/e4t/variants
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" > gpurun_out/r06a_attn_checks.txt 2>&1; stamp "attention checks rc=$?"; tail -15 gpurun_out/r06a_attn_checks.txt
for v in default a_old a_occ1 a_nofence; do
  if [ $v = default ]; then timeout 300 python tools/ab_attn.py default; else E4T_LIB=$V/libe4t_hip_$v.so timeout 300 python tools/ab_attn.py $v; fi
done > gpurun_out/r06a_ab_attn.txt 2>&1; stamp "ab_attn"; cat gpurun_out/r06a_ab_attn.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
run() { name=$1; shift; env "$@" timeout 400 $B > gpurun_out/r06a_$name.json 2> gpurun_out/r06a_$name.err; python - "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(f"gpurun_out/r06a_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print(sys.argv[1], "ms/step %.2f" % j["ms_per_step"], "dominant frac", r.get("frac"), "by op", json.dumps(r.get("ms_per_step_by_op"))[:900])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run new E4T_X=0; stamp bench1
run old E4T_LIB=$V/libe4t_hip_a_old.so; stamp bench2
run new_2 E4T_X=0; stamp bench3
stamp done
