#!/bin/bash
# round 5, GPU call A: kernel checks (incl. the new GEMM tail rows), glue profile, A/B benches (tail rows, CU-masked side stream),
# then every model-level parity case under the measured kink band (default since this round)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x > gpurun_out/r05a_kernels.txt 2>&1; stamp "kernel checks rc=$?"; tail -12 gpurun_out/r05a_kernels.txt
timeout 300 python tests/gpu_report.py gemm > gpurun_out/r05a_gemm_report.txt 2>&1; grep -i "tail" gpurun_out/r05a_gemm_report.txt | head -40
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r05a_$name.json 2> gpurun_out/r05a_$name.err; python - "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(f"gpurun_out/r05a_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print(sys.argv[1], "ms/step %.2f" % j["ms_per_step"], "dominant frac", r.get("frac"), "by op", json.dumps(r.get("ms_per_step_by_op"))[:700])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run tail_on E4T_X=0; stamp bench1
run tail_off E4T_GEMM_NOTAIL=1; stamp bench2
run tail_on_2 E4T_X=0; stamp bench3
run side_cus64 E4T_SIDE_CUS=64; stamp bench4
run side_cus128 E4T_SIDE_CUS=128; stamp bench5
run prefetch_off E4T_PREFETCH=0; stamp bench6
run prefetch_off_notail E4T_PREFETCH=0 E4T_GEMM_NOTAIL=1; stamp bench7
timeout 300 python tools/glue_profile.py sd14 16 > gpurun_out/r05a_glue.log 2>&1; stamp "glue rc=$?"; head -40 gpurun_out/glue_profile_sd14_b16.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py -q --durations=20 > gpurun_out/r05a_parity_tests.txt 2>&1; stamp "parity rc=$?"; tail -40 gpurun_out/r05a_parity_tests.txt
stamp done
