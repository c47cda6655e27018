#!/bin/bash
# round 5, GPU call B: kernel checks (delta folded into dQ, float4 split-K reduce, tails), prefetch / step-graph bitwise tests,
# A/B benches of the prefetch order (VAE first, two events) and start point, idle report of the default configuration, secondary configs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -q > gpurun_out/r05b_kernels.txt 2>&1; stamp "kernel checks rc=$?"; tail -12 gpurun_out/r05b_kernels.txt
timeout 600 python -m pytest tests/test_model_gpu.py -q > gpurun_out/r05b_model.txt 2>&1; stamp "model tests rc=$?"; tail -5 gpurun_out/r05b_model.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r05b_$name.json 2> gpurun_out/r05b_$name.err; python - "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(f"gpurun_out/r05b_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print(sys.argv[1], "ms/step %.2f" % j["ms_per_step"], "dominant frac", r.get("frac"), "by op", json.dumps(r.get("ms_per_step_by_op"))[:600])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run default E4T_X=0; stamp b1
run at_step E4T_PREFETCH_AT=step; stamp b2
run default_2 E4T_X=0; stamp b3
run vae_only E4T_PREFETCH=vae; stamp b4
run vit_only E4T_PREFETCH=vit; stamp b5
run prefetch_off E4T_PREFETCH=0; stamp b6
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-roofline > gpurun_out/r05b_secondary.json 2> gpurun_out/r05b_secondary.err; stamp "secondary rc=$?"
E4T_GEMM_NOTAIL=1 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-roofline > gpurun_out/r05b_secondary_notail.json 2> gpurun_out/r05b_secondary_notail.err; stamp "secondary notail rc=$?"
python - <<'PY'
import json
for n in ("secondary", "secondary_notail"):
    try:
        j = json.loads(open(f"gpurun_out/r05b_{n}.json").read().strip().splitlines()[-1])
        print(n, "%.2f" % j["ms_per_step"], j.get("secondary_ms_per_step"))
    except Exception as e:
        print(n, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/prof_idle; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_idle -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline > $R/gpurun_out/r05b_idle_rocprof.log 2>&1
python $R/tools/idle_report.py /tmp/prof_idle 4 > $R/gpurun_out/r05b_idle_report.txt 2>&1; head -8 $R/gpurun_out/r05b_idle_report.txt
stamp done
