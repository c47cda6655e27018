#!/bin/bash
# round 4, GPU call A: kernel checks of the changed families, prefetch equivalence, A/B benches of every round-4 switch,
# batch consistency of the tuning step at its timed size, parity cases at the per-case kink band
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }

timeout 900 python tests/gpu_report.py gemm conv attention gemm_races > $O/r04a_kernel_checks.txt 2>&1; stamp "kernel checks rc=$?"
grep -c "\[ok\]" $O/r04a_kernel_checks.txt; grep "FAIL\|TOTAL\|Error\|error\|Traceback" $O/r04a_kernel_checks.txt | head -40

timeout 600 python -m pytest tests/test_model_gpu.py -q -x -k "prefetch or deterministic" --durations=10 > $O/r04a_prefetch_tests.txt 2>&1; stamp "prefetch tests rc=$?"; tail -15 $O/r04a_prefetch_tests.txt

bench() {  # name, env...
  local name=$1; shift
  env "$@" timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/r04a_bench_$name.json 2> $O/r04a_bench_$name.err
  stamp "bench $name rc=$?"
}
bench r3like E4T_PREFETCH=0 E4T_TN_NOXCD3=1 E4T_ATTN_NOTSPLIT=1 E4T_VIT_PANELS=0
bench kernels E4T_PREFETCH=0
bench vit_bwd E4T_PREFETCH=vit
bench vitvae_bwd E4T_PREFETCH=vit+vae
bench vitvae_start E4T_PREFETCH=vit+vae E4T_PREFETCH_AT=start
bench vitvae_bwd_mainhi E4T_PREFETCH=vit+vae E4T_MAIN_PRIORITY=-1
bench vitvae_bwd_sidelo E4T_PREFETCH=vit+vae E4T_SIDE_PRIORITY=1
python - <<'PY'
import json, glob
base = None
for n in ("r3like", "kernels", "vit_bwd", "vitvae_bwd", "vitvae_start", "vitvae_bwd_mainhi", "vitvae_bwd_sidelo"):
    try:
        j = json.loads(open(f"gpurun_out/r04a_bench_{n}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "no result", e); continue
    pk = j["roofline"]["per_kernel"]
    print("%-20s ms/step %7.2f  img/s %6.1f  sum(per_kernel) %.1f ms" % (n, j["ms_per_step"], j["value"], sum(v["ms_per_step"] for v in pk.values())))
    if n in ("r3like", "kernels"):
        for k, v in sorted(pk.items(), key=lambda kv: -kv[1]["ms_per_step"]):
            print("      %-18s %7.2f ms %5d launches %7.1f TF %7.0f GB/s" % (k, v["ms_per_step"], v["launches"], v["tflops"], v["gbps"]))
PY

timeout 600 python - > $O/r04a_tuning_full_b16.txt 2>&1 <<'PY'
import json, sys, os, time
sys.path[:0] = ["e4t-diffusion_amd", "tests", "oracle"]
import torch, parity_step
t = time.time()
rep = parity_step.batch_consistency("tuning_full", torch.device("cuda:0"), B=16)
rep["seconds"] = time.time() - t
json.dump(rep, open("gpurun_out/parity_tuning_full_batch16.json", "w"), indent=1)
print(json.dumps({k: v for k, v in rep.items() if k != "bad"}, indent=1)); print("bad:", rep["bad"][:16])
PY
stamp "tuning_full batch consistency rc=$?"; tail -30 $O/r04a_tuning_full_b16.txt

timeout 900 python -m pytest tests/test_configs_gpu.py -q --durations=20 -k "unfrozen_vit or sd2_real_width or tuning or tiny_sd2" > $O/r04a_config_tests.txt 2>&1; stamp "config parity tests rc=$?"; tail -30 $O/r04a_config_tests.txt
stamp done
