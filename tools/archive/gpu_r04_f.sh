#!/bin/bash
# round 4, GPU call F: knob sweeps on the step (E4T_GN_BLOCKS, E4T_GEMM_GM, E4T_ATTN_DKV_OCC)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
bench() { local name=$1; shift
  env "$@" timeout 500 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-secondary > $O/r04f_bench_$name.json 2> $O/r04f_bench_$name.err; stamp "bench $name rc=$?"; }
bench base X=1
for v in 256 512 768 1536 2048 4096; do bench gn$v E4T_GN_BLOCKS=$v; done
for v in 4 16; do bench gm$v E4T_GEMM_GM=$v; done
for v in 2 3; do bench dkvocc$v E4T_ATTN_DKV_OCC=$v; done
bench base2 X=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04f_bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); pk = j["roofline"]["per_kernel"]
        print("%-14s ms/step %7.2f  gn_fwd %.2f+%.2f gn_bwd %.2f attn_bwd40 %.2f conv512 %.2f gemm128 %.2f" % (f.split("bench_")[1][:-5], j["ms_per_step"], pk["gn_fwd_colstats"]["ms_per_step"], pk["gn_fwd_2pass"]["ms_per_step"], pk["gn_bwd"]["ms_per_step"], pk["attn_bwd40"]["ms_per_step"], pk["conv512"]["ms_per_step"], pk["gemm128"]["ms_per_step"]))
    except Exception as e:
        print(f, "no result", e)
PY
stamp done
