#!/bin/bash
# round 4, GPU call C: step-graph test, C5 with the step graph, idle / overlap report of the prefetch configuration
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "step_graph or prefetch or deterministic" > $O/r04c_graph_tests.txt 2>&1; stamp "graph tests rc=$?"; tail -25 $O/r04c_graph_tests.txt
timeout 900 python - > $O/r04c_secondary.txt 2>&1 <<'PY'
import json, sys, torch
sys.path[:0] = ["e4t-diffusion_amd", "."]
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
print(json.dumps(bench.secondary_configs(dev), indent=1))
PY
stamp "secondary rc=$?"; tail -60 $O/r04c_secondary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/prof_idle; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_idle -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-roofline > $R/$O/r04c_rocprof.log 2>&1
stamp "rocprof rc=$?"; tail -2 $R/$O/r04c_rocprof.log
python $R/tools/idle_report.py /tmp/prof_idle 4 > $R/$O/r04c_idle_report.txt 2>&1; head -40 $R/$O/r04c_idle_report.txt
stamp done
