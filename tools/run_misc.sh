# one-off GPU-box checks: bench.py under torch.distributed.run (world 1, RCCL init + all-reduce path), inference kernel profile
cd $GRAFT_REPO_ROOT
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-roofline 2>&1 | grep -v "^$" | tail -4
export TMPDIR=/tmp; rm -rf /tmp/prof_inf
cd /tmp && STEPS=10 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o inf -- python $GRAFT_REPO_ROOT/tools/bench_inference.py > /tmp/prof_inf.log 2>&1
grep "images/call\|VAE" /tmp/prof_inf.log
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_inf
f=$(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1); head -40 "$f" > $GRAFT_REPO_ROOT/gpurun_out/prof_inf/inference_kernel_stats_top40.csv; head -14 "$f"
