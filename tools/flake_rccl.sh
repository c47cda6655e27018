#!/bin/bash
# run tests/rccl_one_rank.py N times, count failures, keep the failing logs (gpurun_out/flake_*.log)
N=${1:-10}; mkdir -p gpurun_out; bad=0
for i in $(seq 1 $N); do
  timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + i)) tests/rccl_one_rank.py > /tmp/flake_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ] || ! grep -q RCCL_ONE_RANK_OK /tmp/flake_$i.log; then bad=$((bad + 1)); cp /tmp/flake_$i.log gpurun_out/flake_$i.log; echo "run $i FAILED rc=$rc"; fi
done
echo "failures: $bad of $N"
