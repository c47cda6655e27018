# one-off GPU-box run: rocprofv3 kernel statistics of the data-path benchmark (image_prep kernel)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; rm -rf /tmp/prof_data
cd /tmp && timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_data -o data -- python $GRAFT_REPO_ROOT/tools/bench_data.py > /tmp/prof_data.log 2>&1
grep "image_prep\|DeviceLoader\|oracle" /tmp/prof_data.log
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_data
grep "image_prep\|DeviceLoader\|oracle" /tmp/prof_data.log > $GRAFT_REPO_ROOT/gpurun_out/prof_data/bench_data.txt
f=$(find /tmp/prof_data -name "*kernel_stats.csv" | head -1); head -12 "$f" > $GRAFT_REPO_ROOT/gpurun_out/prof_data/data_kernel_stats.csv; head -4 "$f"
