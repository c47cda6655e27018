#!/bin/bash
# SQ counters of the ping-pong conv kernels and the 128x128 GEMM on single shapes (tools/probe/pmc_gemm.py):
# separate rocprofv3 --pmc passes, averaged per launch and kernel -> gpurun_out/r03_gemm_pmc.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_LDS_DATA_FIFO_FULL" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pg_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pg_$i -- python $ROOT/tools/probe/pmc_gemm.py > /tmp/pg_$i.log 2>&1 || tail -3 /tmp/pg_$i.log
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pg_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        if not k.startswith("gemm_"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
with open("$OUT/r03_gemm_pmc.txt", "w") as fh:
    for k in sorted(agg):
        fh.write(k + "\n")
        for c in sorted(agg[k]):
            fh.write("   %-28s %16.0f per launch (%d launches)\n" % (c, agg[k][c] / n[k][c], n[k][c]))
print(open("$OUT/r03_gemm_pmc.txt").read())
PY
