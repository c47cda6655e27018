#!/bin/bash
# PMC view of the dh-40 backward attention kernels (tools/ab_attn_bwd.py): where the wave cycles go.  Separate passes (8 counters each).
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/pmc_attn
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -- python $R/tools/ab_attn_bwd.py pmc > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' >> $R/gpurun_out/pmc_attn/summary.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    import re
    m=re.search(r'attn_\w+', r['Kernel_Name'])
    if not m: continue
    k=m.group(0)
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k,v in acc.items():
    print(k)
    for c,x in sorted(v.items()): print(f"   {c:<28s} {x/max(n[(k,c)],1):16.0f}  (per dispatch, {n[(k,c)]} dispatches)")
PY
  tail -2 /tmp/pmc_$i.log | cut -c1-200
done
cat $R/gpurun_out/pmc_attn/summary.txt
