#!/bin/bash
# usage: tools/pmc.sh <kernel-name-regex> <python args...>     e.g. tools/pmc.sh gemm_pp tools/one_kernel.py gemm 512
cd /tmp && export TMPDIR=/tmp
RE="$1"; shift
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm -- python $GRAFT_REPO_ROOT/"$@" > /tmp/pm.log 2>&1 || tail -5 /tmp/pm.log
  RE="$RE" python - <<'PY'
import csv, glob, collections, re, os
f = glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    m = re.search(os.environ["RE"] + r"[\w<>, ]*", r['Kernel_Name'])
    if not m: continue
    k = m.group(0)[:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY
done
