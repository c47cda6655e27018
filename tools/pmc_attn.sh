cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm -- python $GRAFT_REPO_ROOT/tools/one_kernel.py attnbwd > /tmp/pm.log 2>&1 || tail -5 /tmp/pm.log
  python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    import re
    m = re.search(r'attn_\w+<\d+>', r['Kernel_Name'])
    if not m: continue
    k = m.group(0)
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    print(k, {c: f"{v:.3g}" for c, v in d.items()})
PY
done
