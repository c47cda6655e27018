#!/bin/bash
# HBM traffic of every kernel symbol of one bench run, from the TCC counters (separate passes, as the guide prescribes).
# usage: tools/pmc_traffic.sh  -> gpurun_out/r01_pmc_traffic.csv
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline > /tmp/pm_$c.log 2>&1 || tail -3 /tmp/pm_$c.log
done
python - <<'PY'
import csv, glob, collections, re, os
agg = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pm_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")[:80]
        if r["Counter_Name"] != c: continue
        agg[k][c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE": agg[k]["n"] += 1
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r01_pmc_traffic.csv")
with open(out, "w") as fh:
    fh.write("kernel,launches,fetch_kb_raw_per_launch,write_kb_raw_per_launch,hbm_bytes_per_launch_corrected\n")
    for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
        n = max(v["n"], 1)
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of 1024 B?  (guide: FETCH_SIZE = TCC_EA0_RDREQ x 64 B, in KB; x2 for 16-B/lane streaming reads on gfx950)
        fb, wb = v["FETCH_SIZE"] / n, v["WRITE_SIZE"] / n
        fh.write(f"{k},{n},{fb:.1f},{wb:.1f},{(2 * fb + wb) * 1024:.0f}\n")
print(open(out).read()[:3000])
PY
