"""From a rocprofv3 kernel trace of bench.py: where is the MAIN stream (the one with most kernels) without a running kernel, and what
runs on the other stream(s) meanwhile — separates "waiting for the side stream / the host" from "kernels of two streams time-slicing".
usage: stream_gaps.py TRACE_DIR_OR_CSV [n_last_steps] [min_gap_us]"""
import collections, csv, glob, os, re, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = max(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
ad = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel") or r[2].startswith("adamw_rank")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ming = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
# step boundary = the LAST adamw-type launch of a step
ends = [i for k, i in enumerate(ad) if k + 1 == len(ad) or rows[ad[k + 1]][0] - rows[i][1] > 2_000_000]
ends = ends[-(n + 1):]
t0, t1 = rows[ends[0]][1], rows[ends[-1]][1]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
cnt = collections.Counter(r[3] for r in sel)
main = cnt.most_common(1)[0][0]
mk = [r for r in sel if r[3] == main]
sk = [r for r in sel if r[3] != main]
print(f"{n} steps, wall {(t1 - t0) / n / 1e6:.2f} ms/step; streams {dict(cnt)}; main = {main}")
gaps = []
for a, b in zip(mk, mk[1:]):
    g = b[0] - a[1]
    if g > 0:
        gaps.append((g, a, b))
tot = sum(g for g, _, _ in gaps)
print(f"main stream: busy {sum(e - s for s, e, _, _ in mk) / n / 1e6:.2f} ms/step, gaps {tot / n / 1e6:.2f} ms/step in {len(gaps) / n:.0f} gaps/step")
def side_busy(lo, hi):
    t = 0
    for s, e, _, _ in sk:
        if e <= lo or s >= hi:
            continue
        t += min(e, hi) - max(s, lo)
    return t
hist = collections.Counter()
for g, a, b in gaps:
    hist["<5us" if g < 5e3 else "<30us" if g < 30e3 else "<200us" if g < 200e3 else ">=200us"] += g
print("gap time by gap length (ms/step):", {k: round(v / n / 1e6, 2) for k, v in hist.items()})
big = sorted([x for x in gaps if x[0] >= ming * 1e3], key=lambda x: -x[0])
agg = collections.defaultdict(lambda: [0, 0, 0])
for g, a, b in big:
    k = (a[2][:44], b[2][:44])
    agg[k][0] += g; agg[k][1] += 1; agg[k][2] += side_busy(a[1], b[0])
print(f"gaps >= {ming} us, per step, as (main kernel before) -> (main kernel after): total us, count, side-stream busy us inside")
for k, (g, c, sb) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"  {g / n / 1e3:9.1f} us {c / n:5.1f} gaps  side busy {sb / n / 1e3:9.1f} us   {k[0]} -> {k[1]}")
