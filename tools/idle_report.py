"""From a rocprofv3 kernel trace of bench.py: how much of the timed region is the GPU idle (no kernel of any stream running), per
phase of the step — tells whether the step is kernel-bound or launch-bound.  usage: idle_report.py TRACE_DIR_OR_CSV [n_last_steps]"""
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
# steps are delimited by adamw_kernel launches
ad = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ad = ad[-(n + 1):]
t0, t1 = rows[ad[0]][1], rows[ad[-1]][1]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
busy, cur_s, cur_e = 0, None, None
gaps, pairs = [], []
last_name = None         # the kernel that finished last before a gap
for s, e, name, q in sel:
    if cur_e is None:
        cur_s, cur_e, last_name = s, e, name
    elif s <= cur_e:
        if e >= cur_e:
            cur_e, last_name = e, name
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, name))
        pairs.append((s - cur_e, last_name, name))
        cur_s, cur_e, last_name = s, e, name
busy += cur_e - cur_s
wall = t1 - t0
print(f"{n} steps: wall {wall / n / 1e6:.2f} ms/step, some kernel running {busy / n / 1e6:.2f} ms/step, idle {(wall - busy) / n / 1e6:.2f} ms/step ({100 * (wall - busy) / wall:.1f} %), "
      f"kernels/step {len(sel) / n:.0f}, sum of kernel durations {sum(e - s for s, e, _, _ in sel) / n / 1e6:.2f} ms/step")
by = collections.Counter()
cnt = collections.Counter()
for g, name in gaps:
    by[name] += g; cnt[name] += 1
# per stream (queue): busy time, and how much of it ran while a kernel of ANOTHER stream was running too
qs = collections.defaultdict(list)
for s_, e_, name, q in sel:
    qs[q].append((s_, e_))
def union(iv):
    out, cs, ce = [], None, None
    for a, b in sorted(iv):
        if ce is None or a > ce:
            if ce is not None: out.append((cs, ce))
            cs, ce = a, b
        else:
            ce = max(ce, b)
    if ce is not None: out.append((cs, ce))
    return out
def overlap(u1, u2):
    i = j = 0; tot = 0
    while i < len(u1) and j < len(u2):
        a, b = max(u1[i][0], u2[j][0]), min(u1[i][1], u2[j][1])
        if b > a: tot += b - a
        if u1[i][1] < u2[j][1]: i += 1
        else: j += 1
    return tot
un = {q: union(v) for q, v in qs.items()}
for q, u in sorted(un.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
    busy_q = sum(b - a for a, b in u)
    others = union([iv for q2, u2 in un.items() if q2 != q for iv in u2])
    print(f"stream {q}: {len(qs[q]) / n:.0f} kernels/step, busy {busy_q / n / 1e6:.2f} ms/step, of which concurrent with another stream {overlap(u, others) / n / 1e6:.2f} ms/step")
print("idle time in front of (kernel that ended the gap), top 25, per step:")
for name, g in by.most_common(25):
    print(f"  {g / n / 1e3:8.1f} us  {cnt[name] / n:6.1f} gaps  avg {g / cnt[name] / 1e3:6.1f} us  {name[:70]}")
big = sorted(gaps, reverse=True)[:10]
print("largest single gaps (us):", [(round(g / 1e3, 1), nm[:40]) for g, nm in big])

print("gaps over 100 us as (kernel that finished last) -> (kernel that ended the gap), per step:")
pc, pn = collections.Counter(), collections.Counter()
for g, a, b in pairs:
    if g > 100e3:
        pc[(a[:48], b[:48])] += g; pn[(a[:48], b[:48])] += 1
for (a, b), g in pc.most_common(12):
    print(f"  {g / n / 1e3:8.1f} us  {pn[(a, b)] / n:5.1f} gaps  {a}  ->  {b}")
