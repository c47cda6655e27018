"""Join one process's launch log (E4T_LAUNCH_LOG: symbol | shape | algorithmic bytes | flops, in host launch order) with the
rocprofv3 outputs of the SAME command — kernel trace (durations) and the two PMC passes (FETCH_SIZE, WRITE_SIZE) — per kernel
SYMBOL and per SHAPE.  The k-th logged launch of a symbol is the k-th dispatch of that symbol (by Dispatch_Id) in every run:
the launch sequence of a training step is deterministic.

  python tools/roofline_report.py --trace DIR_OR_CSV --log LOG [--fetch DIR --fetch-log LOG --write DIR --write-log LOG] --out PREFIX [--commit SHA]

writes PREFIX_roofline_per_shape.csv (every (symbol, shape): launches, avg us, TFLOP/s, algorithmic GB/s, intensity, bound, roofline
fraction, HBM traffic per launch from the counters, traffic / algorithmic bytes) and PREFIX_pmc_traffic.csv (per symbol).
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB units; the x2 is gfx950's calibration for 16-B/lane streaming reads,
/opt/skills/guides/MI355X_MICROARCH.md "HBM"); sanity row: adamw_kernel must come out at 28 B x parameters."""
import argparse
import collections
import csv
import glob
import os
import re

MFMA_PEAK, HBM_PEAK = 2.5e15, 8.0e12


def norm(name):
    return re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "").strip()


def find_csv(path, suffix):
    if os.path.isfile(path):
        return path
    hits = glob.glob(os.path.join(path, "**", f"*{suffix}"), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {path}")
    return max(hits, key=os.path.getsize)


def read_log(path):
    per = collections.defaultdict(list)
    for line in open(path):
        parts = line.rstrip("\n").split("|")
        if len(parts) == 4:
            per[parts[0]].append((parts[1], float(parts[2]), float(parts[3])))
    return per


def dispatches(csv_path, value_col=None, counter=None):
    """-> {symbol: [value per dispatch, ordered by Dispatch_Id]}; value = duration in ns (kernel trace) or the counter"""
    per = collections.defaultdict(dict)
    with open(csv_path, newline="") as fh:
        for r in csv.DictReader(fh):
            if counter is not None:
                if r.get("Counter_Name") != counter:
                    continue
                v = float(r["Counter_Value"])
            else:
                v = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            d = per[norm(r["Kernel_Name"])]
            did = int(r["Dispatch_Id"])
            d[did] = d.get(did, 0.0) + v          # a counter may be reported per XCD / instance: sum
    return {k: [v for _, v in sorted(d.items())] for k, d in per.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True); ap.add_argument("--log", required=True)
    ap.add_argument("--fetch"); ap.add_argument("--fetch-log"); ap.add_argument("--write"); ap.add_argument("--write-log")
    ap.add_argument("--out", required=True); ap.add_argument("--commit", default="unknown")
    a = ap.parse_args()
    log = read_log(a.log)
    dur = dispatches(find_csv(a.trace, "kernel_trace.csv"))
    pmc = {}
    for name, d, lg in (("FETCH_SIZE", a.fetch, a.fetch_log), ("WRITE_SIZE", a.write, a.write_log)):
        if d:
            pmc[name] = (dispatches(find_csv(d, "counter_collection.csv"), counter=name), read_log(lg))
    rows, sym_rows, skipped = [], [], []
    for sym, launches in sorted(log.items()):
        t = dur.get(sym)
        if t is None or len(t) != len(launches):
            skipped.append((sym, len(launches), 0 if t is None else len(t)))
            continue
        agg = collections.OrderedDict()
        for (shape, nb, fl), ns in zip(launches, t):
            g = agg.setdefault(shape, dict(n=0, ns=0.0, nb=nb, fl=fl, fetch=0.0, write=0.0, nf=0, nw=0))
            g["n"] += 1; g["ns"] += ns
        for cname, key, cnt in (("FETCH_SIZE", "fetch", "nf"), ("WRITE_SIZE", "write", "nw")):
            if cname in pmc:
                vals, plog = pmc[cname]
                v, pl = vals.get(sym), plog.get(sym)
                if v is not None and pl is not None and len(v) == len(pl):
                    for (shape, _, _), x in zip(pl, v):
                        if shape in agg:
                            agg[shape][key] += x; agg[shape][cnt] += 1
        tot = dict(n=0, ns=0.0, nb=0.0, fl=0.0, hbm=0.0, hbm_n=0)
        for shape, g in agg.items():
            us = g["ns"] / g["n"] / 1e3
            sec = us * 1e-6
            inten = g["fl"] / g["nb"] if g["nb"] else 0.0
            bound = "mfma" if inten >= MFMA_PEAK / HBM_PEAK else "hbm"
            frac = (g["fl"] / sec / MFMA_PEAK) if bound == "mfma" else (g["nb"] / sec / HBM_PEAK)
            hbm = (2 * g["fetch"] / g["nf"] + g["write"] / g["nw"]) * 1024 if (g["nf"] and g["nw"]) else None
            rows.append([sym, shape, g["n"], f"{us:.2f}", f"{g['n'] * us / 1e3:.3f}", f"{g['fl'] / sec / 1e12:.1f}", f"{g['nb'] / sec / 1e9:.0f}", f"{inten:.1f}", bound,
                         f"{frac:.3f}", f"{g['nb']:.0f}", "" if hbm is None else f"{hbm:.0f}", "" if hbm is None or not g["nb"] else f"{hbm / g['nb']:.2f}"])
            tot["n"] += g["n"]; tot["ns"] += g["ns"]; tot["nb"] += g["nb"] * g["n"]; tot["fl"] += g["fl"] * g["n"]
            if hbm is not None:
                tot["hbm"] += hbm * g["n"]; tot["hbm_n"] += g["n"]
        f_raw = sum(g["fetch"] for g in agg.values()) / max(sum(g["nf"] for g in agg.values()), 1)
        w_raw = sum(g["write"] for g in agg.values()) / max(sum(g["nw"] for g in agg.values()), 1)
        sym_rows.append([sym, tot["n"], f"{f_raw:.1f}", f"{w_raw:.1f}", f"{(tot['hbm'] / tot['hbm_n']) if tot['hbm_n'] else 0:.0f}"])
    with open(a.out + "_roofline_per_shape.csv", "w", newline="") as fh:
        fh.write(f"# collected at commit {a.commit}; peaks: MFMA 2.5 PFLOP/s bf16 dense, HBM 8.0 TB/s; hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024\n")
        w = csv.writer(fh)
        w.writerow(["kernel", "shape", "launches", "avg_us", "total_ms", "tflops", "algorithmic_gbps", "intensity_flop_per_byte", "bound", "roofline_frac",
                    "algorithmic_bytes_per_launch", "hbm_bytes_per_launch_pmc", "traffic_over_algorithmic"])
        for r in sorted(rows, key=lambda r: -float(r[4])):
            w.writerow(r)
    with open(a.out + "_pmc_traffic.csv", "w") as fh:
        fh.write(f"# collected at commit {a.commit} (tools/profile_round.sh)\n")
        fh.write("kernel,launches,fetch_kb_raw_per_launch,write_kb_raw_per_launch,hbm_bytes_per_launch_corrected\n")
        for r in sorted(sym_rows, key=lambda r: -float(r[4]) * r[1]):
            fh.write(",".join(str(x) for x in r) + "\n")
    for s in skipped:
        print(f"skipped {s[0]}: {s[1]} logged launches vs {s[2]} traced dispatches")
    print(open(a.out + "_roofline_per_shape.csv").read()[:6000])


if __name__ == "__main__":
    main()
