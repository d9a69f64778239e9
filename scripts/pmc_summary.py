"""Per-kernel HBM-side bytes of the last time step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units).
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [nCells]
FETCH_SIZE is scaled by the calibration kernel k_reduce1<0> (reads exactly 8*nCells bytes)."""
import csv, sys, collections
def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows
def last_step(rows):
    marker = "ffm_plume_step::{lambda(long)#1}"
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    return rows[starts[-1]:] if starts else rows
fe, wr = load(sys.argv[1]), load(sys.argv[2])
N = int(sys.argv[3]) if len(sys.argv) > 3 else None
cal = [r[3] for r in fe if "k_reduce1<0>" in r[2]]
scale = 1.0
if cal and N:
    scale = (8.0 * N / 1024.0) / (sum(cal[-3:]) / len(cal[-3:]))
print("FETCH_SIZE calibration factor %.3f (k_reduce1<0>: %s KiB counted for %s KiB read)" % (scale, cal[-1] if cal else None, 8.0 * N / 1024 if N else None))
agg = collections.OrderedDict()
for rows, key in ((last_step(fe), "f"), (last_step(wr), "w")):
    for s, e, n, v in rows:
        a = agg.setdefault(n, {"f": 0.0, "w": 0.0, "nf": 0, "nw": 0, "t": 0})
        a[key] += v; a["n" + key] += 1
        if key == "f": a["t"] += e - s
print("%-70s %6s %10s %10s %10s %9s" % ("kernel (last step)", "calls", "read GB", "write GB", "GB/call", "TB/s"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"])[:40]:
    if not a["nf"]: continue
    rd = a["f"] * scale * 1024 / 1e9; wt = a["w"] * 1024 / 1e9 * (a["nf"] / max(a["nw"], 1))
    print("%-70s %6d %10.2f %10.2f %10.3f %9.2f" % (n[:70], a["nf"], rd, wt, (rd + wt) / a["nf"], (rd + wt) / (a["t"] / 1e9) / 1e3))
