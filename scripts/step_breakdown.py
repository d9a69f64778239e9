"""Per-kernel totals of the LAST time step in a rocprofv3 kernel trace (the --stats table mixes in the start-up).
usage: step_breakdown.py <..._kernel_trace.csv> [marker-substring | --tail-ms MS]
The step is taken to start at the last kernel whose name contains the marker (default: the first lambda of ffm_plume_step), or to be
the last MS milliseconds of the trace (scripts/class_layer_probe.py writes that figure)."""
import csv, sys, collections
path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "ffm_plume_step::{lambda(long)#1}"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if marker == "--tail-ms":
    t0 = rows[-1][1] - float(sys.argv[3]) * 1e6
    starts = [i for i, r in enumerate(rows) if r[0] >= t0][:1]
else:
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
if not starts:
    sys.exit("marker not found")
i0 = starts[-1]
step = rows[i0:]
wall = step[-1][1] - step[0][0]
busy = 0; tot = collections.Counter(); cnt = collections.Counter(); last_end = step[0][0]; gap = 0
gapAfter = collections.Counter(); gapCnt = collections.Counter(); prev = None
for s, e, n in step:
    tot[n] += e - s; cnt[n] += 1; busy += e - s
    if s > last_end:
        gap += s - last_end
        if prev is not None and s - last_end > 2000: gapAfter[prev] += s - last_end; gapCnt[prev] += 1      # idle time by the kernel that ran before it
    last_end = max(last_end, e); prev = n
print("last step: %d kernels, wall %.2f ms, kernel time %.2f ms, idle gaps %.2f ms" % (len(step), wall / 1e6, busy / 1e6, gap / 1e6))
for n, t in tot.most_common(int(__import__('os').environ.get('TOP', '45'))):
    print("%8.2f ms %6d x %8.1f us  %s" % (t / 1e6, cnt[n], t / cnt[n] / 1e3, n[:110]))
if __import__('os').environ.get('GAPS'):
    print("idle time (gaps > 2 us) by the kernel that ran before the gap:")
    for n, t in gapAfter.most_common(int(__import__('os').environ['GAPS'])):
        print("%8.2f ms %6d x %8.1f us  after %s" % (t / 1e6, gapCnt[n], t / gapCnt[n] / 1e3, n[:100]))
