"""Summarise a rocprofv3 --kernel-trace CSV: steady-state busy fraction, gaps, per-kernel time."""
import collections
import csv
import glob
import statistics
import sys

f = (sys.argv[1:] or glob.glob('gpurun_out/prof/*/*kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
sub = rows[int(n * 0.6):int(n * 0.95)]
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
busy = sum(dur(r) for r in sub)
span = int(sub[-1]['End_Timestamp']) - int(sub[0]['Start_Timestamp'])
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(sub, sub[1:])]
print("kernels", len(sub), "span_us %.1f busy_us %.1f busy_frac %.3f" % (span / 1e3, busy / 1e3, busy / span))
print("gap ns: median %.0f mean %.0f max %d  overlapped %d" % (
    statistics.median(gaps), sum(gaps) / len(gaps), max(gaps), sum(1 for g in gaps if g < 0)))
agg = collections.defaultdict(lambda: [0, 0])
for r in sub:
    k = r['Kernel_Name'][:72]
    agg[k][0] += dur(r)
    agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%9.1f us  %5d calls  %7.2f us avg  %s" % (t / 1e3, c, t / 1e3 / c, k))
