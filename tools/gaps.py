"""Digest a rocprofv3 --kernel-trace CSV of a tools/quick_bench.py run: for every sort (the kernels
between two generate kernels) the busy time per kernel class, the idle gaps between consecutive
kernels and the span; then the last sort kernel by kernel.

  python tools/gaps.py <rocprofv3 output dir>
"""
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
ev = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    short = ("scatter" if "scatter_kernel" in name else "hist" if "hist_kernel" in name else
             "scan_rows" if "scan_rows" in name else "scan_groups" if "scan_groups" in name else
             "generate" if "generate" in name else "other")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
ev.sort()
sorts, cur = [], []
for e in ev:
    if e[2] in ("generate", "other"):
        if cur:
            sorts.append(cur)
        cur = []
    else:
        cur.append(e)
if cur:
    sorts.append(cur)
print("sort  span_ms  busy_ms  gaps_ms  | hist  scan_rows  scan_groups  scatter (ms, summed) | launches")
for i, s in enumerate(sorts):
    span = (s[-1][1] - s[0][0]) / 1e6
    busy = sum(e - b for b, e, _ in s) / 1e6
    per = {}
    for b, e, k in s:
        per[k] = per.get(k, 0) + (e - b) / 1e6
    print("%4d  %7.3f  %7.3f  %7.3f  | %5.3f  %5.3f  %5.3f  %6.3f | %d"
          % (i, span, busy, span - busy, per.get("hist", 0), per.get("scan_rows", 0),
             per.get("scan_groups", 0), per.get("scatter", 0), len(s)))
if sorts:
    s = sorts[-1]
    print("last sort, kernel by kernel (us): gap before, duration")
    prev = None
    for b, e, k in s:
        print("  %-12s gap %7.2f  dur %9.2f" % (k, (b - prev) / 1e3 if prev else 0.0, (e - b) / 1e3))
        prev = e
