"""Digest a rocprofv3 --kernel-trace CSV: per kernel name the busy time, and how much of the
exchange (RCCL) kernels' time was overlapped by sort kernels."""
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    short = ("rccl" if "rcclGeneric" in name or "nccl" in name.lower() else "scatter" if "scatter_kernel" in name else
             "hist" if "hist_kernel" in name else "scan" if "scan_" in name else
             "generate" if "generate" in name else "other")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, name, r.get("Queue_Id"), r.get("Stream_Id")))
ev.sort()
t0 = ev[0][0]
busy = {}
queues = {}
for s, e, k, _, q, st in ev:
    queues.setdefault(k, set()).add((q, st))
    busy.setdefault(k, [0, 0])
    busy[k][0] += e - s
    busy[k][1] += 1
print("kernel class: total ms, launches")
for k, (ns, c) in sorted(busy.items()):
    print("  %-9s %10.3f ms %6d   (hardware queue, stream): %s" % (k, ns / 1e6, c, sorted(queues[k])))


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def overlap(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if lo < hi:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


rccl = union([(s, e) for s, e, k, *_ in ev if k == "rccl"])
sort = union([(s, e) for s, e, k, *_ in ev if k in ("scatter", "hist", "scan")])
r_ms = sum(e - s for s, e in rccl) / 1e6
s_ms = sum(e - s for s, e in sort) / 1e6
print("rccl busy %.3f ms, sort busy %.3f ms, both at once %.3f ms" % (r_ms, s_ms, overlap(rccl, sort) / 1e6))
# the last step in detail: every rccl kernel and the sort kernels' span between them
last_gen = max(i for i, x in enumerate(ev) if x[2] == "generate") if any(x[2] == "generate" for x in ev) else 0
rk = [x for x in ev if x[2] == "rccl"]
print("last %d rccl kernels: start(ms) dur(ms)" % min(20, len(rk)))
for s, e, k, name, *_ in rk[-20:]:
    print("  %10.3f %8.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, name[:60]))
sc = [x for x in ev if x[2] == "scatter"]
print("last 10 scatter kernels: start(ms) dur(ms)")
for s, e, k, name, *_ in sc[-10:]:
    print("  %10.3f %8.3f" % ((s - t0) / 1e6, (e - s) / 1e6))
