"""Digest of tools/ab_sweep.sh / abtest_sweep.sh output: one row per size, one column per build
(ms per sort, the rounds separated by '/').   python tools/ab_table.py gpurun_out/ab.txt [type]"""
import re
import sys

rows, cur = {}, None
want = sys.argv[2] if len(sys.argv) > 2 else None
for line in open(sys.argv[1]):
    m = re.match(r"== (\S+) \(round (\d)\)", line)
    if m:
        cur = m.group(1)
        continue
    f = line.split()
    if cur and len(f) > 3 and f[0] in ("u32", "u32kv", "u64", "u64kv") and f[1].isdigit() and int(f[1]) > 0:
        if want and f[0] != want:
            continue
        rows.setdefault((f[0], int(f[1])), {}).setdefault(cur, []).append(float(f[2]))
libs = sorted({k for r in rows.values() for k in r})
print("type  log2n " + " ".join("%15s" % l.rsplit(".", 1)[0][-15:] for l in libs))
for key in sorted(rows):
    print("%-5s %5d " % key + " ".join("%15s" % "/".join("%.4f" % x for x in rows[key].get(l, [])) for l in libs))
