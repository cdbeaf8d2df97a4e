#!/bin/bash
# Per-kernel durations of a tools/quick_bench.py run (development aid). Usage: kstats.sh <quick_bench args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kstats
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/tools/quick_bench.py" "$@" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-70s calls %4s avg %10.1f us  min %10.1f  max %10.1f"%(r["Name"][:70],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
