#!/bin/bash
# Runs on the GPU box: HBM traffic counters (FETCH_SIZE and WRITE_SIZE, one pass each) over a
# tools/quick_bench.py run, digested per kernel.  Usage: pmc_hbm.sh <tag> <quick_bench args...>
# FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request); WRITE_SIZE is exact; both in KiB.
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/hbm_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/tools/quick_bench.py" "$@" > "$OUT/fetch.log" 2>&1 || { tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/tools/quick_bench.py" "$@" > "$OUT/write.log" 2>&1 || { tail -5 "$OUT/write.log"; exit 1; }
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv,glob,sys,collections,re
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch","write"):
    for f in glob.glob(out+"/"+sub+"/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"])
            name=re.sub(r"\(.*","",name)
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-60s %8s %14s %14s %14s"%("kernel","launches","read MB","write MB","total MB"))
for k,c in sorted(agg.items()):
    rd=2*1024*sum(c.get("FETCH_SIZE",[0]))/max(1,len(c.get("FETCH_SIZE",[0])))/1e6
    wr=1024*sum(c.get("WRITE_SIZE",[0]))/max(1,len(c.get("WRITE_SIZE",[0])))/1e6
    print("%-60s %8d %14.1f %14.1f %14.1f"%(k[:60],len(c.get("FETCH_SIZE",[])),rd,wr,rd+wr))
PY
find "$OUT" -name "*.csv" -size +5M -delete
