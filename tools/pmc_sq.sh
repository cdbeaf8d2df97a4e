#!/bin/bash
# SQ counter passes over tools/quick_bench.py (development aid). Usage: pmc_sq.sh <tag> <quick_bench args...>
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"
timeout -s KILL 120 rocprofv3 --pmc $P1 --output-format csv -d "$OUT/p1" -- python3 "$ROOT/tools/quick_bench.py" "$@" > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $P2 --output-format csv -d "$OUT/p2" -- python3 "$ROOT/tools/quick_bench.py" "$@" > "$OUT/p2.log" 2>&1 || { tail -5 "$OUT/p2.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name=r["Kernel_Name"]
        for key in ("scatter_kernel","hist_kernel"):
            if key in name:
                agg[key+("<fuse>" if "Lb1EEE" in name.split("scatter_kernel")[-1][:14] else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    print(k)
    for name,v in sorted(c.items()):
        print("   %-24s n=%3d avg=%.4g"%(name,len(v),sum(v)/len(v)))
PY
