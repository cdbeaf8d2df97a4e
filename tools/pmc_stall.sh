#!/bin/bash
# Runs on the GPU box: memory-path stall counters over a tools/quick_bench.py run, per kernel.
# Usage: pmc_stall.sh <tag> <quick_bench args...>
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stall_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_avr SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
timeout -s KILL 150 rocprofv3 --pmc $P1 --output-format csv -d "$OUT/p1" -- python3 "$ROOT/tools/quick_bench.py" "$@" > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv,glob,sys,collections,re
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); name=re.sub(r"\(.*","",name)
        if "scatter" in name or "hist" in name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in sorted(agg.items()):
    print(k)
    for name,v in sorted(c.items()):
        print("   %-40s n=%3d avg=%.5g"%(name,len(v),sum(v)/len(v)))
PY
find "$OUT" -name "*.csv" -size +5M -delete
