#!/bin/bash
# Runs on the GPU box: kernel trace of tools/quick_bench.py and the gap digest. Usage: trace_gaps.sh <tag> <quick_bench args...>
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gaps_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/tools/quick_bench.py" "$@" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 "$ROOT/tools/gaps.py" "$OUT" | tee "$OUT/gaps.txt"
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
