#!/bin/bash
# Runs on the GPU box: kernel-by-kernel trace (rocprofv3 --kernel-trace, no per-launch events) of
# tools/quick_bench.py at each given size, digested by tools/gaps.py.
#   bash tools/trace_sizes.sh <tag> "23 12" "24 12" "24 12 8 4"     (quick_bench argument strings)
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export QB_PROFILE=0
for cfg in "$@"; do
    name=$(echo "$cfg" | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$name" -- \
        python3 "$ROOT/tools/quick_bench.py" $cfg > "$OUT/$name.log" 2>&1 || { tail -5 "$OUT/$name.log"; exit 1; }
    echo "== quick_bench $cfg" | tee -a "$OUT/summary.txt"
    python3 "$ROOT/tools/gaps.py" "$OUT/$name" | tee -a "$OUT/summary.txt"
done
