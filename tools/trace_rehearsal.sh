#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the single-GPU rehearsal of the multi-GPU path,
# then a timeline digest (which kernels ran beside the exchange kernels) under gpurun_out/.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_rehearsal
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- \
    python3 "$ROOT/bench.py" --rehearse-exchange --steps 2 --warmup 1 --no-cpu --log2n ${1:-30} > "$OUT/bench.json" 2> "$OUT/trace.err"
python3 "$ROOT/tools/timeline.py" "$OUT/trace" > "$OUT/timeline.txt"
tail -40 "$OUT/timeline.txt"
