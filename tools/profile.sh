#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace statistics of the default bench command, then
# the two HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass) on a short run.
# Results land under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py digests them into profiles/.
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats pass done" 
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-extras > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu --no-extras > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
echo "write pass done"
find "$OUT" -name "*.csv" | head -20
