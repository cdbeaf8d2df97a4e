#!/bin/bash
# The single-GPU benchmark configurations through tools/quick_bench.py; output collected in gpurun_out/config_sweep.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/config_sweep.txt
: > "$OUT"
for args in "28 5 4 0" "30 5 4 0" "30 3 8 4" "30 3 8 0" "30 3 4 4"; do
  echo "== quick_bench.py $args (log2n iters key_bytes value_bytes)" >> "$OUT"
  timeout -k 10 300 python3 -u "$ROOT/tools/quick_bench.py" $args >> "$OUT" 2>&1 || exit 1
done
cat "$OUT"
