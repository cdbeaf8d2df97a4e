#!/bin/bash
# Runs on the GPU box: for every variant library first the per-pass parity tests (a variant that
# fails them is not timed), then tools/size_sweep.py for all of them, two rounds.
#   bash tools/abtest_sweep.sh "--lo 20 --hi 25 --step 1 --extra '' --types u32" lib_a.so lib_b.so ...
set -o pipefail
args=$1; shift
cp radx_amd/lib/libradx_hip.so /tmp/radx_ab_keep.so
trap 'cp /tmp/radx_ab_keep.so radx_amd/lib/libradx_hip.so' EXIT
good=()
for lib in "$@"; do
    cp "$lib" radx_amd/lib/libradx_hip.so
    if timeout -k 10 400 python -u -m pytest tests/test_gpu_sort.py tests/test_gpu_stress.py -m gpu -x -q --timeout 120 --timeout-method thread \
        -k "every_pass or skewed or limit or stress or tuning or key_value" > /tmp/parity.log 2>&1; then
        echo "$(basename "$lib") parity: $(tail -1 /tmp/parity.log)"; good+=("$lib")
    else
        echo "$(basename "$lib") parity FAILED: $(tail -3 /tmp/parity.log)"; exit 1
    fi
done
for round in 1 2; do
    for lib in "${good[@]}"; do
        cp "$lib" radx_amd/lib/libradx_hip.so
        echo "== $(basename "$lib") (round $round)"
        timeout -k 10 300 python -u tools/size_sweep.py $args 2>/dev/null | grep -v "^type" || exit 1
    done
done
