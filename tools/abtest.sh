#!/bin/bash
# Runs on the GPU box: for every variant library, first the parity tests selected by $KEXPR, then
# the quick_bench configurations.   KEXPR="kv or key_value" bash tools/abtest.sh "30 4 8 4" -- lib1.so lib2.so
set -o pipefail
cfgs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do cfgs+=("$1"); shift; done
shift
cp radx_amd/lib/libradx_hip.so /tmp/radx_ab_keep.so
trap 'cp /tmp/radx_ab_keep.so radx_amd/lib/libradx_hip.so' EXIT
for lib in "$@"; do
    cp "$lib" radx_amd/lib/libradx_hip.so
    if [ -n "$KEXPR" ]; then
        timeout -k 10 300 python -u -m pytest tests/test_gpu_sort.py tests/test_gpu_stress.py -m gpu -x -q -k "$KEXPR" --timeout 120 --timeout-method thread 2>&1 | tail -1 | sed "s|^|$(basename $lib) parity: |" || exit 1
    fi
done
for cfg in "${cfgs[@]}"; do
    for lib in "$@"; do
        cp "$lib" radx_amd/lib/libradx_hip.so
        echo "$(basename "$lib") [$cfg]: $(timeout -k 10 300 python -u tools/quick_bench.py $cfg 2>/dev/null | grep median)"
    done
done
