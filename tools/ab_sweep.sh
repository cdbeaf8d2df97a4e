#!/bin/bash
# Runs on the GPU box: A/B several builds of the engine on the same box with tools/size_sweep.py.
#   bash tools/ab_sweep.sh "--lo 20 --hi 24 --extra 23 --types u32" lib_a.so lib_b.so ...
# The libraries are built beforehand in-tree under other names (they travel with the snapshot).
set -o pipefail
args=$1; shift
cp radx_amd/lib/libradx_hip.so /tmp/radx_ab_keep.so
trap 'cp /tmp/radx_ab_keep.so radx_amd/lib/libradx_hip.so' EXIT
for round in 1 2; do
    for lib in "$@"; do
        cp "$lib" radx_amd/lib/libradx_hip.so
        echo "== $(basename "$lib") (round $round)"
        timeout -k 10 300 python -u tools/size_sweep.py $args 2>/dev/null | grep -v "^type" || exit 1
    done
done
