#!/bin/bash
# Runs on the GPU box (via gpurun): A/B two or more builds of the engine on the same box, back to
# back, with tools/quick_bench.py (median of the sorts after the first).
#   bash tools/ab.sh "30 13" "28 13" -- radx_amd/lib/libradx_hip_base.so radx_amd/lib/libradx_hip_new.so
# Each quoted string is one quick_bench configuration (log2n iters key_bytes value_bytes fuse);
# the libraries are built beforehand in-tree under other names (they travel with the snapshot).
set -e -o pipefail
cfgs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do cfgs+=("$1"); shift; done
shift
cp radx_amd/lib/libradx_hip.so /tmp/radx_ab_keep.so
trap 'cp /tmp/radx_ab_keep.so radx_amd/lib/libradx_hip.so' EXIT
for cfg in "${cfgs[@]}"; do
    for lib in "$@"; do
        cp "$lib" radx_amd/lib/libradx_hip.so
        echo "$(basename "$lib") [$cfg]: $(timeout -k 10 300 python -u tools/quick_bench.py $cfg | grep median)"
    done
done
