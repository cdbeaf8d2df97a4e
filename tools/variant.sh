#!/bin/bash
# Build a named variant of the engine with extra -D flags and print its scatter kernels' resources.
#   tools/variant.sh <name> [flags...]   ->  radx_amd/lib/libradx_hip_<name>.so
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wall -Wno-unused-function "$@" -shared \
    -o radx_amd/lib/libradx_hip_$name.so radx_amd/csrc/radx_hip.hip -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys
cur=None
for ln in sys.stdin:
    m=re.search(r"Function Name: (\S+)",ln)
    if m: cur=m.group(1); d={}
    for key,pat in (("vgpr",r" VGPRs: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("lds",r"LDS Size \[bytes/block\]: (\d+)"),("spill",r"ScratchSize \[bytes/lane\]: (\d+)")):
        m=re.search(pat,ln)
        if m and cur: d[key]=m.group(1)
    if cur and "LDS Size" in ln and "scatter_kernel" in cur:
        t=re.search(r"scatter_kernelI([jm])Lb([01])ELb([01])",cur)
        print("  %s scatter<%s,v=%s,fuse=%s> vgpr %s occ %s lds %s spill %s"%(sys.argv[1],"u32" if t.group(1)=="j" else "u64",t.group(2),t.group(3),d.get("vgpr"),d.get("occ"),d.get("lds"),d.get("spill")))
' $name
