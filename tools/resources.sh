#!/bin/bash
# Per-kernel registers, LDS and occupancy of the engine's code object (no GPU needed).
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Rpass-analysis=kernel-resource-usage -c -o /dev/null \
    ${1:-radx_amd/csrc/radx_hip.hip} 2>&1 | python3 -c '
import re, sys, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: (?:[^:]*:\d+:\d+: )?\s*([A-Za-z][A-Za-z /\[\]]*?):\s+(\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else: cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    print("%-60s VGPR %3s AGPR %3s SGPR %3s spill %s LDS %6s occ %s" % (name[:60], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
