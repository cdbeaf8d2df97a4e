"""Soak: many full-size sorts back to back, every result checked (sortedness + multiset checksum).

  python tools/soak.py [log2n=30] [sorts=100] [key_bytes=4] [value_bytes=0]
"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import radx_amd

arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
log2n, sorts, kb, vb = arg(1, 30), arg(2, 100), arg(3, 4), arg(4, 0)
n = 1 << log2n
dev = torch.device("cuda:0")
k = torch.empty(n, dtype=torch.int32 if kb == 4 else torch.int64, device=dev)
swap = torch.empty_like(k)
v = torch.empty(n, dtype=torch.int32, device=dev) if vb else None
vswap = torch.empty_like(v) if vb else None
s = radx_amd.Sorter(n, kb, vb)
bad = 0
for it in range(sorts):
    if kb == 4:
        radx_amd.generate_u32(k, seed=it * 2654435761 % (1 << 32))
        if vb:
            v.copy_(torch.arange(n, dtype=torch.int32, device=dev))
    else:
        radx_amd.generate_u64(k, v, seed=it * 2654435761 % (1 << 32))
    before = radx_amd.checksum(k)
    s.sort(k, swap, v, vswap)
    ok = radx_amd.check_sorted(k, v if kb == 8 else None) == 0 and radx_amd.checksum(k) == before
    bad += not ok
    if not ok or it % 20 == 0:
        print("sort %d: %s" % (it, "ok" if ok else "WRONG"), flush=True)
print("done: %d sorts, %d wrong" % (sorts, bad))
sys.exit(1 if bad else 0)
