"""Development timing loop: sort 2^LOG2N hashed uint32 keys a few times, print per-kernel times."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import radx_amd

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
vb = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tpb = int(sys.argv[5]) if len(sys.argv) > 5 else 4
fuse = int(sys.argv[6]) if len(sys.argv) > 6 else 1
n = 1 << log2n
dev = torch.device("cuda:0")
kdt = torch.int32 if kb == 4 else torch.int64
k = torch.empty(n, dtype=kdt, device=dev)
swap = torch.empty_like(k)
v = vswap = None
if vb:
    v = torch.empty(n, dtype=torch.int32, device=dev)
    vswap = torch.empty_like(v)
s = radx_amd.Sorter(n, kb, vb)
s.tune(bool(fuse))
s.profile(True)
print("tiles_per_block", tpb, "fuse", fuse)
passes = kb * 2
for it in range(iters):
    if kb == 4:
        radx_amd.generate_u32(k, seed=it + 1)
    else:
        radx_amd.generate_u64(k, v, seed=it + 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s.sort(k, swap, v, vswap)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = s.kernel_times()
    bytes_alg = passes * n * 2 * (kb + vb)
    print("n=2^%d iter %d: %.3f ms  %.2f Gkeys/s  alg %.0f GB/s (%.1f%% of 8 TB/s) | hist %.3f scan %.3f scatter %.3f ms/launch"
          % (log2n, it, ms, n / ms / 1e6, bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 80.0,
             t["hist"][0] / max(1, t["hist"][1]), t["scan"][0] / max(1, t["scan"][1]),
             t["scatter"][0] / max(1, t["scatter"][1])), flush=True)
bad = radx_amd.check_sorted(k, v)
print("violations:", bad)
