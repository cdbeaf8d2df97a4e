"""Development timing loop: sort 2^LOG2N hashed keys a few times and print per-kernel times.

  python tools/quick_bench.py [log2n=28] [iters=5] [key_bytes=4] [value_bytes=0] [fuse=1]
"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import radx_amd

arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
log2n, iters, kb, vb, fuse = arg(1, 28), arg(2, 5), arg(3, 4), arg(4, 0), arg(5, 1)
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
import os
s.tune(bool(fuse), scan_kernels=os.environ.get('QB_SCAN_KERNELS', '0') == '1')  # QB_SCAN_KERNELS=1: no scan-free passes
s.profile(os.environ.get('QB_PROFILE', '1') == '1')  # QB_PROFILE=0: no per-launch events (for kernel traces)
passes = kb * 2
totals, scatters = [], []
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
    if it:
        totals.append(ms)
        scatters.append(t["scatter"][0] / max(1, t["scatter"][1]))
    print("n=2^%d iter %d: %.3f ms  %.2f Gkeys/s  alg %.0f GB/s (%.1f%% of 8 TB/s) | hist %.3f scan %.3f scatter %.3f ms/launch"
          % (log2n, it, ms, n / ms / 1e6, bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 80.0,
             t["hist"][0] / max(1, t["hist"][1]), t["scan"][0] / max(1, t["scan"][1]),
             t["scatter"][0] / max(1, t["scatter"][1])), flush=True)
if totals:
    totals.sort()
    scatters.sort()
    print("median of %d (first skipped): sort %.3f ms, scatter %.4f ms/launch; best sort %.3f ms"
          % (len(totals), totals[len(totals) // 2], scatters[len(scatters) // 2], totals[0]))
print("violations:", radx_amd.check_sorted(k, v))
