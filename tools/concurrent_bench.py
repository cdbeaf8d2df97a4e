"""Development timing: the same number of sorts issued on one stream, and spread over several
streams with one sorter each (kernels of different sorts then share the GPU).  Shows whether the
scan-free passes' publishing workgroups still run in time when another sort's workgroups compete
for the CUs: a workgroup that misses its words waits its bounded looks and sums the table itself,
which is correct but slow, so a large slowdown here would mean that fallback is being taken.

  python tools/concurrent_bench.py [log2n=22] [streams=2] [sorts=64] [scan_kernels=0]
"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import radx_amd

arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
log2n, nstreams, sorts, scan_kernels = arg(1, 22), arg(2, 2), arg(3, 64), arg(4, 0)
n = 1 << log2n
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
sorters = [radx_amd.Sorter(n) for _ in range(nstreams)]
for s in sorters:
    s.tune(True, scan_kernels=bool(scan_kernels))
keys = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(nstreams)]
swaps = [torch.empty_like(k) for k in keys]


def run(active):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(sorts):
        j = i % active
        with torch.cuda.stream(streams[j]):
            radx_amd.generate_u32(keys[j], seed=i + 1, stream=streams[j])
            sorters[j].sort(keys[j], swaps[j], stream=streams[j])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    run(1); run(nstreams)
one = min(run(1) for _ in range(5))
many = min(run(nstreams) for _ in range(5))
bad = sum(radx_amd.check_sorted(k) for k in keys)
print("2^%d keys, %d sorts (incl. key generation): one stream %.3f ms, %d streams %.3f ms (x%.2f), violations %d"
      % (log2n, sorts, one, nstreams, many, one / many, bad))
