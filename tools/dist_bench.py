"""Development timing: sort throughput by key distribution (uint32, 2^LOG2N keys, median of 7).

  python tools/dist_bench.py [log2n=28] [fuse=1] [key_bytes=4] [value_bytes=0]

With key_bytes=8 every 32-bit pattern is widened to 64 bits (pattern * 0x100000001), which keeps
its digit structure in both halves.
"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import radx_amd

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
fuse = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
vb = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = 1 << log2n
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)


def hashed():
    k = torch.empty(n, dtype=torch.int32, device=dev)
    radx_amd.generate_u32(k, seed=3)
    return k


def skewed(frac=0.9):  # this fraction of the keys is one value
    k = hashed()
    m = torch.rand(n, device=dev, generator=g) < frac
    k[m] = 0x12345678
    return k


cases = {
    "uniform (hashed)": hashed,
    "already sorted": lambda: torch.arange(n, dtype=torch.int64, device=dev).to(torch.int32),
    "reverse sorted": lambda: torch.arange(n - 1, -1, -1, dtype=torch.int64, device=dev).to(torch.int32),
    "all equal": lambda: torch.full((n,), 7, dtype=torch.int32, device=dev),
    "16 distinct values": lambda: torch.randint(0, 16, (n,), dtype=torch.int32, device=dev, generator=g) * 0x01010101,
    "50 % one value": lambda: skewed(0.5),
    "90 % one value": skewed,
    "99 % one value": lambda: skewed(0.99),
    "99.9 % one value": lambda: skewed(0.999),
    "99.99 % one value": lambda: skewed(0.9999),
    "2 values, 50/50": lambda: torch.where(torch.rand(n, device=dev, generator=g) < 0.5, 0x12345678, 0x12345668).to(torch.int32),
    "4 values + 10 % uniform": lambda: torch.where(torch.rand(n, device=dev, generator=g) < 0.1, hashed(),
                                                   (torch.randint(0, 4, (n,), dtype=torch.int32, device=dev, generator=g) * 0x00110011 + 0x12005600)),
    "small range (0..999)": lambda: torch.randint(0, 1000, (n,), dtype=torch.int32, device=dev, generator=g),
    "only the top digit varies": lambda: torch.randint(0, 16, (n,), dtype=torch.int32, device=dev, generator=g) << 27,
}
s = radx_amd.Sorter(n, kb, vb)
s.tune(bool(fuse))
s.profile(True)
swap = torch.empty(n, dtype=torch.int32 if kb == 4 else torch.int64, device=dev)
vals = torch.arange(n, dtype=torch.int64, device=dev).to(torch.int32) if vb else None
v = torch.empty_like(vals) if vb else None
vswap = torch.empty_like(vals) if vb else None
for name, make in cases.items():
    src = make()
    if kb == 8:
        src = (src.to(torch.int64) & 0xFFFFFFFF) * 0x100000001
    k = torch.empty_like(src)
    times = []
    for it in range(8):
        k.copy_(src)
        if vb:
            v.copy_(vals)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s.sort(k, swap, v, vswap)
        e1.record()
        torch.cuda.synchronize()
        if it:
            times.append(e0.elapsed_time(e1))
    times.sort()
    ms = times[len(times) // 2]
    ok = radx_amd.check_sorted(k, v) == 0 and radx_amd.checksum(k) == radx_amd.checksum(src)
    kt = s.kernel_times()
    print("%-28s %8.3f ms  %7.2f Gkeys/s  %s | hist %.3f scan %.3f scatter %.3f ms/launch"
          % (name, ms, n / ms / 1e6, "ok" if ok else "WRONG", kt["hist"][0] / max(1, kt["hist"][1]),
             kt["scan"][0] / max(1, kt["scan"][1]), kt["scatter"][0] / max(1, kt["scatter"][1])), flush=True)
    del src, k
