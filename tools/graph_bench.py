"""Development timing: a sort enqueued kernel by kernel vs the same sort replayed from a HIP graph,
over a range of sizes.  python tools/graph_bench.py [log2n ...]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import radx_amd

dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [16, 20, 22, 24, 26, 28]
for log2n in sizes:
    n = 1 << log2n
    src = torch.empty(n, dtype=torch.int32, device=dev)
    radx_amd.generate_u32(src, seed=5)
    k = torch.empty_like(src)
    swap = torch.empty_like(src)
    s = radx_amd.Sorter(n)
    reps = 20

    def one():
        k.copy_(src)
        s.sort(k, swap)

    def copy_only():
        k.copy_(src)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    t_copy = timed(copy_only)
    t_stream = timed(one) - t_copy
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        one()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one()
    torch.cuda.synchronize()
    t_graph = timed(g.replay) - t_copy
    ok = radx_amd.check_sorted(k) == 0
    print("2^%d: stream %.4f ms  graph %.4f ms  (%.2f / %.2f Gkeys/s) sorted=%s"
          % (log2n, t_stream, t_graph, n / t_stream / 1e6, n / t_graph / 1e6, ok), flush=True)
    s.close()
