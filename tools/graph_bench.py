"""Launch cost of small and mid-size sorts: the same (generate keys, sort) step enqueued call by
call and replayed from a captured HIP graph.

  python tools/graph_bench.py [--lo 10] [--hi 24] [--step 2] [--types u32,u64kv]

Per size: K steps back to back without host synchronisation, bracketed by two events; the
generate-only loop is timed the same way and subtracted, so the figure is the sort alone.  The
library's calls only enqueue kernels on the given stream, so a sort is capturable as it is.
"""
import argparse
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import radx_amd


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def measure(log2n, kb, vb, reps):
    n = 1 << log2n
    dev = torch.device("cuda:0")
    k = torch.empty(n, dtype=torch.int32 if kb == 4 else torch.int64, device=dev)
    swap = torch.empty_like(k)
    v = vswap = None
    if vb:
        v = torch.empty(n, dtype=torch.int32, device=dev)
        vswap = torch.empty_like(v)
    s = radx_amd.Sorter(n, kb, vb)

    def gen():
        if kb == 4:
            radx_amd.generate_u32(k, seed=7)
            if vb:
                radx_amd.generate_u32(v, seed=9)
        else:
            radx_amd.generate_u64(k, v, seed=7)

    def step():
        gen()
        s.sort(k, swap, v, vswap)

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g_gen, g_step = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g_gen, stream=side):
            gen()
        with torch.cuda.graph(g_step, stream=side):
            step()
    torch.cuda.synchronize()
    # warm the clocks
    for _ in range(max(20, int(2e5 / max(n, 1)) // 100)):
        g_step.replay()
    eager = timed(step, reps) - timed(gen, reps)
    graph = timed(g_step.replay, reps) - timed(g_gen.replay, reps)
    g_step.replay()
    bad = radx_amd.check_sorted(k, v)
    s.close()
    return eager, graph, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=10)
    ap.add_argument("--hi", type=int, default=24)
    ap.add_argument("--step", type=int, default=2)
    ap.add_argument("--types", type=str, default="u32,u64kv")
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    shapes = {"u32": (4, 0), "u32kv": (4, 4), "u64": (8, 0), "u64kv": (8, 4)}
    print("type    log2n   call-by-call us   graph replay us   ratio | viol", flush=True)
    for t in a.types.split(","):
        kb, vb = shapes[t]
        for l in range(a.lo, a.hi + 1, a.step):
            eager, graph, bad = measure(l, kb, vb, a.reps)
            print(f"{t:6s} {l:6d} {1e3 * eager:17.1f} {1e3 * graph:17.1f} {eager / graph:7.2f} | {bad}", flush=True)


if __name__ == "__main__":
    main()
