"""Size sweep: whole-sort time, rate, fraction of the HBM roofline, launches per sort and
per-kernel time for n = 2^lo .. 2^hi, for uint32 keys and for uint64 keys + uint32 payload.

  python tools/size_sweep.py [--lo 16] [--hi 30] [--step 2] [--extra 23] [--types u32,u64kv]
                             [--out gpurun_out/size_sweep.json]

Timing: per repetition the input is regenerated on the device and the sort is bracketed by two
events; all repetitions of a size are enqueued without a host synchronisation in between, so the
GPU never idles (an idle gap costs the next ~10 ms of work 13 % in clocks).  The per-kernel
figures come from a second loop with the library's per-launch events on; they include the event
records and are for the split between kernels, not for the total.
"""
import argparse
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import radx_amd

PEAK_GBPS = 8000.0


def measure(log2n, kb, vb, budget_ms, scan_kernels=False):
    n = 1 << log2n
    dev = torch.device("cuda:0")
    k = torch.empty(n, dtype=torch.int32 if kb == 4 else torch.int64, device=dev)
    swap = torch.empty_like(k)
    v = vswap = None
    if vb:
        v = torch.empty(n, dtype=torch.int32, device=dev)
        vswap = torch.empty_like(v)
    s = radx_amd.Sorter(n, kb, vb)
    s.tune(True, scan_kernels=scan_kernels)

    def gen(seed):
        if kb == 4:
            radx_amd.generate_u32(k, seed=seed)
            if vb:
                v.copy_(torch.arange(n, dtype=torch.int32, device=dev))
        else:
            radx_amd.generate_u64(k, v, seed=seed)

    # warm-up: one sort for the estimate, then enough sorts to fill 50 ms
    gen(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s.sort(k, swap, v, vswap); e1.record()
    torch.cuda.synchronize()
    est = max(e0.elapsed_time(e1), 0.02)
    for i in range(min(2000, int(50.0 / est) + 1)):
        gen(2 + i)
        s.sort(k, swap, v, vswap)
    reps = max(7, min(200, int(budget_ms / est)))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        gen(100 + i)
        a.record(); s.sort(k, swap, v, vswap); b.record()
    torch.cuda.synchronize()
    times = sorted(a.elapsed_time(b) for a, b in ev)
    bad = radx_amd.check_sorted(k, v)
    med = times[len(times) // 2]

    # the split between kernels, per-launch events on
    s.profile(True)
    preps = max(3, min(20, reps))
    for i in range(preps):
        gen(500 + i)
        s.sort(k, swap, v, vswap)
    t = s.kernel_times()
    s.profile(False)
    passes = kb * 2
    alg = passes * n * 2 * (kb + vb)
    launches = sum(c for _, c in t.values()) / preps
    # the scan figure is per pass (one or two kernels bracketed together)
    row = {
        "log2n": log2n, "key_bytes": kb, "value_bytes": vb, "reps": reps,
        "ms_median": med, "ms_best": times[0],
        "gkeys_per_s": n / med / 1e6,
        "frac_of_roofline": alg / med / 1e6 / PEAK_GBPS,
        "timed_brackets_per_sort": launches,
        "hist_us": 1e3 * t["hist"][0] / max(1, t["hist"][1]),
        "scan_us_per_pass": 1e3 * t["scan"][0] / max(1, t["scan"][1]),
        "scatter_us": 1e3 * t["scatter"][0] / max(1, t["scatter"][1]),
        "scan_share_of_kernel_time": t["scan"][0] / max(1e-9, sum(m for m, _ in t.values())),
        "violations": bad,
    }
    s.close()
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=16)
    ap.add_argument("--hi", type=int, default=30)
    ap.add_argument("--step", type=int, default=2)
    ap.add_argument("--extra", type=str, default="23")
    ap.add_argument("--types", type=str, default="u32,u64kv")
    ap.add_argument("--budget-ms", type=float, default=150.0)
    ap.add_argument("--out", type=str, default="")
    ap.add_argument("--scan-kernels", action="store_true", help="run the scan kernels at every size (no scan-free passes)")
    a = ap.parse_args()
    sizes = sorted(set(range(a.lo, a.hi + 1, a.step)) | {int(x) for x in a.extra.split(",") if x})
    rows = []
    print("type    log2n      ms    Gkeys/s  %roofline | hist_us  scan_us/pass  scatter_us  scan_share | viol", flush=True)
    for ty in a.types.split(","):
        kb, vb = {"u32": (4, 0), "u64kv": (8, 4), "u32kv": (4, 4), "u64": (8, 0)}[ty]
        for l in sizes:
            r = measure(l, kb, vb, a.budget_ms, a.scan_kernels)
            r["scan_kernels_forced"] = a.scan_kernels
            r["type"] = ty
            rows.append(r)
            print("%-6s  %5d  %8.4f  %8.2f  %8.1f%% | %7.1f  %12.1f  %10.1f  %9.1f%% | %d"
                  % (ty, l, r["ms_median"], r["gkeys_per_s"], 100 * r["frac_of_roofline"], r["hist_us"],
                     r["scan_us_per_pass"], r["scatter_us"], 100 * r["scan_share_of_kernel_time"], r["violations"]),
                  flush=True)
            torch.cuda.empty_cache()
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
