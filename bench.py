#!/usr/bin/env python3
"""Headline benchmark: Gkeys/s of the uint32 LSD radix sort on 1/2/4/8 MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n L]

One step = one complete sort (all 8 four-bit passes) of one batch of 2^L uniform uint32 keys
per GPU that is already resident in HBM.  N > 1 runs one rank per GPU: either this script is
started once per rank by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment), or — started plainly as `python bench.py --gpus N` — it starts the N ranks itself as
child processes before anything touches a GPU, relays rank 0's JSON line and exits with the worst
child status.  A step is then the split pass on the top digit + the all-to-all-v exchange over
RCCL + the 7 local passes, pipelined per sub-bucket, and the per-GPU batch stays 2^L (weak
scaling).  Every step sorts a DIFFERENT pre-generated batch, so no step ever sees already-sorted
input.  Rank 0 prints one JSON line.

The timed region runs with per-kernel event profiling OFF; a second, untimed loop over freshly
generated batches collects the per-launch kernel times behind `roofline`.
"""
import argparse
import ctypes
import datetime
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=30, help="log2 of keys per GPU (30 = README headline size)")
    ap.add_argument("--cpu-log2n", type=int, default=29, help="log2 of the CPU-baseline sample (2^29: about 10 s of host work)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the other single-GPU configs")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="on one GPU, run the N > 1 code path (RCCL process group of one rank, split + "
                         "16 all-to-all-v's + per-region local passes) instead of the plain sort")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 path (nccl = RCCL; gloo only to rehearse the "
                         "multi-rank code on a box whose ranks must share one GPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (needs --backend gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--print-launch", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE in the environment: print the child "
                         "commands and rank environments as JSON instead of starting them")
    return ap.parse_args()


RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
# Set by this script's own launcher: rank 0 opens its store on a port the system picks and writes
# the number here, the other ranks read it.  (A port probed by the launcher could be taken by
# another process before rank 0 binds it.)
PORT_FILE_ENV = "RADX_BENCH_PORT_FILE"


def rendezvous_store(path, rank, world, timeout):
    """The process group's store when this script launched the ranks itself."""
    from torch.distributed import TCPStore
    if rank == 0:
        store = TCPStore("127.0.0.1", 0, world, is_master=True, timeout=timeout, wait_for_workers=False)
        with open(path + ".tmp", "w") as f:
            f.write(str(store.port))
        os.replace(path + ".tmp", path)
        return store
    deadline = time.monotonic() + timeout.total_seconds()
    while not os.path.exists(path):
        if time.monotonic() > deadline:
            raise RuntimeError("rank 0 never published its store's port in %s" % path)
        time.sleep(0.05)
    with open(path) as f:
        port = int(f.read())
    return TCPStore("127.0.0.1", port, world, is_master=False, timeout=timeout)


def launch_plan(n_ranks, argv, port=None):
    """Command and per-rank environment overrides for n_ranks children of this script."""
    if port is None:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    cmd = [sys.executable, "-u", os.path.abspath(__file__)] + [a for a in argv if a != "--print-launch"]
    envs = [{"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
             "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)} for r in range(n_ranks)]
    return cmd, envs


def launch_ranks(n_ranks, argv):
    """Start one child per rank (this process never initialises a GPU and never execs), relay rank
    0's stdout, wait for all, and return the worst exit status.  When a rank fails the others get
    a grace period to notice (their collectives error out) and are then terminated by PID."""
    import tempfile
    cmd, envs = launch_plan(n_ranks, argv)
    port_file = os.path.join(tempfile.mkdtemp(prefix="radx_bench_"), "port")
    procs = []
    for r, over in enumerate(envs):
        env = dict(os.environ)
        env.update(over)
        env[PORT_FILE_ENV] = port_file
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0 owns stdout (the JSON line); the other ranks' stdout joins stderr
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else sys.stderr))
    worst, failed_at = 0, None
    live = set(range(n_ranks))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                print("bench.py: rank %d exited with status %d" % (r, rc), file=sys.stderr)
                worst = worst or (rc if rc > 0 else 128 - rc)
                failed_at = failed_at or time.monotonic()
        if failed_at and live and time.monotonic() - failed_at > 30.0:
            for r in sorted(live):
                procs[r].terminate()
            deadline = time.monotonic() + 10.0
            for r in sorted(live):
                try:
                    procs[r].wait(max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        time.sleep(0.05)
    try:
        if os.path.exists(port_file):
            os.remove(port_file)
        os.rmdir(os.path.dirname(port_file))
    except OSError:
        pass
    return worst


def cpu_baseline(log2n):
    """Parallel std::sort of a bounded sample on the host cores (oracle/cpu_sort.cpp), next to the
    reference's own comparator call std::stable_sort(std::execution::par, ...) (src/test/sort.cpp:218-219)."""
    import numpy as np
    import oracle
    n = 1 << log2n
    keys = oracle.generate_u32(n, 12345)
    best = None
    for _ in range(2):
        k = keys.copy()
        t0 = time.perf_counter()
        oracle.cpu.radx_cpu_parallel_sort_u32(k.ctypes.data_as(ctypes.c_void_p), n)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert np.all(k[1:] >= k[:-1])
    m = 1 << min(log2n, 25)
    k = keys[:m].copy()
    t0 = time.perf_counter()
    oracle.cpu.radx_cpu_stable_sort_u32(k.ctypes.data_as(ctypes.c_void_p), m)
    stable_dt = time.perf_counter() - t0
    return {
        "value": round(n / best / 1e9, 4), "unit": "Gkeys/s", "cores": int(oracle.cpu.radx_cpu_threads()),
        "kind": "port",
        "sample": "2^%d hashed uint32 keys, __gnu_parallel::sort (OpenMP), best of 2: %.3f s" % (log2n, best),
        "reference_call": {
            "what": "std::stable_sort(std::execution::par) on 2^%d keys, PSTL backend '%s'"
                    % (min(log2n, 25), oracle.cpu.radx_cpu_pstl_backend().decode()),
            "value": round(m / stable_dt / 1e9, 4), "unit": "Gkeys/s"},
    }


def other_configs(dev):
    """The other single-GPU benchmark configurations, timed the same way (every step its own
    pre-generated batch, wall clock between synchronisations).  Reported beside the headline
    line, never as `value`."""
    import torch
    import radx_amd
    res = {}
    for name, log2n, kb, vb, steps in (("2^23 uint32 keys", 23, 4, 0, 50),   # the reference harness's own size (src/test/sort.hpp:184)
                                       ("2^28 uint32 keys", 28, 4, 0, 10),
                                       ("2^30 uint64 keys + uint32 payload", 30, 8, 4, 3),
                                       ("2^30 uint32 keys + uint32 payload", 30, 4, 4, 3),
                                       ("2^30 uint64 keys", 30, 8, 0, 3)):
        n = 1 << log2n
        kdt = torch.int32 if kb == 4 else torch.int64
        keys = [torch.empty(n, dtype=kdt, device=dev) for _ in range(steps + 1)]
        vals = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(steps + 1)] if vb else None
        swap = torch.empty(n, dtype=kdt, device=dev)
        vswap = torch.empty(n, dtype=torch.int32, device=dev) if vb else None
        index = torch.arange(n, dtype=torch.int32, device=dev) if kb == 4 and vb else None

        def generate(i, seed):
            if kb == 4:
                radx_amd.generate_u32(keys[i], seed=seed)
                if vb:
                    vals[i].copy_(index)  # payload = original index
            else:
                radx_amd.generate_u64(keys[i], vals[i] if vb else None, seed=seed)

        for i in range(len(keys)):
            generate(i, 1000 + i)
        s = radx_amd.Sorter(n, kb, vb)
        before = radx_amd.checksum(keys[steps])  # multiset of the last timed batch, before its sort
        # warm-up: a few sorts of a scratch batch (regenerated each time), at least ~50 ms of work,
        # so the timed sorts do not include the clock ramp after the idle gap of the set-up above
        for w in range(3 if log2n >= 30 else 16 if log2n >= 26 else 300):
            generate(0, 500 + w)
            s.sort(keys[0], swap, vals[0] if vb else None, vswap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            s.sort(keys[i], swap, vals[i] if vb else None, vswap)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # sorted (with a payload: and stable, the payload being the original index), same multiset
        ok = radx_amd.check_sorted(keys[steps], vals[steps] if vb else None) == 0
        ok = ok and radx_amd.checksum(keys[steps]) == before
        if vb:
            # every payload still names the input slot its key came from: regenerate the batch
            # into the first slot and compare regenerated[payload[i]] with the sorted key i
            generate(0, 1000 + steps)
            ok = ok and bool(torch.equal(keys[0][vals[steps].long()], keys[steps]))
        # per-launch kernel times of this shape: two more sorts of fresh batches with every launch
        # bracketed by events; the dominant kernel is the scatter, one launch = one pass over n elements
        s.profile(True)
        for w in range(2 if log2n >= 28 else 10):
            generate(0, 700 + w)
            s.sort(keys[0], swap, vals[0] if vb else None, vswap)
        kt = s.kernel_times()
        s.profile(False)
        per_launch = {k: round(v[0] / max(1, v[1]), 5) for k, v in kt.items()}
        launch_bytes = 2.0 * (kb + vb) * n
        scatter_ms = kt["scatter"][0] / max(1, kt["scatter"][1])
        bytes_per_elem = (kb * 2) * 2 * (kb + vb)  # passes x (read + write) x element bytes
        res[name] = {"ms_per_sort": round(dt * 1e3, 4), "G_elements_per_s": round(n / dt / 1e9, 3),
                     "algorithmic_bytes_per_element": bytes_per_elem,
                     "frac_of_8TBps": round(n / dt * bytes_per_elem / 1e9 / PEAK_HBM_GBPS, 4), "verified": ok,
                     "kernels_ms_per_launch": per_launch,
                     "launches_per_sort": {k: v[1] // (2 if log2n >= 28 else 10) for k, v in kt.items()},
                     "scatter_roofline": {"algorithmic_bytes_per_launch": launch_bytes,
                                          "achieved_GBps": round(launch_bytes / scatter_ms / 1e6, 1) if scatter_ms else 0.0,
                                          "frac": round(launch_bytes / scatter_ms / 1e6 / PEAK_HBM_GBPS, 4) if scatter_ms else 0.0}}
        s.close()
        del keys, vals, swap, vswap, index
        torch.cuda.empty_cache()
    return res


def kernel_source_sha():
    with open(os.path.join(ROOT, "radx_amd", "csrc", "radx_hip.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per scatter launch from the committed rocprofv3 PMC passes (counters cannot be
    read from inside the run).  The record carries the hash of the kernel source it was taken
    with; a record taken with other kernels than the ones being timed is not reported."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, "no PMC record"
    if rec.get("kernel_source_sha") != kernel_source_sha():
        return None, "PMC record is stale: taken with kernel source %s, timing %s" % (
            rec.get("kernel_source_sha"), kernel_source_sha())
    return rec.get("scatter_bytes_per_launch"), "profiles/pmc_traffic.json (%s)" % rec.get("tag")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started plainly: this process becomes the launcher.  Nothing here touches a GPU.
        if args.print_launch:
            cmd, envs = launch_plan(args.gpus, sys.argv[1:])
            print(json.dumps({"cmd": cmd, "env": envs}))
            return 0
        return launch_ranks(args.gpus, sys.argv[1:])
    # stdout carries exactly one line, the JSON: anything libraries print while the benchmark runs
    # (RCCL prints a version banner to stdout when its first communicator is created) goes to stderr
    sys.stdout.flush()
    json_out = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import radx_amd
    from radx_amd.sharded import HipEngine, HostStagedCollectives, ShardedSorter, init_process_group

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world  # the launcher (ours or torch.distributed.run) decides the number of ranks
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.rehearse_exchange  # the split / exchange / per-region path
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # a collective that does not complete within 4 minutes aborts the rank (and, through the
        # launcher, the run) instead of hanging the whole job
        limit = datetime.timedelta(seconds=240)
        where = {}
        if os.environ.get(PORT_FILE_ENV):  # our own launcher: no fixed port
            where["store"] = rendezvous_store(os.environ[PORT_FILE_ENV], rank, world, limit)
        if args.backend == "nccl":
            init_process_group(dev, rank=rank, world_size=world, timeout=limit, **where)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit, **where)
    # the collectives of the N > 1 path: RCCL on device tensors, or (rehearsal) gloo through host copies
    coll = dist if args.backend == "nccl" else HostStagedCollectives(dist)

    n = 1 << args.log2n
    K, W = args.steps, args.warmup
    cap = n if not multi else ShardedSorter.suggest_capacity(n)
    batch_bytes = cap * 4
    free_b, _ = torch.cuda.mem_get_info(dev)
    budget = int(free_b * 0.75) - 3 * batch_bytes
    pool_n = max(1, min(K + W, budget // batch_bytes))
    wall_mode = pool_n >= K + W

    pool = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(pool_n)]
    work = torch.empty(cap, dtype=torch.int32, device=dev)
    stage = torch.empty(cap, dtype=torch.int32, device=dev) if multi else None

    def fill(i, step):
        radx_amd.generate_u32(pool[i], seed=(step + 1) * 7919 + rank * 104729, n=n)

    for i in range(pool_n):
        fill(i, i)
    engine = HipEngine(cap, 4, local_rank)
    sharder = ShardedSorter(engine, exchange_when_alone=True, dist=coll) if multi else None
    checks = radx_amd.checksum(pool[(K + W - 1) % pool_n], n=n)  # of the last timed step's input

    def step(i):
        if not multi:
            engine.sorter.sort(pool[i], work, n=n)
            return n
        return sharder.sort(pool[i], work, n, stage)[1]

    def fence():
        torch.cuda.synchronize()
        if multi:
            coll.barrier()
            torch.cuda.synchronize()

    for w in range(W):
        step(w % pool_n)
    fence()
    last_len = n
    if wall_mode:
        t0 = time.perf_counter()
        for s in range(K):
            last_len = step(W + s)
        fence()
        elapsed = time.perf_counter() - t0
        timing = "wall clock over K steps, each on its own pre-generated batch"
    else:
        # The batches do not all fit in HBM: regenerate between steps and time each step with events.
        elapsed = 0.0
        for s in range(K):
            i = (W + s) % pool_n
            fill(i, W + s)
            fence()
            if s == K - 1:
                checks = radx_amd.checksum(pool[i], n=n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            last_len = step(i)
            e1.record()
            fence()
            elapsed += e0.elapsed_time(e1) / 1e3
        timing = "per-step HIP events (batch pool of %d regenerated between steps)" % pool_n
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        coll.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of the last timed step, outside the timed region
    last = pool[(W + K - 1) % pool_n]
    violations = radx_amd.check_sorted(last, n=last_len)
    verified = violations == 0
    after = radx_amd.checksum(last, n=last_len)
    if not multi:
        verified = verified and after == checks
    else:
        # the multiset of keys is preserved across the ranks: sum and mixed sum (mod 2^64) of all
        # shards equal those of all inputs
        def wrap(v):
            return v - (1 << 64) if v >= (1 << 63) else v
        sums = torch.tensor([wrap(checks[0]), wrap(checks[2]), wrap(after[0]), wrap(after[2])],
                            dtype=torch.int64, device=dev)
        coll.all_reduce(sums)
        sums = sums.tolist()
        verified = verified and sums[0] == sums[2] and sums[1] == sums[3]
        b = world.bit_length() - 1
        top = (last[:last_len] >> (32 - b)) & ((1 << b) - 1) if b else None
        if top is not None:
            verified = verified and bool((top == rank).all().item())
        tot = torch.tensor([last_len], dtype=torch.int64, device=dev)
        coll.all_reduce(tot)
        verified = verified and int(tot.item()) == n * world
        v = torch.tensor([1 if verified else 0], device=dev)
        coll.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(v.item())

    # Per-launch kernel times, outside the timed region: the same steps again on freshly generated
    # batches with every launch bracketed by HIP events on its own stream.
    P = min(K, 5)
    engine.sorter.profile(True)
    if multi:
        sharder.profile = True
    for s in range(P):
        i = s % pool_n
        fill(i, W + K + s)
        step(i)
        fence()
    kt = engine.sorter.kernel_times()
    engine.sorter.profile(False)
    timeline = sharder.timeline() if multi else None
    # how often a step made the host wait for the device: the gathered run starts, nothing else
    host_waits_per_step = sharder.host_waits / float(W + K + P) if multi else None

    # device-to-device copy ceiling of this GPU, same run
    src, dst = pool[0], work
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(src)
    e0.record()
    for _ in range(3):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbps = 3 * 2 * batch_bytes / (e0.elapsed_time(e1) / 1e3) / 1e9

    extras = None
    if not multi and not args.no_extras:
        del pool, work
        torch.cuda.empty_cache()
        extras = other_configs(dev)

    if rank == 0:
        ms_per_step = elapsed / K * 1e3
        value = world * n * K / elapsed / 1e9
        scatter_ms, scatter_launches = kt["scatter"]
        # algorithmic bytes of the scatter launches: every pass reads and writes every key once.
        # One launch covers n keys on one GPU and one region (n * N / 16 keys on average) on N > 1.
        avg_ms = scatter_ms / max(1, scatter_launches)
        alg_bytes = 8 * 2.0 * 4 * n * P / max(1, scatter_launches)
        achieved = alg_bytes / (avg_ms / 1e3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_source = pmc_traffic()
        out = {
            "metric": "Gkeys/s (uint32) at 1/2/4/8 MI355X; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Gkeys/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": "2^%d uniform (hashed) uint32 keys per GPU, full sort = 8 x 4-bit LSD passes%s"
                            % (args.log2n, "" if not multi else
                               ", split pass on the top digit + RCCL all-to-all-v exchange in %d "
                               "sub-buckets overlapped with the 7 local passes" % (16 // world)),
                "keys_per_gpu": n, "total_keys": n * world, "key_bytes": 4, "passes": 8,
                "parallelism": "1 GPU" if world == 1 else "msd-shard x%d" % world},
            "verified": verified, "timing": timing,
            "sort_roofline": {
                "algorithmic_bytes_per_key": 64, "achieved_GBps": round(64.0 * value, 1),
                "frac_of_8TBps": round(64.0 * value / world / PEAK_HBM_GBPS, 4),
                "copy_ceiling_GBps": round(copy_gbps, 1)},
            "kernels_ms_per_launch": {k: round(v[0] / max(1, v[1]), 4) for k, v in kt.items()},
            "roofline": {
                "kernel": "scatter_kernel<uint32>", "bound": "hbm", "achieved": round(achieved, 1),
                "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBPS, 4),
                "traffic": traffic if args.log2n == 30 and not multi else None,
                "traffic_source": traffic_source,
                "launches": scatter_launches, "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": alg_bytes,
                "timed_by": "HIP events around every launch, %d profiled steps after the timed region" % P},
        }
        if multi:
            # xGMI traffic of one step, reported beside the HBM figures (never part of them)
            out["exchange"] = {"bytes_sent_per_gpu": round(4.0 * n * (world - 1) / world),
                               "all_to_all_calls_per_step": 16 // world,
                               "backend": "nccl (RCCL)" if args.backend == "nccl" else "gloo (rehearsal)",
                               "overlapped_with_local_passes": args.backend == "nccl",
                               "host_waits_per_step": host_waits_per_step,
                               "last_step_timeline_ms_rank0": timeline}
        if extras is not None:
            out["other_configs"] = extras
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_log2n)
        elif world == 1:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(json_out, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    engine.close()
    if multi:
        coll.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
