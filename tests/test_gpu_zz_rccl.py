"""Everything that goes through the real RCCL backend on the one GPU of the test box, collected
after the other GPU tests so that trouble in the communication stack cannot hide a parity
result: ShardedSorter over a "nccl" process group of one rank, bench.py's N > 1 code path
rehearsed the same way, and the C++ host (radx::ShardedSort) over ncclCommInitAll."""
import os
import socket

import numpy as np
import pytest

from datasets import make_keys
from test_gpu_bench import check_contract, run_bench
from test_gpu_harness import run_sharded
from test_gpu_sort import to_dev, to_host

pytestmark = pytest.mark.gpu


def test_sharded_sorter_over_rccl_world_of_one(device):
    """The whole pipeline — split, all_gather, 16 asynchronous all-to-all-v's on RCCL's stream,
    per-region sorts chained behind them — through the real "nccl" backend with one rank."""
    import torch
    import torch.distributed as dist
    import radx_amd
    from radx_amd.sharded import HipEngine, ShardedSorter, init_process_group
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    init_process_group(torch.device(device), rank=0, world_size=1)
    try:
        n = (1 << 22) + 12345
        cap = n + 4096
        eng = HipEngine(cap, 4, 0)
        sharder = ShardedSorter(eng, exchange_when_alone=True)
        keys = torch.empty(cap, dtype=torch.int32, device=device)
        work = torch.empty_like(keys)
        for step in range(3):  # back-to-back steps reuse work/stage while nothing is synchronised in between
            radx_amd.generate_u32(keys, seed=77 + step, n=n)
            before = radx_amd.checksum(keys, n=n)
            shard, total, send, recv = sharder.sort(keys, work, n)
            assert total == n
            assert radx_amd.check_sorted(shard, n=n) == 0
            assert radx_amd.checksum(shard, n=n) == before
        small = make_keys("hash", 70001, 4, seed=9)
        keys[:small.size] = to_dev(small, device)
        shard, total, _, _ = sharder.sort(keys, work, small.size)
        assert np.array_equal(to_host(shard), np.sort(small))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_bench_exchange_path_rehearsed(device):
    line = run_bench("--log2n", "22", "--steps", "2", "--warmup", "1", "--no-cpu", "--rehearse-exchange")
    check_contract(line, 2, 1)
    assert line["exchange"]["all_to_all_calls_per_step"] == 16
    assert line["exchange"]["host_waits_per_step"] == 1.0   # the gathered run starts; nothing else blocks the host
    timeline = line["exchange"]["last_step_timeline_ms_rank0"]
    assert len(timeline["exchange_landed"]) == len(timeline["region_sorted"]) == 16
    assert 0 < timeline["split_done"] < timeline["region_sorted"][-1]
    # one split pass over all keys + 7 passes over each of the 16 regions, per step
    assert line["roofline"]["launches"] == 2 * (1 + 7 * 16)


@pytest.mark.parametrize("log2n", ["16", "22"])
def test_sharded_harness_one_gpu_through_rccl(device, log2n):
    """radx::ShardedSort from a C++ host: split, 16 grouped RCCL send/recv exchanges (to itself),
    per-region passes on their own stream; checked for full equality with std::sort."""
    line = run_sharded("--gpus", "1", "--log2n", log2n, "--iters", "2", "--force-exchange")
    assert line["pass"] is True and line["exchange"] is True and line["full_equality_checked"] is True
    assert line["host_waits_per_sort"] == 1.0


@pytest.mark.parametrize("log2n", ["15", "21"])
def test_sharded_harness_key_value_globally_stable(device, log2n):
    """uint64 keys + uint32 payloads through radx::ShardedSort: the payloads ride through the
    split, a second grouped send/recv per sub-bucket and the local passes; keys (thinned, so ties
    occur) and payloads equal std::stable_sort of the input — global stability."""
    line = run_sharded("--gpus", "1", "--log2n", log2n, "--iters", "1", "--force-exchange", "--type", "u64kv")
    assert line["pass"] is True and line["exchange"] is True and line["full_equality_checked"] is True
    assert line["type"] == "u64kv" and line["host_waits_per_sort"] == 1.0
