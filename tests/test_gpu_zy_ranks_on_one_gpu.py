"""ShardedSorter with the HIP engine in REAL multi-process worlds of 2 and 4 ranks that share the
one GPU of the test box: every rank is its own process with its own sorter handle, streams and
device buffers, and the collectives move data between the processes (gloo through host copies —
HostStagedCollectives — since RCCL refuses a communicator with two ranks on one device).  What this adds to the replayed-rank tests: the split
counts really travel through all_gather, the uneven all-to-all-v's really deliver other
processes' keys into `stage`, and the per-region local passes wait on genuine asynchronous work.
Collected late: it starts child processes that use the GPU."""
import os
import socket

import numpy as np
import pytest

from datasets import make_keys

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_local, dist_name, key_bytes, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import radx_amd
        from radx_amd.sharded import HipEngine, HostStagedCollectives, ShardedSorter
        sdt = np.int32 if key_bytes == 4 else np.int64
        tdt = torch.int32 if key_bytes == 4 else torch.int64
        bits = key_bytes * 8
        sizes = [n_local + 1000 * r for r in range(world)]       # ranks contribute different amounts
        keys = make_keys(dist_name, sizes[rank], key_bytes, seed=700 + rank)
        cap = sum(sizes) + 64
        eng = HipEngine(cap, key_bytes, 0)
        kbuf = torch.zeros(cap, dtype=tdt, device=dev)
        kbuf[:sizes[rank]] = torch.from_numpy(keys.view(sdt).copy()).to(dev)
        work = torch.zeros(cap, dtype=tdt, device=dev)
        sharder = ShardedSorter(eng, key_bits=bits, exchange_when_alone=True, dist=HostStagedCollectives(dist))
        for _ in range(3):  # again and again: every buffer, stream and the sorter's tables are reused, no host sync between
            kbuf[:sizes[rank]] = torch.from_numpy(keys.view(sdt).copy()).to(dev)
            shard, total, send, recv = sharder.sort(kbuf, work, sizes[rank])
        torch.cuda.synchronize()
        got = shard.cpu().numpy().view(keys.dtype).copy()
        assert got.size == total == int(recv.sum()) and int(send.sum()) == sizes[rank]
        assert radx_amd.check_sorted(kbuf, n=total) == 0
        b = world.bit_length() - 1
        assert np.all((got >> keys.dtype.type(bits - b)) == rank), "shard holds a key of another rank's range"
        np.save(os.path.join(out_dir, "shard%d.npy" % rank), got)
        np.save(os.path.join(out_dir, "input%d.npy" % rank), keys)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dist_name,key_bytes", [(2, "hash", 4), (4, "hash", 4), (2, "skewed", 4), (4, "uniform", 8)])
def test_ranks_sharing_one_gpu(device, tmp_path, world, dist_name, key_bytes):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), 150000 + 11, dist_name, key_bytes, str(tmp_path)),
             nprocs=world, join=True)
    shards = [np.load(tmp_path / ("shard%d.npy" % r)) for r in range(world)]
    inputs = [np.load(tmp_path / ("input%d.npy" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate(shards), np.sort(np.concatenate(inputs)))


def _kv_worker(rank, world, port, n_local, key_bytes, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radx_amd.sharded import HipEngine, HostStagedCollectives, ShardedSorter
        sdt = np.int32 if key_bytes == 4 else np.int64
        tdt = torch.int32 if key_bytes == 4 else torch.int64
        keys = make_keys("few_distinct", n_local, key_bytes, seed=800 + rank)
        keys[::3] = make_keys("uniform", n_local, key_bytes, seed=900 + rank)[::3]
        vals = (np.arange(n_local, dtype=np.int64) + rank * n_local).astype(np.int32)
        cap = n_local * world + 64
        eng = HipEngine(cap, key_bytes, 0, value_bytes=4)
        kbuf, work = (torch.zeros(cap, dtype=tdt, device=dev) for _ in range(2))
        vbuf, vwork = (torch.zeros(cap, dtype=torch.int32, device=dev) for _ in range(2))
        sharder = ShardedSorter(eng, key_bits=key_bytes * 8, exchange_when_alone=True, dist=HostStagedCollectives(dist))
        for _ in range(2):
            kbuf[:n_local] = torch.from_numpy(keys.view(sdt).copy()).to(dev)
            vbuf[:n_local] = torch.from_numpy(vals).to(dev)
            shard, total, send, recv = sharder.sort(kbuf, work, n_local, vals=vbuf, vals_work=vwork)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "shard%d.npy" % rank), shard.cpu().numpy().view(keys.dtype).copy())
        np.save(os.path.join(out_dir, "vals%d.npy" % rank), vbuf[:total].cpu().numpy().copy())
        np.save(os.path.join(out_dir, "input%d.npy" % rank), keys)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,key_bytes", [(2, 4), (4, 8)])
def test_key_value_ranks_sharing_one_gpu(device, tmp_path, world, key_bytes):
    """Payloads travel with their keys through the split, the exchange and the local passes; the
    shards concatenate to the stable sort of all ranks' inputs (payload = global input position)."""
    import torch.multiprocessing as mp
    n_local = 90000 + 7
    mp.spawn(_kv_worker, args=(world, _free_port(), n_local, key_bytes, str(tmp_path)), nprocs=world, join=True)
    keys = np.concatenate([np.load(tmp_path / ("input%d.npy" % r)) for r in range(world)])
    got_k = np.concatenate([np.load(tmp_path / ("shard%d.npy" % r)) for r in range(world)])
    got_v = np.concatenate([np.load(tmp_path / ("vals%d.npy" % r)) for r in range(world)])
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(got_k, keys[order])
    assert np.array_equal(got_v, order.astype(np.int32))


@pytest.mark.parametrize("gpus", [2, 8])
def test_bench_launches_its_own_ranks(device, gpus):
    """`python bench.py --gpus N`, started plainly: the launcher starts the rank processes, they run
    the split / exchange / local-pass path and its cross-rank verification, and rank 0's line comes
    back through the launcher.  The ranks share this box's one GPU, hence gloo instead of RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--backend", "gloo", "--share-gpu",
                          "--log2n", "20", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras"],
                         capture_output=True, text=True, timeout=400, env=env, cwd=root)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == gpus and line["verified"] is True and line["scaling"] == "weak"
    assert line["config"]["total_keys"] == gpus << 20 and line["exchange"]["all_to_all_calls_per_step"] == 16 // gpus
    assert line["roofline"]["launches"] > 0
