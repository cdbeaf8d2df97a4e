"""The multi-GPU exchange logic on CPU: world_size-2 and -4 gloo process groups, with a CPU
engine (oracle-backed test double for the split / local-sort kernels) injected into
ShardedSorter.  Checks shard boundaries, per-shard order, multiset preservation and full
equality against a single sort of all ranks' keys."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from datasets import make_keys


class OracleEngine:
    """CPU stand-in for radx_amd.sharded.HipEngine, built on the oracle's pass functions."""

    def split(self, src, dst, n, bit, num_bits):
        k = src[:n].numpy().view(np.uint32)
        # one counting/partition/scattering pass on the digit (key >> bit) & (2^num_bits - 1):
        # single-workgroup geometry, whose partitions row is the start of every digit's run
        cfg = oracle.Config(bits=4, grid=1, wave=64, vec=4)
        digit = (k >> np.uint32(bit)) & np.uint32((1 << num_bits) - 1)
        counts = np.bincount(digit, minlength=16).astype(np.uint32).reshape(1, 16)
        parts = oracle.partition(cfg, counts)
        order = np.argsort(digit, kind="stable")
        dst[:n] = torch.from_numpy(k[order].view(np.int32).copy())
        return np.concatenate([parts[0], [n]]).astype(np.uint32)

    def sort_bits(self, keys, swap, n, begin_bit, end_bit):
        k = keys[:n].numpy().view(np.uint32)
        field = (k >> np.uint32(begin_bit)) & np.uint32((1 << (end_bit - begin_bit)) - 1)
        keys[:n] = torch.from_numpy(k[np.argsort(field, kind="stable")].view(np.int32).copy())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_local, dist_name, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radx_amd.sharded import ShardedSorter
        keys = make_keys(dist_name, n_local, 4, seed=100 + rank)
        cap = n_local * world if dist_name != "hash" else ShardedSorter.suggest_capacity(n_local, 0.2)
        kbuf = torch.zeros(cap, dtype=torch.int32)
        kbuf[:n_local] = torch.from_numpy(keys.view(np.int32).copy())
        work = torch.zeros(cap, dtype=torch.int32)
        shard, total, send, recv = ShardedSorter(OracleEngine()).sort(kbuf, work, n_local)
        got = shard.numpy().view(np.uint32).copy()
        assert got.size == total == int(recv.sum())
        assert int(send.sum()) == n_local
        b = world.bit_length() - 1
        if b:
            assert np.all((got >> np.uint32(32 - b)) == rank), "shard holds a key of another rank's range"
        assert np.all(got[1:] >= got[:-1]), "shard not sorted"
        np.save(os.path.join(out_dir, "shard%d.npy" % rank), got)
        np.save(os.path.join(out_dir, "input%d.npy" % rank), keys)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dist_name", [(2, "hash"), (2, "skewed"), (4, "hash"), (2, "all_equal")])
def test_sharded_sort_gloo(tmp_path, world, dist_name):
    n_local = 20000 + 37
    mp.spawn(_worker, args=(world, _free_port(), n_local, dist_name, str(tmp_path)), nprocs=world, join=True)
    shards = [np.load(tmp_path / ("shard%d.npy" % r)) for r in range(world)]
    inputs = [np.load(tmp_path / ("input%d.npy" % r)) for r in range(world)]
    whole = np.concatenate(shards)
    assert np.array_equal(whole, np.sort(np.concatenate(inputs)))
    assert oracle.checksum(whole) == oracle.checksum(np.concatenate(inputs))
