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
    """CPU stand-in for radx_amd.sharded.HipEngine with the same contract: what each call reads,
    writes and is allowed to clobber."""

    def __init__(self, key_bytes=4):
        self.udt = np.uint32 if key_bytes == 4 else np.uint64
        self.sdt = np.int32 if key_bytes == 4 else np.int64
        self.bits = key_bytes * 8

    def _u(self, t, n):
        return t[:n].numpy().view(self.udt)

    def msd_split(self, src, dst, n, log2_ranks, starts_out, vsrc=None, vdst=None):
        k = self._u(src, n)
        d = (k >> self.udt(self.bits - 4)).astype(np.uint32)
        # bucket (rank r, sub-bucket j), d = r * S + j, goes to position p = j * G + r
        S = 16 >> log2_ranks
        p = (d % S) * (1 << log2_ranks) + d // S
        # one counting/partition pass in single-workgroup geometry: its partitions row is the
        # start of every run
        cfg = oracle.Config(bits=4, grid=1, wave=64, vec=4)
        counts = np.bincount(p, minlength=16).astype(np.uint32).reshape(1, 16)
        parts = oracle.partition(cfg, counts)
        order = np.argsort(p, kind="stable")
        dst[:n] = torch.from_numpy(k[order].view(self.sdt).copy())
        if vsrc is not None:
            vdst[:n] = vsrc[:n][torch.from_numpy(order)]
        starts_out[:17] = torch.from_numpy(np.concatenate([parts[0], [n]]).astype(np.uint32).view(np.int32))

    def _order_on(self, k, begin_bit, end_bit):
        field = (k >> self.udt(begin_bit)) & self.udt((1 << (end_bit - begin_bit)) - 1)
        return np.argsort(field, kind="stable")

    def sort_bits(self, keys, swap, n, begin_bit, end_bit, vals=None, vswap=None):
        k = self._u(keys, n)
        order = self._order_on(k, begin_bit, end_bit)
        res = k[order].view(self.sdt).copy()
        swap[:n] = -1
        keys[:n] = torch.from_numpy(res)
        if vals is not None:
            moved = vals[:n][torch.from_numpy(order)].clone()
            vswap[:n] = -1
            vals[:n] = moved

    def sort_bits_into(self, keys, swap, out, n, begin_bit, end_bit, vals=None, vswap=None, vout=None):
        k = self._u(keys, n)
        order = self._order_on(k, begin_bit, end_bit)
        res = k[order].view(self.sdt).copy()
        moved = vals[:n][torch.from_numpy(order)].clone() if vals is not None else None
        keys[:n] = -1  # both ping-pong buffers are scratch
        swap[:n] = -1
        out[:n] = torch.from_numpy(res)
        if vals is not None:
            vals[:n] = -1
            vswap[:n] = -1
            vout[:n] = moved


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_local, dist_name, key_bytes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radx_amd.sharded import ShardedSorter
        sdt = np.int32 if key_bytes == 4 else np.int64
        tdt = torch.int32 if key_bytes == 4 else torch.int64
        bits = key_bytes * 8
        keys = make_keys(dist_name, n_local, key_bytes, seed=100 + rank)
        cap = n_local * world + 64 if dist_name != "hash" else ShardedSorter.suggest_capacity(n_local, 0.2)
        kbuf = torch.zeros(cap, dtype=tdt)
        kbuf[:n_local] = torch.from_numpy(keys.view(sdt).copy())
        work = torch.zeros(cap, dtype=tdt)
        sharder = ShardedSorter(OracleEngine(key_bytes), key_bits=bits, exchange_when_alone=True)
        shard, total, send, recv = sharder.sort(kbuf, work, n_local)
        got = shard.numpy().view(keys.dtype).copy()
        assert got.size == total == int(recv.sum())
        assert int(send.sum()) == n_local
        b = world.bit_length() - 1
        if b:
            assert np.all((got >> keys.dtype.type(bits - b)) == rank), "shard holds a key of another rank's range"
        assert np.all(got[1:] >= got[:-1]), "shard not sorted"
        np.save(os.path.join(out_dir, "shard%d.npy" % rank), got)
        np.save(os.path.join(out_dir, "input%d.npy" % rank), keys)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dist_name,key_bytes", [
    (2, "hash", 4), (2, "skewed", 4), (4, "hash", 4), (2, "all_equal", 4), (8, "hash", 4),
    (1, "hash", 4), (4, "hash", 8), (4, "sorted", 4)])
def test_sharded_sort_gloo(tmp_path, world, dist_name, key_bytes):
    n_local = 20000 + 37
    mp.spawn(_worker, args=(world, _free_port(), n_local, dist_name, key_bytes, str(tmp_path)),
             nprocs=world, join=True)
    shards = [np.load(tmp_path / ("shard%d.npy" % r)) for r in range(world)]
    inputs = [np.load(tmp_path / ("input%d.npy" % r)) for r in range(world)]
    whole = np.concatenate(shards)
    assert np.array_equal(whole, np.sort(np.concatenate(inputs)))
    assert oracle.checksum(whole) == oracle.checksum(np.concatenate(inputs))


def _overflow_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radx_amd.sharded import ShardedSorter
        n_local = 5000
        keys = np.full(n_local, 0x10000000 + rank, dtype=np.uint32)  # every key belongs to rank 0
        cap = n_local + 100                                          # rank 0 would need 2 * n_local
        kbuf = torch.zeros(cap, dtype=torch.int32)
        kbuf[:n_local] = torch.from_numpy(keys.view(np.int32).copy())
        work = torch.zeros(cap, dtype=torch.int32)
        try:
            ShardedSorter(OracleEngine(4)).sort(kbuf, work, n_local)
            outcome = "sorted"
        except RuntimeError as e:
            outcome = str(e)
        with open(os.path.join(out_dir, "outcome%d.txt" % rank), "w") as f:
            f.write(outcome)
    finally:
        dist.destroy_process_group()


def test_shard_overflow_is_raised_on_every_rank(tmp_path):
    """A shard that does not fit its rank's buffers is an error on ALL ranks, before the exchange
    starts — nobody is left waiting in a collective."""
    mp.spawn(_overflow_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        text = (tmp_path / ("outcome%d.txt" % r)).read_text()
        assert "rank 0 receives 10000 keys" in text and "allocate more slack" in text


def test_plan_regions_do_not_overlap():
    """Region j's staging range starts 16-byte aligned at or after its packed position and ends
    before region j+1's staging range starts."""
    from radx_amd.sharded import ShardedSorter

    class Alone(ShardedSorter):
        def __init__(self, world, rank, key_bits):
            self.world, self.rank, self.align = world, rank, 16 // (key_bits // 8)

    rng = np.random.default_rng(7)
    for world in (1, 2, 4, 8, 16):
        for key_bits in (32, 64):
            for rank in (0, world - 1):
                matrix = rng.integers(0, 50, size=(world, 16)).astype(np.int64)
                matrix[rng.integers(0, world), :] = 0
                recv, length, exact, staged = Alone(world, rank, key_bits).plan(matrix)
                S = 16 // world
                assert recv.shape == (S, world)
                a = 16 // (key_bits // 8)
                for j in range(S):
                    assert staged[j] % a == 0 and staged[j] >= exact[j]
                    assert length[j] == sum(matrix[s, j * world + rank] for s in range(world))
                    if j + 1 < S:
                        assert staged[j] + length[j] <= staged[j + 1]
                        assert exact[j] + length[j] == exact[j + 1]


def _kv_worker(rank, world, port, n_local, key_bytes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radx_amd.sharded import ShardedSorter
        sdt = np.int32 if key_bytes == 4 else np.int64
        tdt = torch.int32 if key_bytes == 4 else torch.int64
        keys = make_keys("few_distinct", n_local, key_bytes, seed=200 + rank)      # many ties
        keys[::3] = make_keys("uniform", n_local, key_bytes, seed=300 + rank)[::3]
        vals = (np.arange(n_local, dtype=np.int64) + rank * n_local).astype(np.int32)  # global input position
        cap = n_local * world + 64
        kbuf, work = torch.zeros(cap, dtype=tdt), torch.zeros(cap, dtype=tdt)
        vbuf, vwork = torch.zeros(cap, dtype=torch.int32), torch.zeros(cap, dtype=torch.int32)
        kbuf[:n_local] = torch.from_numpy(keys.view(sdt).copy())
        vbuf[:n_local] = torch.from_numpy(vals)
        sharder = ShardedSorter(OracleEngine(key_bytes), key_bits=key_bytes * 8, exchange_when_alone=True)
        shard, total, send, recv = sharder.sort(kbuf, work, n_local, vals=vbuf, vals_work=vwork)
        np.save(os.path.join(out_dir, "shard%d.npy" % rank), shard.numpy().view(keys.dtype).copy())
        np.save(os.path.join(out_dir, "vals%d.npy" % rank), vbuf[:total].numpy().copy())
        np.save(os.path.join(out_dir, "input%d.npy" % rank), keys)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,key_bytes", [(2, 4), (4, 8), (1, 4), (8, 4)])
def test_sharded_key_value_sort_is_globally_stable(tmp_path, world, key_bytes):
    """Payload = position in the concatenation of the ranks' inputs: after the sharded sort the
    shards, concatenated, are the stable sort of that concatenation, payloads included."""
    n_local = 6000 + 13
    mp.spawn(_kv_worker, args=(world, _free_port(), n_local, key_bytes, str(tmp_path)), nprocs=world, join=True)
    keys = np.concatenate([np.load(tmp_path / ("input%d.npy" % r)) for r in range(world)])
    got_k = np.concatenate([np.load(tmp_path / ("shard%d.npy" % r)) for r in range(world)])
    got_v = np.concatenate([np.load(tmp_path / ("vals%d.npy" % r)) for r in range(world)])
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(got_k, keys[order])
    assert np.array_equal(got_v, order.astype(np.int32))


def _staged_worker(rank, world, port, n_local, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radx_amd.sharded import HostStagedCollectives, ShardedSorter
        coll = HostStagedCollectives(dist)
        assert coll.get_world_size() == world and coll.get_rank() == rank
        keys = make_keys("hash", n_local, 4, seed=600 + rank)
        cap = ShardedSorter.suggest_capacity(n_local, 0.3)
        kbuf, work = torch.zeros(cap, dtype=torch.int32), torch.zeros(cap, dtype=torch.int32)
        kbuf[:n_local] = torch.from_numpy(keys.view(np.int32).copy())
        shard, total, send, recv = ShardedSorter(OracleEngine(4), dist=coll).sort(kbuf, work, n_local)
        t = torch.tensor([total], dtype=torch.int64)
        coll.all_reduce(t)
        assert int(t.item()) == n_local * world
        coll.barrier()
        np.save(os.path.join(out_dir, "shard%d.npy" % rank), shard.numpy().view(np.uint32).copy())
        np.save(os.path.join(out_dir, "input%d.npy" % rank), keys)
    finally:
        dist.destroy_process_group()


def test_host_staged_collectives_adapter(tmp_path):
    """The adapter that carries the collectives through host copies (used to rehearse the
    multi-rank path where ranks share a GPU) behaves like torch.distributed for the sorter."""
    world = 4
    mp.spawn(_staged_worker, args=(world, _free_port(), 9000 + 5, str(tmp_path)), nprocs=world, join=True)
    shards = [np.load(tmp_path / ("shard%d.npy" % r)) for r in range(world)]
    inputs = [np.load(tmp_path / ("input%d.npy" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate(shards), np.sort(np.concatenate(inputs)))
