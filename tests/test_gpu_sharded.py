"""GPU tests of the multi-GPU path's building blocks and of ShardedSorter itself on one GPU:
the split pass, the pack-into-place local sort, and the whole pipeline with the exchange replayed
by copies for every rank of a 1..16-rank world (the real RCCL backend: test_gpu_zz_rccl.py)."""
import numpy as np
import pytest

from datasets import make_keys
from test_gpu_sort import to_dev, to_host

pytestmark = pytest.mark.gpu


def bucket_position(keys, log2_ranks):
    """Position p = j * G + r of the top digit d = r * S + j of every key."""
    bits = keys.itemsize * 8
    d = (keys >> keys.dtype.type(bits - 4)).astype(np.int64)
    S = 16 >> log2_ranks
    return (d % S) * (1 << log2_ranks) + d // S


@pytest.mark.parametrize("log2_ranks", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("key_bytes,with_vals", [(4, False), (8, False), (8, True), (4, True)])
def test_msd_split(device, log2_ranks, key_bytes, with_vals):
    import torch
    import radx_amd
    n = 3 * 4096 + 1234
    keys = make_keys("hash", n, key_bytes, seed=21 + log2_ranks)
    vals = np.arange(n, dtype=np.uint32)
    s = radx_amd.Sorter(n, key_bytes, 4 if with_vals else 0)
    src, dst = to_dev(keys, device), None
    dst = torch.empty_like(src)
    vsrc = to_dev(vals, device) if with_vals else None
    vdst = torch.empty_like(vsrc) if with_vals else None
    s.msd_split(src, dst, log2_ranks, vsrc, vdst)
    starts = s.digit_starts()
    p = bucket_position(keys, log2_ranks)
    order = np.argsort(p, kind="stable")
    assert np.array_equal(to_host(dst, key_bytes), keys[order])
    if with_vals:
        assert np.array_equal(to_host(vdst), vals[order])
    assert starts[16] == n
    assert np.array_equal(np.diff(starts.astype(np.int64)), np.bincount(p, minlength=16))
    s.close()


def test_msd_split_follows_key_order(device):
    """With a signed, descending key order the split digit is the digit of the order-preserving
    image; the keys come out with their own bits."""
    import torch
    import radx_amd
    n = 9000
    keys = make_keys("uniform", n, 4, seed=5)
    s = radx_amd.Sorter(n, 4, 0)
    s.set_key_order("signed", descending=True)
    src = to_dev(keys, device)
    dst = torch.empty_like(src)
    s.msd_split(src, dst, 2)
    image = ~(keys ^ np.uint32(0x80000000))
    order = np.argsort(bucket_position(image, 2), kind="stable")
    assert np.array_equal(to_host(dst), keys[order])
    s.close()


@pytest.mark.parametrize("key_bytes,begin,end,with_vals", [
    (4, 0, 28, False), (4, 0, 32, False), (4, 0, 4, False), (4, 8, 8, False), (4, 3, 14, True),
    (8, 0, 60, False), (8, 0, 60, True), (8, 4, 12, False)])
@pytest.mark.parametrize("offset", [0, 1, 3])
def test_sort_bits_into(device, key_bytes, begin, end, with_vals, offset):
    """Exact pass count, last pass stored at an arbitrary element offset of a third buffer."""
    import torch
    import radx_amd
    n = 2 * 4096 + 777
    keys = make_keys("hash", n, key_bytes, seed=31)
    vals = np.arange(n, dtype=np.uint32)
    s = radx_amd.Sorter(n, key_bytes, 4 if with_vals else 0)
    k = to_dev(keys, device)
    swap = torch.empty_like(k)
    out = torch.zeros(n + 8, dtype=k.dtype, device=device)
    v = to_dev(vals, device) if with_vals else None
    vswap = torch.empty_like(v) if with_vals else None
    vout = torch.zeros(n + 8, dtype=torch.int32, device=device) if with_vals else None
    s.sort_bits_into(k, swap, out[offset:], begin, end, v, vswap, vout[offset:] if with_vals else None)
    torch.cuda.synchronize()
    field = (keys >> keys.dtype.type(begin)) & keys.dtype.type((1 << (end - begin)) - 1)
    order = np.argsort(field, kind="stable")
    got = to_host(out, key_bytes)
    assert np.array_equal(got[offset:offset + n], keys[order])
    assert not got[:offset].any() and not got[offset + n:].any(), "stored outside the destination range"
    if with_vals:
        assert np.array_equal(to_host(vout)[offset:offset + n], vals[order])
    s.close()


def test_sort_bits_into_may_pack_into_its_own_swap_buffer(device):
    """The destination may overlap the ping-pong buffer the last pass does NOT read: an odd pass
    count reads `keys` last, so `swap` shifted by a few elements is a legal destination."""
    import torch
    import radx_amd
    n = 4096 + 99
    keys = make_keys("uniform", n, 4, seed=32)
    s = radx_amd.Sorter(n + 8, 4, 0)
    k = to_dev(keys, device)
    big = torch.zeros(n + 8, dtype=torch.int32, device=device)
    s.sort_bits_into(k, big[4:], big[1:], 0, 28, n=n)
    torch.cuda.synchronize()
    order = np.argsort(keys & np.uint32((1 << 28) - 1), kind="stable")
    assert np.array_equal(to_host(big)[1:1 + n], keys[order])
    with pytest.raises(radx_amd.RadxError):   # even pass count: the last pass reads swap
        s.sort_bits_into(k, big[4:], big[1:], 0, 32, n=n)
    with pytest.raises(radx_amd.RadxError):   # destination is the last pass's own source
        s.sort_bits_into(k, big[4:], k, 0, 28, n=n)
    s.close()


def split_starts(eng, src, dst, n, log2_ranks):
    """HipEngine.msd_split leaves its 17 run starts on the device; here they are read back."""
    import torch
    out = torch.full((17,), -1, dtype=torch.int32, device=src.device)
    eng.msd_split(src, dst, n, log2_ranks, out)
    return out.cpu().numpy().view(np.uint32).astype(np.int64)


class ReplayDist:
    """Stands in for torch.distributed from ONE rank's point of view: the other ranks' split
    outputs were computed beforehand on the same GPU, and each all-to-all-v is replayed by
    device copies in exactly the pattern the collective would move data."""

    class Done:
        def wait(self):
            return True

    def __init__(self, rank, world, split_bufs, starts):
        self.rank, self.world, self.split_bufs, self.starts = rank, world, split_bufs, starts
        self.calls = 0

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return self.rank

    def all_gather_into_tensor(self, out, mine, group=None):
        import torch
        mine = mine.cpu().numpy().view(np.uint32).astype(np.int64)   # 17 run starts + capacity
        assert np.array_equal(mine[:17], self.starts[self.rank])
        rows = np.concatenate([np.stack(self.starts), np.full((self.world, 1), mine[17])], axis=1)  # same capacity everywhere
        out.copy_(torch.from_numpy(rows.astype(np.uint32).view(np.int32).reshape(-1)).to(out.device))

    def all_to_all_single(self, out, inp, output_split_sizes, input_split_sizes, group=None, async_op=False):
        j, G = self.calls, self.world
        self.calls += 1
        mine = self.starts[self.rank]
        if inp.numel():
            assert inp.data_ptr() == self.split_bufs[self.rank][mine[j * G]:].data_ptr()
        assert inp.numel() == mine[(j + 1) * G] - mine[j * G] == sum(input_split_sizes)
        assert input_split_sizes == np.diff(mine[j * G:(j + 1) * G + 1]).tolist()
        assert out.numel() == sum(output_split_sizes)
        at = 0
        for src in range(G):
            st = self.starts[src]
            lo, hi = st[j * G + self.rank], st[j * G + self.rank + 1]
            assert hi - lo == output_split_sizes[src]
            out[at:at + hi - lo] = self.split_bufs[src][lo:hi]
            at += hi - lo
        return self.Done()


@pytest.mark.parametrize("world,key_bytes,dist_name", [
    (1, 4, "hash"), (2, 4, "hash"), (4, 4, "hash"), (8, 4, "hash"), (16, 4, "hash"),
    (8, 8, "hash"), (4, 4, "skewed"), (8, 4, "sorted"), (2, 4, "all_equal")])
def test_sharded_sorter_every_rank_replayed(device, world, key_bytes, dist_name):
    """ShardedSorter.sort with the HIP engine, run once per rank of the world on this one GPU."""
    import torch
    from radx_amd.sharded import HipEngine, ShardedSorter
    b = world.bit_length() - 1
    bits = key_bytes * 8
    n_local = 50000 + 17
    cap = n_local * world + 64 if dist_name != "hash" else ShardedSorter.suggest_capacity(n_local, 0.25)
    eng = HipEngine(cap, key_bytes, 0)
    inputs = [make_keys(dist_name, n_local, key_bytes, seed=300 + r) for r in range(world)]
    split_bufs, starts = [], []
    for r in range(world):
        src = to_dev(inputs[r], device)
        dst = torch.empty_like(src)
        starts.append(split_starts(eng, src, dst, n_local, b))
        split_bufs.append(dst)
    tdt = torch.int32 if key_bytes == 4 else torch.int64
    whole = []
    for r in range(world):
        keys = torch.zeros(cap, dtype=tdt, device=device)
        keys[:n_local] = to_dev(inputs[r], device)
        work = torch.zeros(cap, dtype=tdt, device=device)
        fake = ReplayDist(r, world, split_bufs, starts)
        sharder = ShardedSorter(eng, key_bits=bits, exchange_when_alone=True, dist=fake)
        # the sorter's own split must reproduce this rank's precomputed one, into `work`
        split_bufs[r] = work
        shard, total, send, recv = sharder.sort(keys, work, n_local)
        torch.cuda.synchronize()
        assert fake.calls == 16 // world
        assert sharder.host_waits == 1     # the gathered run starts; nothing else makes the host wait
        got = to_host(shard, key_bytes).copy()
        assert got.size == total == int(recv.sum()) and int(send.sum()) == n_local
        if b:
            assert np.all((got >> got.dtype.type(bits - b)) == r)
        whole.append(got)
    eng.close()
    assert np.array_equal(np.concatenate(whole), np.sort(np.concatenate(inputs)))


def test_empty_split_reports_empty_runs(device):
    """A split (or pass) over zero elements leaves 17 zero run starts, not the previous pass's."""
    import torch
    import radx_amd
    n = 5000
    s = radx_amd.Sorter(n, 4, 0)
    src = to_dev(make_keys("uniform", n, 4, seed=1), device)
    dst = torch.empty_like(src)
    assert not s.digit_starts().any()          # fresh sorter: zeros, not uninitialised memory
    s.msd_split(src, dst, 2)
    assert s.digit_starts()[16] == n
    s.msd_split(src, dst, 2, n=0)
    assert not s.digit_starts().any()
    with pytest.raises(radx_amd.RadxError):    # and no tables are left to read for the old n
        s.tables(n)
    s.run_pass(src, dst, bit=0)
    assert s.digit_starts()[16] == n
    s.run_pass(src, dst, bit=0, n=0)
    assert not s.digit_starts().any()
    s.close()


@pytest.mark.parametrize("empty_rank", [0, 2])
def test_sharded_sorter_with_an_empty_rank(device, empty_rank):
    """One rank contributes no keys: its 16 run lengths are zeros for every rank's plan, it still
    takes part in every exchange and receives its shard of the others' keys."""
    import torch
    from radx_amd.sharded import HipEngine, ShardedSorter
    world, b = 4, 2
    sizes = [30000 + 7 * r for r in range(world)]
    sizes[empty_rank] = 0
    cap = sum(sizes) + 64
    eng = HipEngine(cap, 4, 0)
    inputs = [make_keys("hash", sizes[r], 4, seed=500 + r) for r in range(world)]
    split_bufs, starts = [], []
    for r in range(world):
        src = to_dev(inputs[r], device) if sizes[r] else torch.empty(4, dtype=torch.int32, device=device)
        dst = torch.empty_like(src)
        starts.append(split_starts(eng, src, dst, sizes[r], b))
        split_bufs.append(dst)
    assert not starts[empty_rank].any()
    whole = []
    for r in range(world):
        keys = torch.zeros(cap, dtype=torch.int32, device=device)
        if sizes[r]:
            keys[:sizes[r]] = to_dev(inputs[r], device)
        work = torch.zeros(cap, dtype=torch.int32, device=device)
        split_bufs[r] = work
        sharder = ShardedSorter(eng, exchange_when_alone=True, dist=ReplayDist(r, world, split_bufs, starts))
        shard, total, send, recv = sharder.sort(keys, work, sizes[r])
        torch.cuda.synchronize()
        assert int(send.sum()) == sizes[r] and recv[empty_rank] == 0
        got = to_host(shard).copy()
        assert got.size == total and np.all((got >> np.uint32(30)) == r)
        whole.append(got)
    eng.close()
    assert np.array_equal(np.concatenate(whole), np.sort(np.concatenate(inputs)))


@pytest.mark.parametrize("kind,descending", [("signed", False), ("float", True), ("unsigned", True)])
def test_sharded_sorter_follows_key_order(device, kind, descending):
    """With a key order set on the engine's sorter the split digit, the shard boundaries and the
    local passes all work on the order-preserving image: rank r holds the r-th quarter of that
    order and the shards concatenate to the stable sort by image."""
    import torch
    from radx_amd.sharded import HipEngine, ShardedSorter
    from test_gpu_sort import ordered_image
    world, n_local = 4, 30000 + 5
    cap = ShardedSorter.suggest_capacity(n_local, 0.3)
    eng = HipEngine(cap, 4, 0)
    eng.sorter.set_key_order(kind, descending)
    inputs = [make_keys("uniform", n_local, 4, seed=400 + r) for r in range(world)]
    split_bufs, starts = [], []
    for r in range(world):
        src = to_dev(inputs[r], device)
        dst = torch.empty_like(src)
        starts.append(split_starts(eng, src, dst, n_local, 2))
        split_bufs.append(dst)
    whole = []
    for r in range(world):
        keys = torch.zeros(cap, dtype=torch.int32, device=device)
        keys[:n_local] = to_dev(inputs[r], device)
        work = torch.zeros(cap, dtype=torch.int32, device=device)
        split_bufs[r] = work
        sharder = ShardedSorter(eng, dist=ReplayDist(r, world, split_bufs, starts))
        shard, total, _, _ = sharder.sort(keys, work, n_local)
        torch.cuda.synchronize()
        got = to_host(shard).copy()
        assert np.all((ordered_image(got, kind, descending) >> np.uint32(30)) == r)
        whole.append(got)
    eng.close()
    everything = np.concatenate(inputs)
    expect = everything[np.argsort(ordered_image(everything, kind, descending), kind="stable")]
    assert np.array_equal(np.concatenate(whole), expect)


def test_run_starts_stay_on_the_device_and_table_views(device):
    """radx_hip_digit_starts_async copies the 17 run starts in stream order into device memory (what
    the multi-GPU hosts gather without a host wait), equal to the synchronising getter; and
    radx_hip_device_tables names the tables' device addresses: one complete offsets table for
    scan-free passes, offsets + group bases where the scan kernels ran."""
    import torch
    import radx_amd
    n = 100 * 4096 + 17
    s = radx_amd.Sorter(n, 4, 0)
    src = to_dev(make_keys("hash", n, 4, seed=5), device)
    dst = torch.empty_like(src)
    out = torch.full((18,), -1, dtype=torch.int32, device=device)
    s.msd_split(src, dst, 3)
    s.digit_starts_into(out)
    torch.cuda.synchronize()
    assert np.array_equal(out[:17].cpu().numpy().view(np.uint32), s.digit_starts())
    assert int(out[17].item()) == -1                      # exactly 17 words are written
    counts, offsets, bases, per_group = s.device_tables(n)
    assert counts and offsets and bases is None and per_group == 256   # 101 tiles: scan-free, complete table
    s.tune(True, scan_kernels=True)
    s.msd_split(src, dst, 3)
    counts, offsets, bases, _ = s.device_tables(n)
    assert counts and offsets and bases
    with pytest.raises(radx_amd.RadxError):
        s.device_tables(n - 1)                            # no tables for another element count
    with pytest.raises(ValueError):
        s.digit_starts_into(torch.empty(8, dtype=torch.int32, device=device))
    s.close()
