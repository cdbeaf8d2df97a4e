"""Seeded random stress of the C ABI on the GPU: random sizes, key widths, payloads, bit ranges,
key orders and distributions (heavy skew included), every result compared bit-for-bit with
numpy's stable sort of the same field."""
import os

import numpy as np
import pytest

from datasets import DISTRIBUTIONS, make_keys
from test_gpu_sort import ordered_image, to_dev, to_host

pytestmark = pytest.mark.gpu


def skewed_keys(rng, n, key_bytes):
    """A few heavy values (so that digit runs cross tile boundaries with many equal keys) plus noise."""
    dt = np.uint32 if key_bytes == 4 else np.uint64
    heavy = rng.integers(0, np.iinfo(dt).max, size=int(rng.integers(1, 5)), dtype=dt)
    keys = heavy[rng.integers(0, heavy.size, size=n)]
    noise = rng.random(n) < rng.choice([0.0, 0.001, 0.1, 0.5])
    keys[noise] = rng.integers(0, np.iinfo(dt).max, size=int(noise.sum()), dtype=dt)
    return keys


@pytest.mark.parametrize("seed", range(int(os.environ.get("RADX_STRESS_SEEDS", "12"))))  # 6 sorts per seed
def test_random_sorts(device, seed):
    import torch
    import radx_amd
    rng = np.random.default_rng(1000 + seed)
    for _ in range(6):
        key_bytes = int(rng.choice([4, 8]))
        with_vals = bool(rng.integers(0, 2))
        n = int(rng.choice([rng.integers(1, 5000), rng.integers(5000, 300000), 4096 * int(rng.integers(1, 40))]))
        if rng.random() < 0.5:
            keys = skewed_keys(rng, n, key_bytes)
        else:
            keys = make_keys(str(rng.choice(DISTRIBUTIONS)), n, key_bytes, seed=int(rng.integers(1, 1 << 30)))
        bits = key_bytes * 8
        whole = rng.random() < 0.5
        begin = 0 if whole else int(rng.integers(0, bits))
        end = bits if whole else int(rng.integers(begin, bits + 1))
        kind = str(rng.choice(["unsigned", "signed", "float"])) if whole else "unsigned"
        descending = bool(rng.integers(0, 2)) if whole else False
        vals = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)

        s = radx_amd.Sorter(n, key_bytes, 4 if with_vals else 0)
        s.set_key_order(kind, descending)
        s.tune(bool(rng.integers(0, 4)), scan_kernels=bool(rng.integers(0, 2)))  # mostly fused; scan-free or not
        k = to_dev(keys, device)
        swap = torch.empty_like(k)
        v = to_dev(vals, device) if with_vals else None
        vswap = torch.empty_like(v) if with_vals else None
        if whole:
            s.sort(k, swap, v, vswap)
            field = ordered_image(keys, kind, descending)
        else:
            s.sort_bits(k, swap, begin, end, v, vswap)
            field = (keys >> keys.dtype.type(begin)) & keys.dtype.type((1 << (end - begin)) - 1)
        torch.cuda.synchronize()
        order = np.argsort(field, kind="stable")
        what = "n=%d key_bytes=%d vals=%s bits=[%d,%d) %s desc=%s" % (n, key_bytes, with_vals, begin, end, kind, descending)
        assert np.array_equal(to_host(k, key_bytes), keys[order]), what
        if with_vals:
            assert np.array_equal(to_host(v), vals[order]), what
        s.close()


@pytest.mark.parametrize("key_bytes,with_vals,tiles,dist", [
    (4, False, 2049, "skewed"), (4, True, 3000, "heavy"), (8, True, 2600, "few_distinct"),
    (8, False, 5000, "sorted"), (4, False, 8192, "all_equal"), (8, True, 8192, "heavy"), (4, True, 4097, "top_digit_only"),
    (4, False, 8193, "skewed"), (4, True, 12345, "heavy"), (8, True, 16384, "few_distinct"), (4, False, 16384, "all_equal"),
    (8, False, 9999, "top_digit_only")])
def test_scan_free_sizes_with_many_row_groups(device, key_bytes, with_vals, tiles, dist):
    """More than 32 groups of 64 table rows (above 2048 tiles): the scan-free passes then take
    their output bases from the scanning workgroup; above 8192 tiles the groups have 128 rows.
    Ragged sizes up to the last scan-free size (16384 tiles), skewed and degenerate keys whose digit runs span many tiles and row groups,
    against numpy's stable sort; once more with the scan kernels on the same input."""
    import torch
    import radx_amd
    rng = np.random.default_rng(tiles)
    tile = 4096 if key_bytes == 4 else 2048
    n = (tiles - 1) * tile + int(rng.integers(1, tile + 1))
    keys = skewed_keys(rng, n, key_bytes) if dist == "heavy" else make_keys(dist, n, key_bytes, seed=tiles)
    vals = np.arange(n, dtype=np.uint32)
    order = np.argsort(keys, kind="stable")
    s = radx_amd.Sorter(n, key_bytes, 4 if with_vals else 0)
    assert s.geometry(n)[1] == tiles
    for scan_kernels in (False, True):
        s.tune(True, scan_kernels=scan_kernels)
        k = to_dev(keys, device)
        swap = torch.empty_like(k)
        v = to_dev(vals, device) if with_vals else None
        vswap = torch.empty_like(v) if with_vals else None
        s.sort(k, swap, v, vswap)
        torch.cuda.synchronize()
        assert np.array_equal(to_host(k, key_bytes), keys[order]), "scan kernels: %s" % scan_kernels
        if with_vals:
            assert np.array_equal(to_host(v), vals[order].astype(np.uint32)), "scan kernels: %s" % scan_kernels
        del k, swap, v, vswap
    s.close()


@pytest.mark.parametrize("key_bytes,with_vals,n,dist", [
    (4, False, 1_500_007, "uniform"), (4, True, 1_000_003, "heavy"), (8, False, 1_048_576, "few_distinct"),
    (8, True, 900_001, "heavy"), (4, True, 393_216, "sorted"), (4, False, 4_194_304, "skewed")])
def test_sizes_whose_passes_store_nontemporally(device, key_bytes, with_vals, n, dist):
    """Passes that write between 3 and 16 MiB leave with nontemporal stores (and a staggered first
    round): every shape at such a size, the two ends of the range included, ties with payloads
    (stability), a signed descending order on the way — against numpy's stable sort, and once
    more with the scan kernels (ordinary stores) on the same input."""
    import torch
    import radx_amd
    rng = np.random.default_rng(n)
    assert 3 << 20 <= n * (key_bytes + (4 if with_vals else 0)) <= 16 << 20
    keys = skewed_keys(rng, n, key_bytes) if dist == "heavy" else make_keys(dist, n, key_bytes, seed=n % 1000)
    vals = np.arange(n, dtype=np.uint32)
    s = radx_amd.Sorter(n, key_bytes, 4 if with_vals else 0)
    for kind, descending in (("unsigned", False), ("signed", True)):
        order = np.argsort(ordered_image(keys, kind, descending), kind="stable")
        s.set_key_order(kind, descending)
        for scan_kernels in (False, True):
            s.tune(True, scan_kernels=scan_kernels)
            k = to_dev(keys, device)
            swap = torch.empty_like(k)
            v = to_dev(vals, device) if with_vals else None
            vswap = torch.empty_like(v) if with_vals else None
            s.sort(k, swap, v, vswap)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(k, key_bytes), keys[order]), (kind, descending, scan_kernels)
            if with_vals:
                assert np.array_equal(to_host(v), vals[order]), (kind, descending, scan_kernels)
    s.close()
