"""GPU parity tests: the HIP engine, called through the C ABI (radx_amd -> libradx_hip.so),
against the CPU oracle on the same seeded inputs.  Integer work: every comparison is bit-exact."""
import numpy as np
import pytest

import oracle
from datasets import DISTRIBUTIONS, EDGE_SIZES, make_keys

pytestmark = pytest.mark.gpu


def to_dev(a, device):
    import torch
    signed = a.view(np.int32 if a.itemsize == 4 else np.int64)
    return torch.from_numpy(np.ascontiguousarray(signed)).to(device)


def to_host(t, key_bytes=4):
    return t.cpu().numpy().view(np.uint32 if key_bytes == 4 else np.uint64)


def gpu_sort(device, keys, vals=None, sorter=None):
    import torch
    import radx_amd
    n = keys.size
    kb = keys.itemsize
    own = sorter is None
    if own:
        sorter = radx_amd.Sorter(max(n, 1), kb, 4 if vals is not None else 0)
    k = to_dev(keys, device) if n else torch.empty(4, dtype=torch.int32 if kb == 4 else torch.int64, device=device)
    swap = torch.empty_like(k)
    v = vswap = None
    if vals is not None:
        v = to_dev(vals, device) if n else torch.empty(4, dtype=torch.int32, device=device)
        vswap = torch.empty_like(v)
    sorter.sort(k, swap, v, vswap, n=n)
    torch.cuda.synchronize()
    if own:
        sorter.close()
    if vals is None:
        return to_host(k, kb)[:n]
    return to_host(k, kb)[:n], to_host(v, 4)[:n]


@pytest.mark.parametrize("n", EDGE_SIZES)
def test_u32_edge_sizes(device, n):
    keys = make_keys("uniform", n, 4, seed=n + 1)
    got = gpu_sort(device, keys)
    assert np.array_equal(got, oracle.stable_sort(keys))


@pytest.mark.parametrize("n", [0, 1, 2047, 2048, 2049, 5000, 2 * 2048 + 1])
def test_u64_edge_sizes(device, n):
    keys = make_keys("uniform", n, 8, seed=n + 2)
    got = gpu_sort(device, keys)
    assert np.array_equal(got, oracle.stable_sort(keys))


@pytest.mark.parametrize("dist", DISTRIBUTIONS)
@pytest.mark.parametrize("key_bytes", [4, 8])
def test_distributions(device, dist, key_bytes):
    n = (1 << 18) + 3
    keys = make_keys(dist, n, key_bytes, seed=3)
    got = gpu_sort(device, keys)
    assert np.array_equal(got, oracle.stable_sort(keys))


def check_geometry(s, n, cfg):
    """The engine's chunking is the reference's get_blocks_info for its grid (includes.glsl:162-173)."""
    tile, tiles = s.geometry(n)
    assert tile == cfg.wave * cfg.vec and tiles == cfg.grid
    for g in (0, cfg.grid // 2, cfg.grid - 1):
        b = oracle.ref.radx_ref_blocks_info(cfg, n, g)
        assert b.offset == min(g * tile, n) and b.limit == min((g + 1) * tile, n) and b.count == 1


# The three ways a pass gets its tables: histogram kernel + scan kernels every pass; histogram
# accumulated by the previous scatter + scan kernels; and scan-free (inputs of at most 16384 tiles
# with the fused histogram: no scan kernel, every scatter workgroup derives its own bases).
# "fused_self_sum" / "fused_scanner_self_sum" are the scan-free mode with its publishing workgroups
# switched off (all of them / only those of the group sums): the tile workgroups, or the scanning
# workgroup, then wait their bounded number of looks and sum the table themselves, the paths that
# in normal operation are only a safety net.
PASS_MODES = ["unfused", "fused_scan_kernels", "fused", "fused_self_sum", "fused_scanner_self_sum"]


def set_mode(s, mode):
    s.tune(fuse_histogram=mode != "unfused", scan_kernels=mode == "fused_scan_kernels",
           publish={"fused_self_sum": "none", "fused_scanner_self_sum": "bases"}.get(mode, "all"))
    return mode != "unfused"


@pytest.mark.parametrize("mode", PASS_MODES)
@pytest.mark.parametrize("n", [1000, 4096, 3 * 4096 + 17, 70000, 16 * 4096 + 1, 65 * 4096 + 3, 300 * 4096 + 5,
                               1027 * 4096 + 99])
def test_u32_every_pass_matches_oracle(device, n, mode):
    """Histogram table, offsets table and scattered output after EACH pass, in all three modes.
    With scan kernels the two largest sizes have more than 256 tiles, i.e. more than one scan
    workgroup: the second-level scan kernel and the two-part offsets (within-group scan + group
    base) are compared table for table with the reference's single-workgroup partition
    (shaders-sdk/radix/partition.comp:29-68), 2 and 5 scan groups.  Scan-free, the sizes cover
    1, 2, 5 and 17 row groups of 64 tiles (and a last group of one tile)."""
    import torch
    import radx_amd
    keys = make_keys("hash", n, 4, seed=11)
    cfg = oracle.engine_config(n, 4)
    s = radx_amd.Sorter(n, 4, 0)
    fused = set_mode(s, mode)
    check_geometry(s, n, cfg)
    cur = to_dev(keys, device)
    nxt = torch.empty_like(cur)
    host = keys
    for p in range(8):
        if fused:
            s.run_pass_fused(cur, nxt, 4 * p, 4, 4 * (p + 1) if p < 7 else -1, 4, use_fused_counts=p > 0)
        else:
            s.run_pass(cur, nxt, bit=4 * p, num_bits=4)
        counts, offsets = s.tables(n)
        c, part, out = oracle.one_pass(cfg, host, p)
        assert np.array_equal(counts, c), "histogram table, pass %d" % p
        assert np.array_equal(offsets, part), "offset table, pass %d" % p
        assert np.array_equal(to_host(nxt), out), "scattered keys, pass %d" % p
        starts = s.digit_starts()
        assert starts[16] == n and np.array_equal(starts[:16], part[0])
        cur, nxt, host = nxt, cur, out
    s.close()
    assert np.array_equal(host, np.sort(keys))


@pytest.mark.parametrize("mode", PASS_MODES)
@pytest.mark.parametrize("n", [777, 2048, 5 * 2048 + 9, 9 * 2048 + 1, 300 * 2048 + 5])
def test_u64_kv_every_pass_matches_oracle(device, n, mode):
    import torch
    import radx_amd
    keys = make_keys("few_distinct", n, 8, seed=13)
    keys[::3] = make_keys("uniform", n, 8, seed=14)[::3]
    vals = np.arange(n, dtype=np.uint32)
    cfg = oracle.engine_config(n, 8)
    s = radx_amd.Sorter(n, 8, 4)
    fused = set_mode(s, mode)
    check_geometry(s, n, cfg)
    ck, cv = to_dev(keys, device), to_dev(vals, device)
    nk, nv = torch.empty_like(ck), torch.empty_like(cv)
    hk, hv = keys, vals
    for p in range(16):
        if fused:
            s.run_pass_fused(ck, nk, 4 * p, 4, 4 * (p + 1) if p < 15 else -1, 4, use_fused_counts=p > 0,
                             vals_in=cv, vals_out=nv)
        else:
            s.run_pass(ck, nk, bit=4 * p, num_bits=4, vals_in=cv, vals_out=nv)
        counts, offsets = s.tables(n)
        c, part, ok, ov = oracle.one_pass(cfg, hk, p, hv)
        assert np.array_equal(counts, c) and np.array_equal(offsets, part), "tables, pass %d" % p
        assert np.array_equal(to_host(nk, 8), ok) and np.array_equal(to_host(nv, 4), ov), "pass %d" % p
        ck, nk, cv, nv, hk, hv = nk, ck, nv, cv, ok, ov
    s.close()


@pytest.mark.parametrize("tiles,mode", [(8192, "fused"), (8193, "fused"), (8193, "fused_self_sum"),
                                        (8200, "fused_scanner_self_sum"), (16384, "fused"), (16385, "fused")])
def test_u64_passes_around_the_scan_free_limits(device, tiles, mode):
    """8192 tiles: the last size with groups of 64 table rows (128 groups).  8193: the first with
    groups of 128 rows (65 groups, the last one of a single row), also with nothing published
    (every tile workgroup sums the table itself) and with no group sums published (the scanning
    workgroup sums them).  16384: the last scan-free size (128 groups of 128).  16385: one more
    than the scan-free mode takes, so the default (fused) mode runs the scan kernels with 65 scan
    groups.  Tables, run starts and output of the first passes against the oracle."""
    import torch
    import radx_amd
    n = (tiles - 1) * 2048 + 77
    keys = make_keys("hash", n, 8, seed=tiles)
    cfg = oracle.engine_config(n, 8)
    s = radx_amd.Sorter(n, 8, 0)
    set_mode(s, mode)
    assert s.geometry(n)[1] == tiles
    cur = to_dev(keys, device)
    nxt = torch.empty_like(cur)
    host = keys
    for p in range(3 if tiles <= 8200 else 2):
        s.run_pass_fused(cur, nxt, 4 * p, 4, 4 * (p + 1), 4, use_fused_counts=p > 0)
        counts, offsets = s.tables(n)
        assert (s.device_tables(n)[2] is None) == (tiles <= 16384)   # complete table <=> scan-free
        c, part, out = oracle.one_pass(cfg, host, p)
        assert np.array_equal(counts, c) and np.array_equal(offsets, part), "tables, pass %d" % p
        assert np.array_equal(to_host(nxt, 8), out), "scattered keys, pass %d" % p
        starts = s.digit_starts()
        assert starts[16] == n and np.array_equal(starts[:16], part[0])
        cur, nxt, host = nxt, cur, out
    s.close()


@pytest.mark.parametrize("scan_kernels", [False, True])
@pytest.mark.parametrize("dist", ["skewed", "all_equal", "sorted", "top_digit_only"])
def test_fused_histogram_on_skewed_inputs(device, dist, scan_kernels):
    """Every scatter pass's accumulated next-pass histogram equals counting on its output, for
    inputs whose digit runs are long (one run can fill whole destination blocks and cross row
    groups), with scan kernels and scan-free."""
    import torch
    import radx_amd
    n = 140 * 4096 + 333
    keys = make_keys(dist, n, 4, seed=17)
    cfg = oracle.engine_config(n, 4)
    s = radx_amd.Sorter(n, 4, 0)
    s.tune(True, scan_kernels)
    cur = to_dev(keys, device)
    nxt = torch.empty_like(cur)
    host = keys
    for p in range(8):
        s.run_pass_fused(cur, nxt, 4 * p, 4, 4 * (p + 1) if p < 7 else -1, 4, use_fused_counts=p > 0)
        counts, offsets = s.tables(n)
        c, part, out = oracle.one_pass(cfg, host, p)
        assert np.array_equal(counts, c) and np.array_equal(offsets, part), "tables, pass %d" % p
        assert np.array_equal(to_host(nxt), out), "scattered keys, pass %d" % p
        cur, nxt, host = nxt, cur, out
    s.close()


@pytest.mark.parametrize("mode", PASS_MODES)
def test_result_independent_of_tuning(device, mode):
    import radx_amd
    n = 300007
    keys = make_keys("uniform", n, 4, seed=8)
    s = radx_amd.Sorter(n, 4, 0)
    set_mode(s, mode)
    assert np.array_equal(gpu_sort(device, keys, sorter=s), np.sort(keys))
    s.close()


@pytest.mark.parametrize("key_bytes", [4, 8])
@pytest.mark.parametrize("dist", ["few_distinct", "hash", "all_equal", "skewed"])
def test_key_value_stable(device, key_bytes, dist):
    n = 100003
    keys = make_keys(dist, n, key_bytes, seed=21)
    vals = np.arange(n, dtype=np.uint32)
    k, v = gpu_sort(device, keys, vals)
    wk, wv = oracle.stable_sort_kv(keys, vals)
    assert np.array_equal(k, wk) and np.array_equal(v, wv)


def test_sorter_reuse_and_smaller_n(device):
    import radx_amd
    s = radx_amd.Sorter(50000, 4, 0)
    for n in (50000, 4097, 1, 33333):
        keys = make_keys("uniform", n, 4, seed=n)
        assert np.array_equal(gpu_sort(device, keys, sorter=s), np.sort(keys))
    s.close()


def test_sort_bits_subrange(device):
    import torch
    import radx_amd
    n = 40000
    keys = make_keys("uniform", n, 4, seed=5)
    s = radx_amd.Sorter(n, 4, 0)
    for lo, hi in [(0, 32), (0, 29), (8, 20), (4, 8), (12, 12), (0, 3)]:
        k = to_dev(keys, device)
        swap = torch.empty_like(k)
        s.sort_bits(k, swap, lo, hi)
        torch.cuda.synchronize()
        field = (keys >> np.uint32(lo)) & np.uint32((1 << (hi - lo)) - 1)
        want = keys[np.argsort(field, kind="stable")]
        assert np.array_equal(to_host(k), want), (lo, hi)
    s.close()


def test_generators_checksum_and_check_sorted(device):
    import torch
    import radx_amd
    n = 100000
    k32 = torch.empty(n, dtype=torch.int32, device=device)
    radx_amd.generate_u32(k32, seed=42)
    torch.cuda.synchronize()
    h32 = oracle.generate_u32(n, 42)
    assert np.array_equal(to_host(k32), h32)
    assert radx_amd.checksum(k32) == oracle.checksum(h32)
    assert radx_amd.check_sorted(k32) == oracle.violations(h32)

    k64 = torch.empty(n, dtype=torch.int64, device=device)
    v = torch.empty(n, dtype=torch.int32, device=device)
    radx_amd.generate_u64(k64, v, seed=43)
    torch.cuda.synchronize()
    h64, hv = oracle.generate_u64(n, 43)
    assert np.array_equal(to_host(k64, 8), h64) and np.array_equal(to_host(v), hv)
    assert radx_amd.checksum(k64) == oracle.checksum(h64)
    assert radx_amd.check_sorted(k64, v) == oracle.violations(h64, hv)


def test_argument_errors(device):
    import torch
    import radx_amd
    s = radx_amd.Sorter(1000, 4, 0)
    k = torch.zeros(2000, dtype=torch.int32, device=device)
    swap = torch.zeros(2000, dtype=torch.int32, device=device)
    with pytest.raises(radx_amd.RadxError):
        s.sort(k, swap, n=2000)          # n > max_n
    with pytest.raises(radx_amd.RadxError):
        s.sort(k, k, n=1000)             # keys == swap
    with pytest.raises(radx_amd.RadxError):
        s.sort(k[1:], swap, n=1000)      # misaligned keys
    with pytest.raises(radx_amd.RadxError):
        s.sort(k, swap, vals=k, vals_swap=swap, n=1000)  # payload on a keys-only sorter
    with pytest.raises(radx_amd.RadxError):
        s.run_pass(k, swap, bit=30, num_bits=4, n=1000)  # digit outside the key
    s.sort(k, swap, n=0)                 # empty input is a no-op
    s.close()
    with pytest.raises(radx_amd.RadxError):
        radx_amd.Sorter(1 << 33, 4, 0)   # 32-bit element count, as in the reference
    with pytest.raises(radx_amd.RadxError):
        radx_amd.Sorter(1000, 3, 0)


def test_two_sorters_two_streams(device):
    import torch
    import radx_amd
    n = 300000
    a, b = make_keys("uniform", n, 4, seed=1), make_keys("uniform", n, 4, seed=2)
    sa, sb = radx_amd.Sorter(n), radx_amd.Sorter(n)
    st_a, st_b = torch.cuda.Stream(), torch.cuda.Stream()
    ka, kb = to_dev(a, device), to_dev(b, device)
    wa, wb = torch.empty_like(ka), torch.empty_like(kb)
    torch.cuda.synchronize()
    sa.sort(ka, wa, stream=st_a)
    sb.sort(kb, wb, stream=st_b)
    torch.cuda.synchronize()
    assert np.array_equal(to_host(ka), np.sort(a)) and np.array_equal(to_host(kb), np.sort(b))
    sa.close()
    sb.close()


def test_medium_2_24_vs_cpu(device):
    n = 1 << 24
    keys = make_keys("hash", n, 4, seed=99)
    got = gpu_sort(device, keys)
    assert np.array_equal(got, oracle.parallel_sort(keys))


def test_config2_2_28_properties_and_full_equality(device):
    """The 2^28-key uint32 benchmark configuration: sortedness and a preserved multiset checksum
    (the size-independent properties), then the whole array bit for bit against the CPU sort of
    the same generated keys."""
    import torch
    import radx_amd
    n = 1 << 28
    k = torch.empty(n, dtype=torch.int32, device=device)
    swap = torch.empty_like(k)
    radx_amd.generate_u32(k, seed=7)
    before = radx_amd.checksum(k)
    s = radx_amd.Sorter(n)
    s.sort(k, swap)
    torch.cuda.synchronize()
    assert radx_amd.check_sorted(k) == 0
    assert radx_amd.checksum(k) == before
    s.close()
    want = oracle.parallel_sort(oracle.generate_u32(n, 7))
    assert np.array_equal(to_host(k), want)


def ordered_image(keys, kind, descending):
    """numpy restatement of the order-preserving map: unsigned image whose ascending order is the wanted order."""
    bits = keys.itemsize * 8
    dt = keys.dtype
    sign = dt.type(1 << (bits - 1))
    full = dt.type((1 << bits) - 1)
    k = keys.copy()
    if kind == "float":
        neg = (k >> dt.type(bits - 1)).astype(bool)
        k = np.where(neg, k ^ full, k ^ sign)
    elif kind == "signed":
        k = k ^ sign
    if descending:
        k = k ^ full
    return k.astype(dt)


@pytest.mark.parametrize("key_bytes", [4, 8])
@pytest.mark.parametrize("kind", ["unsigned", "signed", "float"])
@pytest.mark.parametrize("descending", [False, True])
def test_key_orders(device, key_bytes, kind, descending):
    """Descending / signed / float orders (declared, unused RadicePropStruct of the reference):
    the result is the stable sort of the keys by their order-preserving unsigned image, payload
    following.  No reference output exists for these orders: pinned to numpy only."""
    import torch
    import radx_amd
    n = 70001
    keys = make_keys("uniform", n, key_bytes, seed=31)
    keys[::5] = make_keys("few_distinct", n, key_bytes, seed=32)[::5]      # ties
    keys[:8] = np.array([0, 1 << (key_bytes * 8 - 1), (1 << (key_bytes * 8)) - 1, 1, 2, 3, 4, 5], dtype=keys.dtype)
    if kind == "float":  # include +-0, +-inf, NaNs
        ft = np.float32 if key_bytes == 4 else np.float64
        special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 1.5, -1.5], dtype=ft).view(keys.dtype)
        keys[8:16] = special
    vals = np.arange(n, dtype=np.uint32)
    s = radx_amd.Sorter(n, key_bytes, 4)
    s.set_key_order(kind, descending)
    k, v = gpu_sort(device, keys, vals, sorter=s)
    s.close()
    order = np.argsort(ordered_image(keys, kind, descending), kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])
    if kind == "signed" and not descending:
        st = np.int32 if key_bytes == 4 else np.int64
        assert np.array_equal(k.view(st), np.sort(keys.view(st), kind="stable"))
    if kind == "float":
        ft = np.float32 if key_bytes == 4 else np.float64
        f = k.view(ft)
        finite = f[np.isfinite(f)]
        assert np.all(np.diff(finite) <= 0) if descending else np.all(np.diff(finite) >= 0)


@pytest.mark.parametrize("row_words", [1, 3, 4, 12, 16])
def test_index_path_sort_then_gather(device, row_words):
    """indiction -> (key, index) sort -> permutation: wide payload rows moved once, against the
    oracle's restatement of indiction.comp / permutation.comp and a numpy stable argsort."""
    import torch
    import radx_amd
    rng = np.random.default_rng(row_words)
    n = 30000 + row_words
    keys = make_keys("few_distinct", n, 4, seed=4)
    keys[::2] = make_keys("uniform", n, 4, seed=5)[::2]
    rows = rng.integers(0, 1 << 32, size=(n, row_words), dtype=np.uint32)
    d_rows = torch.from_numpy(rows.view(np.int32)).to(device)
    d_cache = torch.empty_like(d_rows)
    d_ptrs = torch.empty(n, dtype=torch.int32, device=device)
    radx_amd.indiction(d_ptrs, d_cache, d_rows)
    torch.cuda.synchronize()
    ptrs, cache = oracle.indiction(rows)
    assert np.array_equal(to_host(d_ptrs), ptrs) and np.array_equal(d_cache.cpu().numpy().view(np.uint32), cache)
    s = radx_amd.Sorter(n, 4, 4)
    dk = to_dev(keys, device)
    s.sort(dk, torch.empty_like(dk), d_ptrs, torch.empty_like(d_ptrs))
    radx_amd.permutation(d_rows, d_cache, d_ptrs)
    torch.cuda.synchronize()
    s.close()
    hk, hidx = oracle.sort(oracle.engine_config(n, 4), keys, ptrs)
    assert np.array_equal(to_host(dk), hk) and np.array_equal(to_host(d_ptrs), hidx)
    want = oracle.permutation(cache, hidx)
    assert np.array_equal(d_rows.cpu().numpy().view(np.uint32), want)
    assert np.array_equal(want, rows[np.argsort(keys, kind="stable")])


@pytest.mark.parametrize("n", [(1 << 31) + 12345, (1 << 32) - 1])
def test_largest_sizes_properties(device, n):
    """Above 2^31 and at the largest element count the 32-bit index arithmetic allows (the
    reference's NumElements is a 32-bit uniform): sortedness and a preserved multiset checksum.
    Also the only sizes at which the second-level scan needs more than one round."""
    import torch
    import radx_amd
    k = torch.empty(n, dtype=torch.int32, device=device)
    swap = torch.empty_like(k)
    radx_amd.generate_u32(k, seed=77)
    before = radx_amd.checksum(k)
    s = radx_amd.Sorter(n)
    s.sort(k, swap)
    torch.cuda.synchronize()
    assert radx_amd.check_sorted(k) == 0
    assert radx_amd.checksum(k) == before
    # the extremes of the sorted array against the generator's known range
    lo, hi = int(k[0].item()) & 0xFFFFFFFF, int(k[-1].item()) & 0xFFFFFFFF
    assert lo <= hi
    s.close()
    del k, swap
    torch.cuda.empty_cache()


def test_config3_2_30_full_equality_vs_cpu(device):
    """The 2^30-key uint32 configuration (the reference README's headline size): bit-exact against the CPU sort."""
    import torch
    import radx_amd
    n = 1 << 30
    k = torch.empty(n, dtype=torch.int32, device=device)
    radx_amd.generate_u32(k, seed=5)
    torch.cuda.synchronize()
    host = oracle.generate_u32(n, 5)
    s = radx_amd.Sorter(n)
    s.sort(k, torch.empty_like(k))
    torch.cuda.synchronize()
    got = to_host(k)
    want = oracle.parallel_sort(host)
    assert np.array_equal(got, want)
    s.close()


@pytest.mark.parametrize("log2n", [28, 30])
def test_config4_u64_kv_properties(device, log2n):
    """The key-value benchmark configuration (uint64 key + uint32 payload = original index) at 2^28
    and at its full 2^30 pairs: keys sorted, payloads strictly increasing within equal-key runs
    (stability), every output key equals the input key its payload points to, and the key
    multiset checksum is preserved."""
    import torch
    import radx_amd
    n = 1 << log2n
    k = torch.empty(n, dtype=torch.int64, device=device)
    v = torch.empty(n, dtype=torch.int32, device=device)
    radx_amd.generate_u64(k, v, seed=9)
    k &= 0x0000FFFFFFFF0000 | (0xFF << 56)  # thin the key space so equal keys occur
    orig = k.clone()
    before = radx_amd.checksum(k)
    s = radx_amd.Sorter(n, 8, 4)
    s.sort(k, torch.empty_like(k), v, torch.empty_like(v))
    torch.cuda.synchronize()
    assert radx_amd.check_sorted(k, v) == 0          # order + stability (payload = input index)
    assert radx_amd.checksum(k) == before
    assert bool((orig[v.long()] == k).all().item())  # keys_in[payload[i]] == keys_out[i]
    s.close()
    del k, v, orig
    torch.cuda.empty_cache()


def test_sort_is_graph_capturable(device):
    """radx_hip_sort only enqueues kernels and, for small inputs, one memset of its own tables (no
    allocation, copy or synchronisation), so a caller can capture it into a HIP graph and replay
    it on new contents of the same buffers."""
    import torch
    import radx_amd
    n = 200000 + 7
    s = radx_amd.Sorter(n, 4, 0)
    k = torch.zeros(n, dtype=torch.int32, device=device)
    swap = torch.empty_like(k)
    s.sort(k, swap)                       # warm-up outside capture (code object load)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s.sort(k, swap)
    for seed in (1, 2, 3):
        keys = make_keys("uniform", n, 4, seed=seed)
        k.copy_(to_dev(keys, device))
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(to_host(k), np.sort(keys)), seed
    s.close()


def test_profiling_for_a_long_time_without_reading(device):
    """More launches than the pending-timings list holds (1024) before kernel_times is read: the
    finished ones are folded in along the way, nothing is lost and no sort fails."""
    import torch
    import radx_amd
    n = 300 * 4096 + 1  # 2 scan groups: 25 launches per sort
    keys = make_keys("uniform", n, 4, seed=3)
    s = radx_amd.Sorter(n, 4, 0)
    s.tune(True, scan_kernels=True)
    s.profile(True)
    k = to_dev(keys, device)
    swap = torch.empty_like(k)
    sorts = 90
    for _ in range(sorts):
        s.sort(k, swap)
    torch.cuda.synchronize()
    t = s.kernel_times()
    assert t["scatter"][1] == 8 * sorts and t["hist"][1] == sorts and t["scan"][1] == 8 * sorts
    assert t["scatter"][0] > 0
    assert np.array_equal(to_host(k), np.sort(keys))
    s.close()


@pytest.mark.parametrize("key_bytes", [4, 8])
@pytest.mark.parametrize("with_vals", [False, True])
def test_single_tile_inputs(device, key_bytes, with_vals):
    """Inputs of at most one tile are sorted by one launch of one workgroup (all passes in LDS):
    every size class up to the tile, ties with payloads (stability), bit sub-ranges with odd pass
    counts, key orders, the packed destination of sort_bits_into at an odd element offset — and
    the same inputs once more through the multi-launch passes, which must agree bit for bit."""
    import torch
    import radx_amd
    tile = 4096 if key_bytes == 4 else 2048
    bits = key_bytes * 8
    s = radx_amd.Sorter(tile, key_bytes, 4 if with_vals else 0)
    for n in (1, 2, 3, 63, 64, 255, 257, 1000, tile - 1, tile):
        for dist, (lo, hi), (kind, desc) in (("uniform", (0, bits), ("unsigned", False)),
                                             ("few_distinct", (0, bits), ("signed", True)),
                                             ("skewed", (4, 17), ("unsigned", False)),
                                             ("all_equal", (bits - 3, bits), ("unsigned", False)),
                                             ("hash", (0, bits), ("float", False)),
                                             ("uniform", (8, 8), ("unsigned", False))):
            keys = make_keys(dist, n, key_bytes, seed=n + lo)
            vals = np.arange(n, dtype=np.uint32)[::-1].copy()
            whole = (lo, hi) == (0, bits)
            if whole:
                field = ordered_image(keys, kind, desc)
            else:
                field = (keys >> keys.dtype.type(lo)) & keys.dtype.type((1 << (hi - lo)) - 1)
            order = np.argsort(field, kind="stable")
            results = []
            for scan_kernels in (False, True):   # one launch of one workgroup / the pass structure
                s.tune(True, scan_kernels=scan_kernels)
                s.set_key_order(kind if whole else "unsigned", desc if whole else False)
                k = to_dev(keys, device) if n >= 4 else to_dev(np.concatenate([keys, np.zeros(4, keys.dtype)]), device)
                swap = torch.empty_like(k)
                v = vswap = None
                if with_vals:
                    v = to_dev(vals if n >= 4 else np.concatenate([vals, np.zeros(4, np.uint32)]), device)
                    vswap = torch.empty_like(v)
                if whole:
                    s.sort(k, swap, v, vswap, n=n)
                else:
                    s.sort_bits(k, swap, lo, hi, v, vswap, n=n)
                torch.cuda.synchronize()
                got = (to_host(k, key_bytes)[:n], to_host(v)[:n] if with_vals else None)
                assert np.array_equal(got[0], keys[order]), (n, dist, lo, hi, kind, desc, scan_kernels)
                if with_vals:
                    assert np.array_equal(got[1], vals[order]), (n, dist, lo, hi, kind, desc, scan_kernels)
                results.append(got + (np.asarray(s.digit_starts()),))
            assert np.array_equal(results[0][0], results[1][0])
            if hi > lo:   # the 17 run starts of the last pass are the same either way
                assert np.array_equal(results[0][2], results[1][2]), (n, dist, lo, hi, results[0][2], results[1][2])
    # sort_bits_into: exact pass count, result stored at an odd element offset of a third buffer
    s.tune(True)
    s.set_key_order("unsigned", False)
    n = tile - 5
    keys = make_keys("few_distinct", n, key_bytes, seed=77)
    vals = np.arange(n, dtype=np.uint32)
    k = to_dev(np.concatenate([keys, np.zeros(5, keys.dtype)]), device)
    swap = torch.empty_like(k)
    out = torch.zeros(n + 8, dtype=k.dtype, device=device)
    v = to_dev(np.concatenate([vals, np.zeros(5, np.uint32)]), device) if with_vals else None
    vswap = torch.empty_like(v) if with_vals else None
    vout = torch.zeros(n + 8, dtype=torch.int32, device=device) if with_vals else None
    s.sort_bits_into(k, swap, out[3:], 0, bits - 4, v, vswap, vout[3:] if with_vals else None, n=n)
    torch.cuda.synchronize()
    field = keys & keys.dtype.type((1 << (bits - 4)) - 1)
    order = np.argsort(field, kind="stable")
    got = to_host(out, key_bytes)
    assert np.array_equal(got[3:3 + n], keys[order]) and not got[:3].any() and not got[3 + n:].any()
    if with_vals:
        assert np.array_equal(to_host(vout)[3:3 + n], vals[order])
    with pytest.raises(radx_amd.RadxError):   # one launch, keys never left the CU: no tables were made
        s.tables(n)
    s.close()
