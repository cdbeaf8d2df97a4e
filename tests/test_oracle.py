"""CPU tests of the oracle: the restated counting/partition/scattering loop must equal the
reference's own CPU comparator, std::stable_sort (src/test/sort.cpp:218-219), for every
geometry, and its tables must satisfy the identities the shaders rely on."""
import numpy as np
import pytest

import oracle
from datasets import DISTRIBUTIONS, EDGE_SIZES, make_keys

GEOMETRIES = {
    "head_2bit_108wg": lambda n, kb: oracle.head_config(),
    "sketched_4bit_108wg": lambda n, kb: oracle.Config(bits=4, grid=108, wave=64, vec=4),
    "turing_8bit_108wg": lambda n, kb: oracle.Config(bits=8, grid=108, wave=32, vec=4),
    "engine_4bit_tiles": lambda n, kb: oracle.engine_config(n, kb),
    "small_grid": lambda n, kb: oracle.Config(bits=4, grid=4, wave=64, vec=4),
}


def test_hash_known_values():
    # src/test/sort.cpp:226-234 evaluated by hand-independent numpy arithmetic
    def h(a):
        m = 0xFFFFFFFF
        a = ((a + 0x7ed55d16) + (a << 12)) & m
        a = ((a ^ 0xc761c23c) ^ (a >> 19)) & m
        a = ((a + 0x165667b1) + (a << 5)) & m
        a = ((a + 0xd3a2646c) ^ (a << 9)) & m
        a = ((a + 0xfd7046c5) + (a << 3)) & m
        a = ((a ^ 0xb55a4f09) ^ (a >> 16)) & m
        return a
    for x in [0, 1, 2, 0xFFFFFFFF, 0x12345678, 1 << 31]:
        assert oracle.ref.radx_ref_hash(x) == h(x)
    g = oracle.generate_u32(1000, 7)
    assert g[5] == h(5 ^ 7)


def test_blocks_info_partitions_the_input():
    # shaders-sdk/radix/includes.glsl:162-173: chunks are disjoint, ordered and cover [0, n)
    for n in [0, 1, 255, 256, 257, 1000, 27647, 27648, 27649, 1 << 20, (1 << 20) + 3]:
        for cfg in (oracle.head_config(), oracle.engine_config(n), oracle.Config(4, 7, 32, 8)):
            covered = 0
            for g in range(cfg.grid):
                b = oracle.ref.radx_ref_blocks_info(cfg, n, g)
                assert b.offset == min(covered, n) or b.offset == b.limit
                if b.limit > b.offset:
                    assert b.offset == covered
                    covered = b.limit
                    assert (b.limit - b.offset) <= b.count * cfg.wave * cfg.vec
            assert covered == n


@pytest.mark.parametrize("geom", sorted(GEOMETRIES))
@pytest.mark.parametrize("n", EDGE_SIZES)
def test_sort_equals_stable_sort_edge_sizes(geom, n):
    keys = make_keys("uniform", n, 4, seed=n + 1)
    cfg = GEOMETRIES[geom](n, 4)
    got = oracle.sort(cfg, keys)
    assert np.array_equal(got, oracle.stable_sort(keys))
    assert np.array_equal(got, np.sort(keys, kind="stable"))


@pytest.mark.parametrize("dist", DISTRIBUTIONS)
@pytest.mark.parametrize("key_bytes", [4, 8])
def test_sort_distributions(dist, key_bytes):
    n = 3 * 4096 + 17
    keys = make_keys(dist, n, key_bytes, seed=3)
    want = oracle.stable_sort(keys)
    for geom in ("head_2bit_108wg", "engine_4bit_tiles"):
        got = oracle.sort(GEOMETRIES[geom](n, key_bytes), keys)
        assert np.array_equal(got, want), geom


def test_result_is_geometry_independent_and_tables_consistent():
    n = 1000
    keys = make_keys("hash", n, 4, seed=11)
    cfg = oracle.Config(bits=4, grid=4, wave=64, vec=4)
    got, clog, plog = oracle.sort(cfg, keys, log=True)
    assert np.array_equal(got, np.sort(keys))
    assert clog.shape == (8, 4, 16)
    for p in range(8):
        assert clog[p].sum() == n
        # partition.comp:29-68: exclusive scan in (radix, workgroup) order
        flat = clog[p].T.reshape(-1).astype(np.uint64)
        want = np.concatenate([[0], np.cumsum(flat)[:-1]]).reshape(16, 4).T
        assert np.array_equal(plog[p].astype(np.uint64), want)
    assert np.array_equal(oracle.sort(oracle.head_config(), keys), got)


def test_per_pass_output_is_stable_partition():
    n = 5000
    keys = make_keys("few_distinct", n, 4, seed=5)
    cfg = oracle.engine_config(n, 4)
    cur = keys
    for shift in range(8):
        c, p, out = oracle.one_pass(cfg, cur, shift)
        digit = (cur >> np.uint32(4 * shift)) & np.uint32(15)
        want = cur[np.argsort(digit, kind="stable")]
        assert np.array_equal(out, want)
        cur = out
    assert np.array_equal(cur, np.sort(keys))


@pytest.mark.parametrize("key_bytes", [4, 8])
def test_key_value_sort_is_stable(key_bytes):
    n = 10000
    keys = make_keys("few_distinct", n, key_bytes, seed=9)
    vals = np.arange(n, dtype=np.uint32)
    k, v = oracle.sort(oracle.engine_config(n, key_bytes), keys, vals)
    wk, wv = oracle.stable_sort_kv(keys, vals)
    assert np.array_equal(k, wk) and np.array_equal(v, wv)
    assert oracle.violations(k, v) == 0
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(v, order.astype(np.uint32))


def test_parallel_sort_and_checksums():
    keys = make_keys("hash", 1 << 20, 4, seed=1)  # config 1: 1 Mi uniform uint32 on the CPU path
    s = oracle.parallel_sort(keys)
    assert np.array_equal(s, oracle.stable_sort(keys))
    assert oracle.checksum(s) == oracle.checksum(keys)
    assert oracle.violations(s) == 0
    assert oracle.checksum(keys)[0] == int(keys.astype(np.uint64).sum())
    assert oracle.cpu.radx_cpu_threads() >= 1
    assert oracle.cpu.radx_cpu_pstl_backend() in (b"tbb", b"serial")


def test_index_path_indiction_permutation():
    # indiction.comp:27 / permutation.comp:27: sort (key, index), then gather rows through the index
    rng = np.random.default_rng(3)
    n = 5000
    keys = make_keys("few_distinct", n, 4, seed=2)
    rows = rng.integers(0, 1 << 32, size=(n, 5), dtype=np.uint32)
    ptrs, cache = oracle.indiction(rows)
    assert np.array_equal(ptrs, np.arange(n, dtype=np.uint32)) and np.array_equal(cache, rows)
    k, idx = oracle.sort(oracle.engine_config(n, 4), keys, ptrs)
    out = oracle.permutation(cache, idx)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(out, rows[order])
