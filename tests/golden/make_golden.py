#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  The reference pins no results (it has no tests and never
compares its GPU and CPU arrays), so these fixtures come from this repository's CPU restatement
(oracle/radx_oracle.c), each one cross-checked here against std::stable_sort — the reference's own
CPU comparator (src/test/sort.cpp:218-219) — before it is written.

  python tests/golden/make_golden.py      (needs `make` to have built oracle/*.so)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from datasets import make_keys  # noqa: E402


def main():
    # (1) final arrays for edge sizes and distributions, uint32
    final = {}
    for dist, n in [("uniform", 1), ("uniform", 2), ("uniform", 255), ("uniform", 257), ("hash", 1000),
                    ("iota_shuffle", 1000), ("all_equal", 300), ("reverse", 500), ("top_digit_only", 600),
                    ("bottom_digit_only", 600), ("extremes", 400), ("skewed", 900), ("uniform", 4097)]:
        keys = make_keys(dist, n, 4, seed=42)
        out = oracle.sort(oracle.engine_config(n, 4), keys)
        assert np.array_equal(out, oracle.stable_sort(keys))
        assert np.array_equal(out, oracle.sort(oracle.head_config(), keys))
        final["%s_%d_in" % (dist, n)] = keys
        final["%s_%d_out" % (dist, n)] = out
    np.savez_compressed(os.path.join(HERE, "final_u32.npz"), **final)

    # (2) per-pass tables and outputs: n = 1000, 4 workgroups, 4-bit digits (and the same input
    #     through the 2-bit / 108-workgroup geometry the reference ships: same final array)
    keys = make_keys("hash", 1000, 4, seed=11)
    cfg = oracle.Config(bits=4, grid=4, wave=64, vec=4)
    snap = {"input": keys}
    cur = keys
    for p in range(8):
        c, part, out = oracle.one_pass(cfg, cur, p)
        snap["counts_%d" % p], snap["partitions_%d" % p], snap["keys_%d" % p] = c, part, out
        cur = out
    assert np.array_equal(cur, oracle.stable_sort(keys))
    head = oracle.sort(oracle.head_config(), keys)
    assert np.array_equal(head, cur)
    snap["head_2bit_108wg_final"] = head
    np.savez_compressed(os.path.join(HERE, "passes_u32_n1000_grid4.npz"), **snap)

    # (3) uint64 keys + uint32 payload, duplicate keys: stability
    k64 = make_keys("few_distinct", 700, 8, seed=5)
    vals = np.arange(700, dtype=np.uint32)
    ok, ov = oracle.sort(oracle.engine_config(700, 8), k64, vals)
    wk, wv = oracle.stable_sort_kv(k64, vals)
    assert np.array_equal(ok, wk) and np.array_equal(ov, wv)
    np.savez_compressed(os.path.join(HERE, "kv_u64_n700.npz"), keys=k64, vals=vals, keys_out=ok, vals_out=ov)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
