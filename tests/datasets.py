"""Seeded key distributions shared by the CPU and GPU tests."""
import numpy as np

import oracle

# Sizes around the engine's tile (4096 uint32 / 2048 uint64) and the reference's 256-key tile.
EDGE_SIZES = [0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193,
              12345, 3 * 4096 + 17]

DISTRIBUTIONS = ["uniform", "hash", "iota_shuffle", "all_equal", "sorted", "reverse", "top_digit_only",
                 "bottom_digit_only", "extremes", "skewed", "few_distinct"]


def make_keys(dist, n, key_bytes=4, seed=1):
    """n keys of the named distribution as a uint32/uint64 numpy array."""
    dt = np.uint32 if key_bytes == 4 else np.uint64
    bits = key_bytes * 8
    top = (1 << bits) - 1
    rng = np.random.default_rng(seed)
    if n == 0:
        return np.empty(0, dtype=dt)
    if dist == "uniform":
        return rng.integers(0, top, size=n, dtype=dt, endpoint=True)
    if dist == "hash":  # the generator the harness and bench use (src/test/sort.cpp:226-234 applied to i ^ seed)
        if key_bytes == 4:
            return oracle.generate_u32(n, seed)
        return oracle.generate_u64(n, seed)[0]
    if dist == "iota_shuffle":  # the reference harness's input: 0..n-1 shuffled (src/test/sort.cpp:349-350)
        a = np.arange(n, dtype=dt)
        rng.shuffle(a)
        return a
    if dist == "all_equal":
        return np.full(n, 0x5A5A5A5A5A5A5A5A & top, dtype=dt)
    if dist == "sorted":
        return np.sort(rng.integers(0, top, size=n, dtype=dt, endpoint=True))
    if dist == "reverse":
        return np.sort(rng.integers(0, top, size=n, dtype=dt, endpoint=True))[::-1].copy()
    if dist == "top_digit_only":
        return (rng.integers(0, 16, size=n, dtype=dt) << dt(bits - 4)).astype(dt)
    if dist == "bottom_digit_only":
        return rng.integers(0, 16, size=n, dtype=dt)
    if dist == "extremes":
        return rng.choice(np.array([0, top, 1, top - 1], dtype=dt), size=n)
    if dist == "skewed":  # 90 % of the keys share every digit; stresses one bin
        a = rng.integers(0, top, size=n, dtype=dt, endpoint=True)
        a[rng.random(n) < 0.9] = dt(0x7777777777777777 & top)
        return a
    if dist == "few_distinct":
        return rng.choice(rng.integers(0, top, size=7, dtype=dt, endpoint=True), size=n)
    raise ValueError(dist)
