// asan_driver.cpp — host-only sanitizer run of the CPU side: the oracle restatement
// (radx_oracle.c) and the CPU comparator (cpu_sort.cpp), built with
// -fsanitize=address,undefined.  TEST INFRASTRUCTURE ONLY (see radx_oracle.h).
//
// Every buffer is a heap allocation of exactly the size the call is documented to touch, so an
// access one element outside any of them is a sanitizer report; every result is also compared
// with std::stable_sort.  Geometries and sizes are those of tests/test_oracle.py: the
// reference's HEAD (2-bit digits, 108 workgroups: shaders-sdk/radix/includes.glsl:21-39,
// include/radx/radx_internal.hpp:143), the sketched 4-bit and the Turing 8-bit variants, the
// engine's one-tile-per-workgroup grid and a small grid.  Exit status 0 = every check passed
// (a sanitizer report aborts the process with its own non-zero status).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>
#include <vector>

#include "radx_oracle.h"

extern "C" {
void radx_cpu_stable_sort_u32(uint32_t* keys, uint64_t n);
void radx_cpu_stable_sort_u64(uint64_t* keys, uint64_t n);
void radx_cpu_parallel_sort_u32(uint32_t* keys, uint64_t n);
void radx_cpu_parallel_sort_u64(uint64_t* keys, uint64_t n);
int radx_cpu_threads(void);
}

namespace {

int failures = 0;
void expect(bool ok, const char* what, size_t n, const char* geom) {
    if (!ok) { ++failures; std::fprintf(stderr, "FAIL: %s (n = %zu, %s)\n", what, n, geom); }
}

// exactly n elements on the heap (n == 0: a one-byte block nobody may touch)
template <typename T>
std::unique_ptr<T[]> exact(size_t n) { return std::unique_ptr<T[]>(new T[n ? n : 0]); }

struct Geometry { const char* name; uint32_t bits, grid, wave, vec; };  // grid 0: ceil(n / tile)

radx_ref_config config_for(const Geometry& g, size_t n) {
    radx_ref_config c{g.bits, g.grid, g.wave, g.vec};
    if (!c.grid) c.grid = static_cast<uint32_t>(std::max<size_t>(1, (n + size_t(g.wave) * g.vec - 1) / (size_t(g.wave) * g.vec)));
    return c;
}

template <typename K>
void fill(K* keys, size_t n, uint32_t seed, int dist) {
    for (size_t i = 0; i < n; ++i) {
        const uint32_t lo = radx_ref_hash(static_cast<uint32_t>(i) ^ seed);
        const uint64_t wide = (static_cast<uint64_t>(radx_ref_hash(~lo)) << 32) | lo;
        K k = static_cast<K>(sizeof(K) == 8 ? wide : lo);
        if (dist == 1) k = static_cast<K>(k & static_cast<K>(0x0F00F00Full));            // few distinct values: ties
        if (dist == 2) k = static_cast<K>(~static_cast<K>(0));                            // all equal, all ones
        if (dist == 3) k = static_cast<K>(i);                                             // sorted
        if (dist == 4) k = static_cast<K>(static_cast<K>(i & 15u) << (sizeof(K) * 8 - 4)); // top digit only
        keys[i] = k;
    }
}

template <typename K>
void check_sort(const Geometry& g, size_t n, int dist, bool with_vals, bool with_log) {
    const radx_ref_config cfg = config_for(g, n);
    const uint32_t passes = sizeof(K) * 8 / cfg.bits;
    auto keys = exact<K>(n), swap = exact<K>(n);
    auto vals = exact<uint32_t>(with_vals ? n : 0), vswap = exact<uint32_t>(with_vals ? n : 0);
    fill(keys.get(), n, static_cast<uint32_t>(n * 31 + dist), dist);
    std::vector<K> want(keys.get(), keys.get() + n);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return want[a] < want[b]; });
    for (size_t i = 0; i < n && with_vals; ++i) vals[i] = static_cast<uint32_t>(i);
    const size_t table = size_t(cfg.grid) << cfg.bits;
    auto clog = exact<uint32_t>(with_log ? passes * table : 0), plog = exact<uint32_t>(with_log ? passes * table : 0);
    const int where = radx_ref_sort(&cfg, keys.get(), swap.get(), with_vals ? vals.get() : nullptr,
                                    with_vals ? vswap.get() : nullptr, sizeof(K), static_cast<uint32_t>(n), 0,
                                    with_log ? clog.get() : nullptr, with_log ? plog.get() : nullptr);
    const K* got = where ? swap.get() : keys.get();
    const uint32_t* gotv = where ? vswap.get() : vals.get();
    bool same = true;
    for (size_t i = 0; i < n && same; ++i) same = got[i] == want[order[i]] && (!with_vals || gotv[i] == order[i]);
    expect(same, "radx_ref_sort differs from std::stable_sort", n, g.name);
    expect(where == 0, "even pass count must leave the result in keys", n, g.name);
    if (with_log) {
        bool sums = true;
        for (uint32_t p = 0; p < passes && sums; ++p) {
            uint64_t total = 0;
            for (size_t i = 0; i < table; ++i) total += clog[p * table + i];
            sums = total == n;
        }
        expect(sums, "every pass's histogram table sums to n", n, g.name);
    }
    expect(radx_ref_violations(got, with_vals ? gotv : nullptr, n, sizeof(K)) == 0, "violations in a sorted array", n, g.name);
}

// one pass through the three kernels on their own, tables exactly [grid][1 << bits]
void check_single_pass(const Geometry& g, size_t n) {
    const radx_ref_config cfg = config_for(g, n);
    const size_t table = size_t(cfg.grid) << cfg.bits;
    auto in = exact<uint32_t>(n), out = exact<uint32_t>(n), counts = exact<uint32_t>(table), parts = exact<uint32_t>(table);
    fill(in.get(), n, 77, 0);
    radx_ref_counting(&cfg, in.get(), 4, static_cast<uint32_t>(n), 1, counts.get());
    radx_ref_partition(&cfg, counts.get(), parts.get());
    radx_ref_scattering(&cfg, in.get(), out.get(), nullptr, nullptr, 4, static_cast<uint32_t>(n), 1, parts.get());
    std::vector<uint32_t> want(in.get(), in.get() + n);
    const uint32_t mask = (1u << cfg.bits) - 1u, sh = cfg.bits;
    std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return ((a >> sh) & mask) < ((b >> sh) & mask); });
    expect(std::equal(want.begin(), want.end(), out.get()), "one pass differs from a stable sort on the digit", n, g.name);
    uint32_t covered = 0;
    for (uint32_t w = 0; w < cfg.grid; ++w) {
        const radx_ref_blocks b = radx_ref_blocks_info(&cfg, static_cast<uint32_t>(n), w);
        if (b.limit > b.offset) { expect(b.offset == covered, "chunks are contiguous", n, g.name); covered = b.limit; }
    }
    expect(covered == n, "chunks cover the input", n, g.name);
}

void check_index_path(size_t n, size_t elem_bytes) {
    auto rows = exact<unsigned char>(n * elem_bytes), cache = exact<unsigned char>(n * elem_bytes);
    auto ptrs = exact<uint32_t>(n);
    for (size_t i = 0; i < n * elem_bytes; ++i) rows[i] = static_cast<unsigned char>(radx_ref_hash(static_cast<uint32_t>(i)));
    std::vector<unsigned char> orig(rows.get(), rows.get() + n * elem_bytes);
    radx_ref_indiction(ptrs.get(), cache.get(), rows.get(), n, elem_bytes);
    std::reverse(ptrs.get(), ptrs.get() + n);
    radx_ref_permutation(rows.get(), cache.get(), ptrs.get(), n, elem_bytes);
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i)
        ok = std::memcmp(rows.get() + i * elem_bytes, orig.data() + (n - 1 - i) * elem_bytes, elem_bytes) == 0;
    expect(ok, "indiction + permutation through a reversed index", n, "index path");
}

template <typename K>
void check_cpu_comparators(size_t n) {
    auto a = exact<K>(n), b = exact<K>(n);
    fill(a.get(), n, 5, 1);
    std::copy(a.get(), a.get() + n, b.get());
    std::vector<K> want(a.get(), a.get() + n);
    std::stable_sort(want.begin(), want.end());
    if (sizeof(K) == 4) {
        radx_cpu_stable_sort_u32(reinterpret_cast<uint32_t*>(a.get()), n);
        radx_cpu_parallel_sort_u32(reinterpret_cast<uint32_t*>(b.get()), n);
    } else {
        radx_cpu_stable_sort_u64(reinterpret_cast<uint64_t*>(a.get()), n);
        radx_cpu_parallel_sort_u64(reinterpret_cast<uint64_t*>(b.get()), n);
    }
    expect(std::equal(want.begin(), want.end(), a.get()), "std::stable_sort(par) comparator", n, "cpu_sort");
    expect(std::equal(want.begin(), want.end(), b.get()), "parallel std::sort comparator", n, "cpu_sort");
}

}  // namespace

int main() {
    const Geometry geometries[] = {
        {"head 2-bit x 108 workgroups", 2, 108, 64, 4},
        {"sketched 4-bit x 108 workgroups", 4, 108, 64, 4},
        {"turing 8-bit x 108 workgroups", 8, 108, 32, 4},
        {"engine: one 4096-key tile per workgroup", 4, 0, 64, 64},
        {"small grid of 4", 4, 4, 64, 4},
    };
    const size_t sizes[] = {0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193,
                            12345, 3 * 4096 + 17};
    size_t runs = 0;
    for (const Geometry& g : geometries) {
        for (size_t n : sizes) {
            check_sort<uint32_t>(g, n, 0, false, false);
            check_single_pass(g, n);
            runs += 2;
        }
        for (int dist = 0; dist < 5; ++dist) {
            check_sort<uint32_t>(g, 3 * 4096 + 17, dist, true, dist == 0);
            check_sort<uint64_t>(g, 2 * 2048 + 9, dist, dist & 1, false);
            runs += 2;
        }
    }
    for (size_t n : {size_t(0), size_t(1), size_t(1000)})
        for (size_t eb : {size_t(4), size_t(12), size_t(64)}) { check_index_path(n, eb); ++runs; }
    for (size_t n : {size_t(0), size_t(1), size_t(100003), size_t(1) << 20}) {
        check_cpu_comparators<uint32_t>(n);
        check_cpu_comparators<uint64_t>(n);
        runs += 2;
    }
    // generators and checksum on exact buffers
    {
        const size_t n = 4097;
        auto k32 = exact<uint32_t>(n);
        auto k64 = exact<uint64_t>(n);
        auto v = exact<uint32_t>(n);
        radx_ref_generate_u32(k32.get(), n, 3);
        radx_ref_generate_u64(k64.get(), v.get(), n, 3);
        radx_ref_generate_u64(k64.get(), nullptr, n, 4);
        uint64_t c[3];
        radx_ref_checksum(k32.get(), n, 4, c);
        radx_ref_checksum(k64.get(), n, 8, c);
        expect(k32[5] == radx_ref_hash(5u ^ 3u), "generator formula", n, "generators");
        ++runs;
    }
    std::printf("asan_driver: %zu checks run on %d threads, %d failed\n", runs, radx_cpu_threads(), failures);
    return failures ? 1 : 0;
}
