// radx_hip.hip — gfx950 (MI355X, CDNA4) LSD radix sort: kernels and the C ABI of include/radx_hip.h.
//
// One pass = histogram -> scan -> scatter on a 4-bit digit, the three phases the reference
// dispatches per pass (include/radx/radx_implement.inl:427-444; shaders-sdk/radix/counting.comp,
// partition.comp, scattering.comp).  The decomposition is this engine's own:
//
//   * the input is cut into tiles of 16 KiB of keys (4096 uint32 / 2048 uint64), one 256-thread
//     workgroup per tile, instead of the reference's fixed 108 workgroups looping over 256-key
//     tiles (include/radx/radx_internal.hpp:143, shaders-sdk/radix/includes.glsl:162-173);
//   * the tables are digit-major, counts[d][tile], so the reference's two-phase partition
//     (partition.comp:33-48 per radix over workgroups, :51-67 over radices) is ONE linear
//     exclusive scan of 16*num_tiles words;
//   * inside a tile every key gets its stable rank from per-thread digit counters in LDS
//     (packed 2 x 16 bit, laid out [digit & 7][thread]) and one block-wide scan over them, the
//     tile is written to LDS in sorted order, and each digit's run leaves the CU as contiguous
//     dword stores.  Address order is preserved per digit, which is the reference's rank
//     (scattering.comp:119-123,141-143).
//
// All cross-wave LDS hand-offs are separated by __syncthreads().
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "radx_hip.h"

namespace {

constexpr int kBlock = 256;           // threads per workgroup (4 waves of 64)
constexpr int kRadices = RADX_HIP_RADICES;
constexpr int kDigitBits = RADX_HIP_BITS_PER_PASS;
constexpr int kThreadDwords = 16;     // every thread owns 64 B of keys
constexpr int kTileDwords = kBlock * kThreadDwords;  // 16 KiB of keys per tile
constexpr int kRowStride = 20;        // LDS dwords per thread row: 16 + 4 pad (bank spread for b128 reads)
constexpr int kCtrRows = kRadices / 2;  // packed counters: digit d lives in row d&7, half d>>3
constexpr int kScanChunk = 2048;      // table words scanned by one workgroup (256 threads x 8)

template <typename K> struct KeyTraits;
template <> struct KeyTraits<uint32_t> { static constexpr int kpt = 16; };
template <> struct KeyTraits<uint64_t> { static constexpr int kpt = 8; };

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------

// Workgroups are dealt round-robin over the 8 XCDs, so workgroups b and b+8 share an L2.  This
// bijection gives each of those 8 groups one contiguous range of tiles: tiles that run next to
// each other in time on one XCD are neighbours in memory, and the two halves of an output line
// that straddles two tiles' runs meet in the same L2.  Speed only; any placement is correct.
__device__ __forceinline__ uint32_t tile_of_block(uint32_t bid, uint32_t nwg) {
    const uint32_t q = nwg >> 3, r = nwg & 7u, x = bid & 7u, k = bid >> 3;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + k;
}

template <typename K>
__device__ __forceinline__ uint32_t digit_of(K key, int bit, uint32_t mask) {
    return static_cast<uint32_t>(key >> bit) & mask;
}

// The 32-bit integer hash the reference harness carries for key generation
// (src/test/sort.cpp:226-234): six add/xor-shift rounds.
__host__ __device__ __forceinline__ uint32_t harness_hash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

template <typename K> __device__ __forceinline__ void unpack4(const uint4& v, K* out);
template <> __device__ __forceinline__ void unpack4<uint32_t>(const uint4& v, uint32_t* out) {
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <> __device__ __forceinline__ void unpack4<uint64_t>(const uint4& v, uint64_t* out) {
    out[0] = (static_cast<uint64_t>(v.y) << 32) | v.x;
    out[1] = (static_cast<uint64_t>(v.w) << 32) | v.z;
}

// Inclusive scan over the 64 lanes of a wave.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(v, off, 64);
        if (lane >= static_cast<uint32_t>(off)) v += up;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// histogram: counts[d][tile] = number of keys of `tile` whose digit is d
// (what shaders-sdk/radix/counting.comp:49-116 leaves in counts[g][r], per tile instead of per
// 108th of the input).  LDS bins are replicated 32x so that lane l only ever touches bank l&31:
// no bank conflict and no same-address serialisation however skewed the digits are.
// ---------------------------------------------------------------------------------------------
template <typename K>
__global__ __launch_bounds__(kBlock) void hist_kernel(const K* __restrict__ keys, uint32_t n, int bit,
                                                       uint32_t mask, uint32_t* __restrict__ counts,
                                                       uint32_t num_tiles) {
    constexpr int KPT = KeyTraits<K>::kpt;
    constexpr uint32_t TILE = kBlock * KPT;
    constexpr int PER_VEC = 16 / sizeof(K);
    __shared__ uint32_t bins[kRadices * 32];

    const uint32_t tid = threadIdx.x;
    const uint32_t tile = tile_of_block(blockIdx.x, num_tiles);
    bins[tid] = 0u;
    bins[tid + kBlock] = 0u;
    __syncthreads();

    const uint32_t base = tile * TILE;
    const uint32_t valid = min(TILE, n - base);
    const uint32_t col = tid & 31u;
    if (valid == TILE) {
        const uint4* src = reinterpret_cast<const uint4*>(keys + base);
        uint4 v[kThreadDwords / 4];
#pragma unroll
        for (int m = 0; m < kThreadDwords / 4; ++m) v[m] = src[m * kBlock + tid];
#pragma unroll
        for (int m = 0; m < kThreadDwords / 4; ++m) {
            K k[PER_VEC];
            unpack4<K>(v[m], k);
#pragma unroll
            for (int j = 0; j < PER_VEC; ++j) {
                const uint32_t d = digit_of<K>(k[j], bit, mask);
                __hip_atomic_fetch_add(&bins[d * 32u + col], 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    } else {
        for (uint32_t i = tid; i < valid; i += kBlock) {
            const uint32_t d = digit_of<K>(keys[base + i], bit, mask);
            __hip_atomic_fetch_add(&bins[d * 32u + col], 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();

    // bins[d*32 + c]: thread tid holds words tid (digit tid/32) and tid+256 (digit tid/32 + 8);
    // sum each aligned group of 32 lanes.
    uint32_t lo = bins[tid], hi = bins[tid + kBlock];
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        lo += __shfl_xor(lo, off, 64);
        hi += __shfl_xor(hi, off, 64);
    }
    if (col == 0u) {
        const uint32_t d = tid >> 5;
        counts[static_cast<size_t>(d) * num_tiles + tile] = lo;
        counts[static_cast<size_t>(d + 8u) * num_tiles + tile] = hi;
    }
}

// ---------------------------------------------------------------------------------------------
// scan: exclusive prefix sum over the digit-major table, i.e. over (digit, tile) in that order —
// the order shaders-sdk/radix/partition.comp:29-68 produces with its two phases.
// Kernel 1 scans chunks of 2048 words and leaves each chunk's total; kernel 2 (one workgroup)
// scans the totals.  The scatter kernel adds the two.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void scan_chunks_kernel(const uint32_t* __restrict__ counts,
                                                              uint32_t* __restrict__ offsets,
                                                              uint32_t* __restrict__ chunk_sums,
                                                              uint32_t len) {
    __shared__ uint32_t wave_tot[kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t base = blockIdx.x * kScanChunk + tid * 8u;
    uint32_t v[8];
    if (base + 8u <= len) {
        const uint4 a = *reinterpret_cast<const uint4*>(counts + base);
        const uint4 b = *reinterpret_cast<const uint4*>(counts + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (base + i < len) ? counts[base + i] : 0u;
    }
    uint32_t total = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) total += v[i];
    const uint32_t incl = wave_inclusive_scan(total, lane);
    if (lane == 63u) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        const uint32_t t = wave_tot[w];
        if (static_cast<uint32_t>(w) < wave) before += t;
        all += t;
    }
    uint32_t run = incl - total + before;
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { o[i] = run; run += v[i]; }
    if (base + 8u <= len) {
        *reinterpret_cast<uint4*>(offsets + base) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(offsets + base + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) if (base + i < len) offsets[base + i] = o[i];
    }
    if (tid == 0u) chunk_sums[blockIdx.x] = all;
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t* __restrict__ chunk_sums,
                                                          uint32_t num_chunks,
                                                          const uint32_t* __restrict__ offsets,
                                                          uint32_t num_tiles, uint32_t n,
                                                          uint32_t* __restrict__ digit_starts) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0u) carry_s = 0u;
    __syncthreads();
    for (uint32_t start = 0; start < num_chunks; start += 1024u) {
        const uint32_t i = start + tid;
        const uint32_t v = i < num_chunks ? chunk_sums[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v, lane);
        if (lane == 63u) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t t = wave_tot[w];
            if (static_cast<uint32_t>(w) < wave) before += t;
            all += t;
        }
        const uint32_t carry = carry_s;
        if (i < num_chunks) chunk_sums[i] = carry + before + incl - v;
        __syncthreads();
        if (tid == 0u) carry_s = carry + all;
        __syncthreads();
    }
    // Start of every digit's output run, for callers that split by digit (multi-GPU exchange).
    __threadfence_block();
    __syncthreads();
    if (tid < static_cast<uint32_t>(kRadices)) {
        const uint32_t idx = tid * num_tiles;
        digit_starts[tid] = offsets[idx] + chunk_sums[idx / kScanChunk];
    }
    if (tid == static_cast<uint32_t>(kRadices)) digit_starts[kRadices] = n;
}

// ---------------------------------------------------------------------------------------------
// scatter: stable partition of one tile by digit, written to offsets[d][tile] + rank
// (shaders-sdk/radix/scattering.comp:57-149; out index = localPartitions[d] + earlier keys of
// digit d in address order, :141-143).
//
//   1. coalesced 16-B loads, transposed through LDS so thread t holds the tile's keys
//      [t*KPT, (t+1)*KPT) — consecutive in address order (rows padded 16->20 dwords);
//   2. per key one returning LDS add on the thread's own packed counter: the old value is the
//      key's rank among the thread's earlier keys of that digit;
//   3. one block-wide exclusive scan over the 8x256 packed counters in (row, thread) order, the
//      low-half grand total carried into the high halves: counter (d, t) becomes the number of
//      tile keys that sort before thread t's first key of digit d;
//   4. keys go to LDS at their tile rank and come back in rank order, so lane l of a store
//      instruction writes run position s = m*256 + l: consecutive lanes, consecutive addresses.
// ---------------------------------------------------------------------------------------------
template <typename K, bool HAS_V>
__global__ __launch_bounds__(kBlock) void scatter_kernel(
    const K* __restrict__ kin, K* __restrict__ kout, const uint32_t* __restrict__ vin,
    uint32_t* __restrict__ vout, uint32_t n, int bit, uint32_t mask,
    const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ chunk_sums,
    uint32_t num_tiles) {
    constexpr int KPT = KeyTraits<K>::kpt;
    constexpr uint32_t TILE = kBlock * KPT;
    constexpr int PER_VEC = 16 / sizeof(K);
    constexpr int NVEC = kThreadDwords / 4;

    __shared__ __attribute__((aligned(16))) uint32_t stage[kBlock * kRowStride];
    __shared__ __attribute__((aligned(16))) uint32_t ctr[kCtrRows * kBlock];
    __shared__ uint32_t delta[kRadices];
    __shared__ uint32_t wave_tot[kBlock / 64];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tile = tile_of_block(blockIdx.x, num_tiles);
    const uint32_t base = tile * TILE;
    const uint32_t valid = min(TILE, n - base);

    // Global output base of every digit for this tile; issued first, consumed after the scan.
    uint32_t gbase = 0;
    if (tid < static_cast<uint32_t>(kRadices)) {
        const uint32_t idx = tid * num_tiles + tile;
        gbase = offsets[idx] + chunk_sums[idx / kScanChunk];
    }

    K key[KPT];
    uint32_t val[HAS_V ? KPT : 1];
    if (valid == TILE) {
        const uint4* src = reinterpret_cast<const uint4*>(kin + base);
        uint4 v[NVEC];
#pragma unroll
        for (int m = 0; m < NVEC; ++m) v[m] = src[m * kBlock + tid];
        if (HAS_V) {
            const uint4* vsrc = reinterpret_cast<const uint4*>(vin + base);
#pragma unroll
            for (int j = 0; j < KPT / 4; ++j) {
                const uint4 q = vsrc[tid * (KPT / 4) + j];
                val[4 * j + 0] = q.x; val[4 * j + 1] = q.y; val[4 * j + 2] = q.z; val[4 * j + 3] = q.w;
            }
        }
#pragma unroll
        for (int m = 0; m < NVEC; ++m) {
            const uint32_t row = m * (kBlock / 4) + (tid >> 2);
            *reinterpret_cast<uint4*>(&stage[row * kRowStride + 4u * (tid & 3u)]) = v[m];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < NVEC; ++m) {
            const uint4 q = *reinterpret_cast<const uint4*>(&stage[tid * kRowStride + 4 * m]);
            unpack4<K>(q, &key[m * PER_VEC]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = tid * KPT + i;
            key[i] = idx < valid ? kin[base + idx] : static_cast<K>(~static_cast<K>(0));
            if (HAS_V) val[i] = idx < valid ? vin[base + idx] : 0u;
        }
    }

    // 2. thread-private packed counters.
#pragma unroll
    for (int r = 0; r < kCtrRows; ++r) ctr[r * kBlock + tid] = 0u;
    uint32_t rank[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t d = digit_of<K>(key[i], bit, mask);
        const uint32_t sh = (d >> 3) << 4;
        const uint32_t old = __hip_atomic_fetch_add(&ctr[(d & 7u) * kBlock + tid], 1u << sh,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        rank[i] = (old >> sh) & 0xffffu;
    }
    __syncthreads();

    // 3. block-wide exclusive scan, thread t owning words 8t..8t+7 of the row-major counter array.
    {
        const uint4 a = *reinterpret_cast<const uint4*>(&ctr[tid * 8u]);
        const uint4 b = *reinterpret_cast<const uint4*>(&ctr[tid * 8u + 4u]);
        const uint32_t c[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t total = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) total += c[i];
        const uint32_t incl = wave_inclusive_scan(total, lane);
        if (lane == 63u) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            const uint32_t t = wave_tot[w];
            if (static_cast<uint32_t>(w) < wave) before += t;
            all += t;
        }
        // digits 8..15 (high halves) sort after all of digits 0..7 (low halves)
        uint32_t run = incl - total + before + ((all & 0xffffu) << 16);
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { o[i] = run; run += c[i]; }
        *reinterpret_cast<uint4*>(&ctr[tid * 8u]) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(&ctr[tid * 8u + 4u]) = make_uint4(o[4], o[5], o[6], o[7]);
    }
    __syncthreads();

    // 4. tile ranks; digit d's run starts at the counter of (d, thread 0).
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t d = digit_of<K>(key[i], bit, mask);
        const uint32_t sh = (d >> 3) << 4;
        rank[i] += (ctr[(d & 7u) * kBlock + tid] >> sh) & 0xffffu;
    }
    if (tid < static_cast<uint32_t>(kRadices)) {
        const uint32_t start = (ctr[(tid & 7u) * kBlock] >> ((tid >> 3) << 4)) & 0xffffu;
        delta[tid] = gbase - start;
    }
    K* skeys = reinterpret_cast<K*>(stage);
#pragma unroll
    for (int i = 0; i < KPT; ++i) skeys[rank[i]] = key[i];
    __syncthreads();

    uint32_t dst[HAS_V ? KPT : 1];
#pragma unroll
    for (int m = 0; m < KPT; ++m) {
        const uint32_t s = m * kBlock + tid;
        const K k = skeys[s];
        const uint32_t where = delta[digit_of<K>(k, bit, mask)] + s;
        if (HAS_V) dst[m] = where;
        if (s < valid) kout[where] = k;
    }
    if (HAS_V) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) stage[rank[i]] = val[i];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < KPT; ++m) {
            const uint32_t s = m * kBlock + tid;
            if (s < valid) vout[dst[m]] = stage[s];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// harness kernels
// ---------------------------------------------------------------------------------------------
__global__ void generate_u32_kernel(uint32_t* __restrict__ keys, uint64_t n, uint32_t seed) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
        keys[i] = harness_hash(static_cast<uint32_t>(i) ^ seed);
}

__global__ void generate_u64_kernel(uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                    uint64_t n, uint32_t seed) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t lo = static_cast<uint32_t>(i);
        keys[i] = (static_cast<uint64_t>(harness_hash(lo ^ seed)) << 32) | harness_hash(~lo ^ seed);
        if (vals) vals[i] = lo;
    }
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_xor_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v ^= __shfl_xor(v, off, 64);
    return v;
}

template <typename K>
__global__ void check_sorted_kernel(const K* __restrict__ keys, const uint32_t* __restrict__ vals,
                                    uint64_t n, unsigned long long* __restrict__ bad) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    uint64_t mine = 0;
    for (uint64_t i = 1 + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const K a = keys[i - 1], b = keys[i];
        if (a > b) ++mine;
        else if (vals && a == b && vals[i - 1] > vals[i]) ++mine;
    }
    mine = wave_sum_u64(mine);
    if ((threadIdx.x & 63u) == 0u && mine) atomicAdd(bad, static_cast<unsigned long long>(mine));
}

template <typename K> __device__ __forceinline__ uint64_t key_mix(K k);
template <> __device__ __forceinline__ uint64_t key_mix<uint32_t>(uint32_t k) { return harness_hash(k); }
template <> __device__ __forceinline__ uint64_t key_mix<uint64_t>(uint64_t k) {
    const uint32_t lo = harness_hash(static_cast<uint32_t>(k));
    const uint32_t hi = harness_hash(static_cast<uint32_t>(k >> 32) ^ lo);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

template <typename K>
__global__ void checksum_kernel(const K* __restrict__ keys, uint64_t n,
                                unsigned long long* __restrict__ out) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    uint64_t sum = 0, x = 0, mix = 0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const K k = keys[i];
        sum += k; x ^= k; mix += key_mix<K>(k);
    }
    sum = wave_sum_u64(sum); x = wave_xor_u64(x); mix = wave_sum_u64(mix);
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&out[0], static_cast<unsigned long long>(sum));
        atomicXor(&out[1], static_cast<unsigned long long>(x));
        atomicAdd(&out[2], static_cast<unsigned long long>(mix));
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
thread_local char g_error[256] = "";

int fail(hipError_t e, const char* what) {
    snprintf(g_error, sizeof(g_error), "%s: %s", what, hipGetErrorString(e));
    return static_cast<int>(e);
}
int bad_arg(const char* what) {
    snprintf(g_error, sizeof(g_error), "invalid argument: %s", what);
    return static_cast<int>(hipErrorInvalidValue);
}

#define RADX_TRY(expr)                                   \
    do {                                                 \
        const hipError_t e_ = (expr);                    \
        if (e_ != hipSuccess) return fail(e_, #expr);    \
    } while (0)

constexpr uint64_t kMaxElements = 0xffffffffull;  // element indices are 32-bit, as in the reference

uint32_t tile_keys(int key_bytes) { return static_cast<uint32_t>(kTileDwords * 4 / key_bytes); }
uint32_t tiles_for(uint64_t n, int key_bytes) {
    const uint64_t t = tile_keys(key_bytes);
    return static_cast<uint32_t>((n + t - 1) / t);
}
size_t align256(size_t b) { return (b + 255u) & ~static_cast<size_t>(255u); }

struct ScratchLayout {
    size_t counts, offsets, chunk_sums, digit_starts, total;
};
ScratchLayout layout_for(uint64_t max_n, int key_bytes) {
    const size_t tiles = tiles_for(max_n ? max_n : 1, key_bytes);
    const size_t words = tiles * kRadices;
    const size_t chunks = (words + kScanChunk - 1) / kScanChunk;
    ScratchLayout l;
    l.counts = 0;
    l.offsets = l.counts + align256(words * 4);
    l.chunk_sums = l.offsets + align256(words * 4);
    l.digit_starts = l.chunk_sums + align256(chunks * 4);
    l.total = l.digit_starts + align256(32 * 4);
    return l;
}

struct Timed {
    hipEvent_t start, stop;
    int kind;
};

class DeviceGuard {
public:
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        if (prev_ != device) { status_ = hipSetDevice(device); changed_ = status_ == hipSuccess; }
    }
    ~DeviceGuard() { if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_); }
    hipError_t status() const { return status_; }
private:
    int prev_ = -1;
    bool changed_ = false;
    hipError_t status_ = hipSuccess;
};

}  // namespace

struct radx_hip_sorter {
    int device = 0;
    uint64_t max_n = 0;
    int key_bytes = 4;
    int value_bytes = 0;
    char* scratch = nullptr;
    ScratchLayout layout{};
    uint32_t* counts = nullptr;
    uint32_t* offsets = nullptr;
    uint32_t* chunk_sums = nullptr;
    uint32_t* digit_starts = nullptr;
    bool profile = false;
    std::vector<Timed> pending;
    std::vector<Timed> pool;
    double ms[3] = {0, 0, 0};
    uint64_t launches[3] = {0, 0, 0};
};

namespace {

// Brackets one launch with events when profiling is on.
class LaunchTimer {
public:
    LaunchTimer(radx_hip_sorter* s, int kind, hipStream_t stream) : s_(s), stream_(stream) {
        if (!s_->profile) return;
        if (!s_->pool.empty()) { t_ = s_->pool.back(); s_->pool.pop_back(); }
        else if (hipEventCreate(&t_.start) != hipSuccess || hipEventCreate(&t_.stop) != hipSuccess) return;
        t_.kind = kind;
        active_ = hipEventRecord(t_.start, stream_) == hipSuccess;
    }
    ~LaunchTimer() {
        if (!active_) return;
        (void)hipEventRecord(t_.stop, stream_);
        s_->pending.push_back(t_);
    }
private:
    radx_hip_sorter* s_;
    hipStream_t stream_;
    Timed t_{};
    bool active_ = false;
};

template <typename K, bool HAS_V>
int run_pass_typed(radx_hip_sorter* s, const void* kin, void* kout, const void* vin, void* vout,
                   uint32_t n, int bit, int num_bits, hipStream_t stream) {
    const uint32_t tiles = tiles_for(n, sizeof(K));
    const uint32_t words = tiles * kRadices;
    const uint32_t chunks = (words + kScanChunk - 1) / kScanChunk;
    const uint32_t mask = (1u << num_bits) - 1u;
    {
        LaunchTimer t(s, 0, stream);
        hipLaunchKernelGGL(hist_kernel<K>, dim3(tiles), dim3(kBlock), 0, stream,
                           static_cast<const K*>(kin), n, bit, mask, s->counts, tiles);
    }
    {
        LaunchTimer t(s, 1, stream);
        hipLaunchKernelGGL(scan_chunks_kernel, dim3(chunks), dim3(kBlock), 0, stream, s->counts,
                           s->offsets, s->chunk_sums, words);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, s->chunk_sums, chunks,
                           s->offsets, tiles, n, s->digit_starts);
    }
    {
        LaunchTimer t(s, 2, stream);
        hipLaunchKernelGGL((scatter_kernel<K, HAS_V>), dim3(tiles), dim3(kBlock), 0, stream,
                           static_cast<const K*>(kin), static_cast<K*>(kout),
                           static_cast<const uint32_t*>(vin), static_cast<uint32_t*>(vout), n, bit,
                           mask, s->offsets, s->chunk_sums, tiles);
    }
    RADX_TRY(hipGetLastError());
    return 0;
}

int run_pass(radx_hip_sorter* s, const void* kin, void* kout, const void* vin, void* vout, uint32_t n,
             int bit, int num_bits, hipStream_t stream) {
    const bool has_v = s->value_bytes != 0;
    if (s->key_bytes == 4)
        return has_v ? run_pass_typed<uint32_t, true>(s, kin, kout, vin, vout, n, bit, num_bits, stream)
                     : run_pass_typed<uint32_t, false>(s, kin, kout, vin, vout, n, bit, num_bits, stream);
    return has_v ? run_pass_typed<uint64_t, true>(s, kin, kout, vin, vout, n, bit, num_bits, stream)
                 : run_pass_typed<uint64_t, false>(s, kin, kout, vin, vout, n, bit, num_bits, stream);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_buffers(const radx_hip_sorter* s, const void* k0, const void* k1, const void* v0,
                  const void* v1, uint64_t n) {
    if (!s) return bad_arg("sorter is NULL");
    if (n > s->max_n) return bad_arg("n exceeds the sorter's max_n");
    if (n == 0) return 0;
    if (!k0 || !k1) return bad_arg("keys / keys_swap is NULL");
    if (k0 == k1) return bad_arg("keys and keys_swap are the same buffer");
    if (!aligned16(k0) || !aligned16(k1)) return bad_arg("key buffers must be 16-byte aligned");
    if (s->value_bytes) {
        if (!v0 || !v1) return bad_arg("vals / vals_swap is NULL for a key-value sorter");
        if (v0 == v1) return bad_arg("vals and vals_swap are the same buffer");
        if (!aligned16(v0) || !aligned16(v1)) return bad_arg("value buffers must be 16-byte aligned");
    } else if (v0 || v1) {
        return bad_arg("vals given to a keys-only sorter");
    }
    return 0;
}

}  // namespace

extern "C" {

int radx_hip_version(void) { return 100; }

const char* radx_hip_error_string(void) { return g_error; }

size_t radx_hip_scratch_bytes(uint64_t max_n, int key_bytes) {
    if (key_bytes != 4 && key_bytes != 8) return 0;
    if (max_n > kMaxElements) return 0;
    return layout_for(max_n, key_bytes).total;
}

int radx_hip_create(int device, uint64_t max_n, int key_bytes, int value_bytes, radx_hip_sorter** out) {
    if (!out) return bad_arg("out is NULL");
    *out = nullptr;
    if (key_bytes != 4 && key_bytes != 8) return bad_arg("key_bytes must be 4 or 8");
    if (value_bytes != 0 && value_bytes != 4) return bad_arg("value_bytes must be 0 or 4");
    if (max_n > kMaxElements) return bad_arg("max_n must be < 2^32");
    int count = 0;
    RADX_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return bad_arg("no such HIP device");
    DeviceGuard guard(device);
    RADX_TRY(guard.status());
    radx_hip_sorter* s = new (std::nothrow) radx_hip_sorter;
    if (!s) return fail(hipErrorOutOfMemory, "new radx_hip_sorter");
    s->device = device;
    s->max_n = max_n;
    s->key_bytes = key_bytes;
    s->value_bytes = value_bytes;
    s->layout = layout_for(max_n, key_bytes);
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->scratch), s->layout.total);
    if (e != hipSuccess) { delete s; return fail(e, "hipMalloc(scratch)"); }
    s->counts = reinterpret_cast<uint32_t*>(s->scratch + s->layout.counts);
    s->offsets = reinterpret_cast<uint32_t*>(s->scratch + s->layout.offsets);
    s->chunk_sums = reinterpret_cast<uint32_t*>(s->scratch + s->layout.chunk_sums);
    s->digit_starts = reinterpret_cast<uint32_t*>(s->scratch + s->layout.digit_starts);
    *out = s;
    return 0;
}

int radx_hip_destroy(radx_hip_sorter* s) {
    if (!s) return 0;
    DeviceGuard guard(s->device);
    for (auto& t : s->pending) { (void)hipEventSynchronize(t.stop); s->pool.push_back(t); }
    for (auto& t : s->pool) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    const hipError_t e = s->scratch ? hipFree(s->scratch) : hipSuccess;
    delete s;
    if (e != hipSuccess) return fail(e, "hipFree(scratch)");
    return 0;
}

int radx_hip_pass(radx_hip_sorter* s, const void* keys_in, void* keys_out, const void* vals_in,
                  void* vals_out, uint64_t n, int bit, int num_bits, void* stream) {
    if (const int rc = check_buffers(s, keys_in, keys_out, vals_in, vals_out, n)) return rc;
    if (num_bits < 0 || num_bits > kDigitBits) return bad_arg("num_bits must be 0..4");
    if (bit < 0 || bit + num_bits > s->key_bytes * 8) return bad_arg("digit outside the key");
    if (n == 0) return 0;
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    return run_pass(s, keys_in, keys_out, vals_in, vals_out, static_cast<uint32_t>(n), bit, num_bits,
                    static_cast<hipStream_t>(stream));
}

int radx_hip_sort_bits(radx_hip_sorter* s, void* keys, void* keys_swap, void* vals, void* vals_swap,
                       uint64_t n, int begin_bit, int end_bit, void* stream) {
    if (const int rc = check_buffers(s, keys, keys_swap, vals, vals_swap, n)) return rc;
    if (begin_bit < 0 || end_bit > s->key_bytes * 8 || begin_bit > end_bit)
        return bad_arg("need 0 <= begin_bit <= end_bit <= key bits");
    if (n == 0) return 0;
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    int passes = (end_bit - begin_bit + kDigitBits - 1) / kDigitBits;
    passes += passes & 1;  // even, so the result ends in `keys` (scattering.comp:40,143)
    void* kbuf[2] = {keys, keys_swap};
    void* vbuf[2] = {vals, vals_swap};
    for (int p = 0; p < passes; ++p) {
        const int bit = begin_bit + p * kDigitBits;
        int width = end_bit - bit;
        if (width > kDigitBits) width = kDigitBits;
        if (width < 0) width = 0;
        const int src = p & 1, dstb = 1 - src;
        const int rc = run_pass(s, kbuf[src], kbuf[dstb], vbuf[src], vbuf[dstb], static_cast<uint32_t>(n),
                                width ? bit : 0, width, static_cast<hipStream_t>(stream));
        if (rc) return rc;
    }
    return 0;
}

int radx_hip_sort(radx_hip_sorter* s, void* keys, void* keys_swap, void* vals, void* vals_swap,
                  uint64_t n, void* stream) {
    if (!s) return bad_arg("sorter is NULL");
    return radx_hip_sort_bits(s, keys, keys_swap, vals, vals_swap, n, 0, s->key_bytes * 8, stream);
}

int radx_hip_geometry(const radx_hip_sorter* s, uint64_t n, uint32_t* tile, uint32_t* num_tiles) {
    if (!s) return bad_arg("sorter is NULL");
    if (n > kMaxElements) return bad_arg("n must be < 2^32");
    if (tile) *tile = tile_keys(s->key_bytes);
    if (num_tiles) *num_tiles = tiles_for(n, s->key_bytes);
    return 0;
}

int radx_hip_copy_tables(radx_hip_sorter* s, uint64_t n, uint32_t* counts, uint32_t* offsets, void* stream) {
    if (!s) return bad_arg("sorter is NULL");
    if (n == 0 || n > s->max_n) return bad_arg("n out of range");
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t tiles = tiles_for(n, s->key_bytes);
    const size_t words = static_cast<size_t>(tiles) * kRadices;
    const size_t chunks = (words + kScanChunk - 1) / kScanChunk;
    RADX_TRY(hipStreamSynchronize(st));
    if (counts) RADX_TRY(hipMemcpy(counts, s->counts, words * 4, hipMemcpyDeviceToHost));
    if (offsets) {
        std::vector<uint32_t> sums(chunks);
        RADX_TRY(hipMemcpy(offsets, s->offsets, words * 4, hipMemcpyDeviceToHost));
        RADX_TRY(hipMemcpy(sums.data(), s->chunk_sums, chunks * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < words; ++i) offsets[i] += sums[i / kScanChunk];
    }
    return 0;
}

int radx_hip_digit_starts(radx_hip_sorter* s, uint32_t out[17], void* stream) {
    if (!s || !out) return bad_arg("NULL argument");
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    RADX_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    RADX_TRY(hipMemcpy(out, s->digit_starts, (kRadices + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

int radx_hip_profile(radx_hip_sorter* s, int enable) {
    if (!s) return bad_arg("sorter is NULL");
    s->profile = enable != 0;
    return 0;
}

int radx_hip_kernel_times(radx_hip_sorter* s, double ms[3], uint64_t launches[3]) {
    if (!s) return bad_arg("sorter is NULL");
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    for (auto& t : s->pending) {
        RADX_TRY(hipEventSynchronize(t.stop));
        float f = 0.f;
        RADX_TRY(hipEventElapsedTime(&f, t.start, t.stop));
        s->ms[t.kind] += f;
        s->launches[t.kind] += 1;
        s->pool.push_back(t);
    }
    s->pending.clear();
    for (int i = 0; i < 3; ++i) {
        if (ms) ms[i] = s->ms[i];
        if (launches) launches[i] = s->launches[i];
        s->ms[i] = 0;
        s->launches[i] = 0;
    }
    return 0;
}

int radx_hip_generate_u32(void* keys, uint64_t n, uint32_t seed, void* stream) {
    if (n == 0) return 0;
    if (!keys) return bad_arg("keys is NULL");
    hipLaunchKernelGGL(generate_u32_kernel, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint32_t*>(keys), n, seed);
    RADX_TRY(hipGetLastError());
    return 0;
}

int radx_hip_generate_u64(void* keys, void* vals, uint64_t n, uint32_t seed, void* stream) {
    if (n == 0) return 0;
    if (!keys) return bad_arg("keys is NULL");
    hipLaunchKernelGGL(generate_u64_kernel, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint64_t*>(keys), static_cast<uint32_t*>(vals), n, seed);
    RADX_TRY(hipGetLastError());
    return 0;
}

int radx_hip_check_sorted(const void* keys, const void* vals, uint64_t n, int key_bytes,
                          uint64_t* violations, void* stream) {
    if (!violations) return bad_arg("violations is NULL");
    if (key_bytes != 4 && key_bytes != 8) return bad_arg("key_bytes must be 4 or 8");
    *violations = 0;
    if (n < 2) return 0;
    if (!keys) return bad_arg("keys is NULL");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long* d = nullptr;
    RADX_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), st);
    if (e == hipSuccess) {
        if (key_bytes == 4)
            hipLaunchKernelGGL(check_sorted_kernel<uint32_t>, dim3(4096), dim3(256), 0, st,
                               static_cast<const uint32_t*>(keys), static_cast<const uint32_t*>(vals), n, d);
        else
            hipLaunchKernelGGL(check_sorted_kernel<uint64_t>, dim3(4096), dim3(256), 0, st,
                               static_cast<const uint64_t*>(keys), static_cast<const uint32_t*>(vals), n, d);
        e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(e, "check_sorted");
    *violations = h;
    return 0;
}

int radx_hip_checksum(const void* keys, uint64_t n, int key_bytes, uint64_t out[3], void* stream) {
    if (!out) return bad_arg("out is NULL");
    if (key_bytes != 4 && key_bytes != 8) return bad_arg("key_bytes must be 4 or 8");
    out[0] = out[1] = out[2] = 0;
    if (n == 0) return 0;
    if (!keys) return bad_arg("keys is NULL");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long* d = nullptr;
    RADX_TRY(hipMalloc(reinterpret_cast<void**>(&d), 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), st);
    if (e == hipSuccess) {
        if (key_bytes == 4)
            hipLaunchKernelGGL(checksum_kernel<uint32_t>, dim3(4096), dim3(256), 0, st,
                               static_cast<const uint32_t*>(keys), n, d);
        else
            hipLaunchKernelGGL(checksum_kernel<uint64_t>, dim3(4096), dim3(256), 0, st,
                               static_cast<const uint64_t*>(keys), n, d);
        e = hipGetLastError();
    }
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(e, "checksum");
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    return 0;
}

}  // extern "C"
