// radx_hip.hip — gfx950 (MI355X, CDNA4) LSD radix sort: kernels and the C ABI of include/radx_hip.h.
//
// One pass = histogram -> scan -> scatter on a 4-bit digit, the three phases the reference
// dispatches per pass (include/radx/radx_implement.inl:427-444; shaders-sdk/radix/counting.comp,
// partition.comp, scattering.comp).  How the work is cut is this engine's own:
//
//   * Chunking is the reference's get_blocks_info (shaders-sdk/radix/includes.glsl:162-173) with a
//     grid that grows with n instead of the fixed 108 workgroups (include/radx/radx_internal.hpp:143):
//     one workgroup per tile of 16 KiB of keys (4096 uint32 / 2048 uint64), so every workgroup's
//     chunk is exactly one tile.
//   * The tables keep the reference's layout, counts[workgroup][radix] and
//     partitions[workgroup][radix] (counting.comp:45, partition.comp:25-26): one 64-byte line per
//     workgroup.
//   * Inside a tile every key gets its stable rank from per-thread digit counters in LDS
//     (packed 2 x 16 bit, laid out [digit & 7][thread]) and one block-wide scan over them; the
//     tile is written to LDS in sorted order and each digit's run leaves the CU as contiguous
//     dword stores.  Address order is preserved per digit, which is the reference's rank
//     (scattering.comp:119-123,141-143).
//   * Only the first pass of a sort runs the histogram kernel.  While scatter pass p stores its
//     keys it also counts, per destination tile, the NEXT digit of every key it stores, and adds
//     those counts to the table pass p+1 will scan: the table pass p+1 needs is exactly "digit
//     counts per tile of pass p's output", and pass p is the one kernel that knows where every key
//     lands.  A digit run of one source tile is contiguous and no longer than a tile, so it
//     touches at most 2 destination tiles: at most 2*16*16 counters per workgroup, kept in LDS
//     and flushed once with global atomic adds.
//
//   * Inputs of at most 16384 tiles run without scan kernels: a few extra workgroups at the front
//     of every scatter launch scan that pass's table and publish the results, and every tile
//     workgroup picks up its 16 output bases from them (see scatter_kernel, SCANLESS).  One launch
//     per pass instead of three.
//     Such launches are short, so what surrounds the streaming counts: the workgroups that share
//     a CU at the start of a launch begin a step apart instead of moving in lockstep, and a pass
//     whose output would sit dirty in the L2s until the kernel ends stores it nontemporally.
//   * Inputs of at most one tile are sorted by one launch of one workgroup: the keys stay in LDS
//     through all passes (sort_tile_kernel).  Same digits, same stable rank, one launch instead
//     of nine.
//
// All cross-wave LDS hand-offs are separated by __syncthreads().  Workgroups never wait on each
// other without a bound inside a launch: the inter-workgroup traffic is no-return atomic adds
// whose sums are read by a later kernel on the same stream, and, in the scan-free passes, words
// published once by workgroups that depend on nobody, which a reader that does not find them
// after a bounded number of looks replaces by its own computation.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "radx_hip.h"

namespace {

constexpr int kBlock = 256;             // threads per workgroup (4 waves of 64)
constexpr int kRadices = RADX_HIP_RADICES;
constexpr int kDigitBits = RADX_HIP_BITS_PER_PASS;
constexpr int kThreadDwords = 16;       // every thread owns 64 B of keys
constexpr int kTileDwords = kBlock * kThreadDwords;  // 16 KiB of keys per tile
constexpr int kCtrRows = kRadices / 2;  // packed counters: digit d lives in row d&7, half d>>3
constexpr int kScanBlocks = 256;        // table rows (workgroups) scanned by one scan workgroup
// Inputs of at most kScanlessMaxTiles tiles run without scan kernels: a few extra workgroups of
// the scatter launch publish the column sums of every group of 64 table rows (128 rows above
// kSmallGroupMaxTiles tiles: never more than kMaxGroups groups), and every
// scatter workgroup derives its own 16 output bases from those sums and the rows of its own group.
constexpr uint32_t kScanlessMaxTiles = 16384;
constexpr uint32_t kSmallGroupMaxTiles = 8192;
constexpr uint32_t kMaxGroups = 128;
__host__ __device__ constexpr int log_group_tiles(uint32_t num_tiles) { return num_tiles <= kSmallGroupMaxTiles ? 6 : 7; }
__host__ __device__ constexpr uint32_t group_count(uint32_t num_tiles) {
    return (num_tiles + (1u << log_group_tiles(num_tiles)) - 1u) >> log_group_tiles(num_tiles);
}
constexpr int kGroupIters = kMaxGroups / 16;                      // group rows per 16-lane team: 8
constexpr int kMaxRowIters = (kScanlessMaxTiles / kMaxGroups) / 16;  // own-group rows per 16-lane team: 4 or 8
constexpr uint32_t kGroupValid = 0x80000000u;  // a published word (at most 2^26: 16384 tiles of 4096 keys) carries this bit
constexpr int kMaxPolls = 64;                  // looks at an unpublished word before a workgroup sums the table itself
// One buffer of group words: kMaxGroups rows of 16 column sums, then kMaxGroups rows of 16 output bases.
constexpr uint32_t kGroupBaseWords = kMaxGroups * 16;  // where the bases start
// Up to this many groups every tile workgroup adds up the group sums itself (2 table rows per
// team of 16 lanes) instead of waiting for the scanning workgroup's bases: one step less in the
// chain, which counts when the whole launch is one round of workgroups.
// A scan-free pass that writes between these many bytes stores them nontemporally (see scatter_kernel, NT):
// up to half of what the eight L2s hold together; 24 MiB is already 8 % slower that way, and
// below 3 MiB there is too little to write back for it to matter (1-2 % the other way).
constexpr uint64_t kNtStoreMinBytes = 3ull << 20, kNtStoreMaxBytes = 16ull << 20;
constexpr uint32_t kSelfSumTiles = 64;  // scan-free tables of at most this many rows: no publishing, see scatter_kernel
constexpr int kStaggerSleep = 32;  // x 64 cycles between the workgroups that share a CU in a launch's first round
constexpr uint32_t kOneLevelGroups = 32;
constexpr int kOneLevelIters = kOneLevelGroups / 16;

template <typename K> struct KeyTraits;
template <> struct KeyTraits<uint32_t> { static constexpr int kpt = 16; static constexpr int log_tile = 12; };
template <> struct KeyTraits<uint64_t> { static constexpr int kpt = 8;  static constexpr int log_tile = 11; };

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------

// Workgroups are dealt round-robin over the 8 XCDs, so workgroups b and b+8 share an L2.  This
// bijection hands every XCD blocks of B consecutive tiles, the 8 XCDs' blocks interleaved: tiles
// that run next to each other in time on one XCD are neighbours in memory, so the two halves of
// an output line that straddles two tiles' runs meet in the same L2 (dealing single tiles round
// robin, B = 1, costs the scatter 35 %), while all 8 XCDs still work inside one window of 8 * B
// tiles, which keeps the chip's write streams (one per digit and XCD) closer together than giving
// each XCD one eighth of the array does (B = 512: -1.1 % on 4-byte keys; 8-byte keys and
// payloads want B = 2048).  The tiles past the last whole group of 8 * B are split into one
// contiguous range per XCD.
// Speed only; any placement is correct.
template <uint32_t B>
__device__ __forceinline__ uint32_t tile_of_workgroup(uint32_t bid, uint32_t nwg) {
    // inputs of fewer than 8 such windows (which largely fit the memory-side cache) measured
    // better with one contiguous range per XCD throughout
    const uint32_t full = nwg >= 64u * B ? nwg / (8u * B) * (8u * B) : 0u;
    const uint32_t x = bid & 7u;
    if (bid < full) {
        const uint32_t k = bid >> 3;
        return (k / B) * (8u * B) + x * B + (k % B);
    }
    // the rest (all of a small input): one contiguous range per XCD
    const uint32_t rest = nwg - full, k = (bid - full) >> 3;
    const uint32_t q = rest >> 3, r = rest & 7u;
    return full + (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + k;
}
template <typename K, bool HAS_V> constexpr uint32_t xcd_block() { return sizeof(K) == 4 && !HAS_V ? 512u : 2048u; }

// The digit of (mask + 1) values at `bit`.  For 32-bit keys this is one v_bfe_u32 (the width is
// the mask's population count, a scalar the compiler hoists); 64-bit keys shift and mask.
template <typename K>
__device__ __forceinline__ uint32_t digit_of(K key, int bit, uint32_t mask) {
    return static_cast<uint32_t>(key >> bit) & mask;
}
template <>
__device__ __forceinline__ uint32_t digit_of<uint32_t>(uint32_t key, int bit, uint32_t mask) {
    return __builtin_amdgcn_ubfe(key, static_cast<uint32_t>(bit), static_cast<uint32_t>(__builtin_popcount(mask)));
}

// Key orders other than "unsigned ascending" (the reference declares them as
// RadicePropStruct{Descending, IsSigned}, shaders-sdk/radix/includes.glsl:137, and never reads
// them).  The kernels only ever sort unsigned ascending; a sort in another order maps every key
// to the unsigned integer with the same rank as it is loaded by the first pass and maps it back
// as the last pass stores it, so no extra traffic.  mode bits: 1 two's-complement signed,
// 2 IEEE float (total order by sign and magnitude: -NaN < -inf < ... < -0 < +0 < ... < +inf < +NaN),
// 4 descending.  Bits 4-5 (the multi-GPU split only): rotate the image's top 4-bit digit left by
// that many bits, so that a digit made of (destination rank, sub-bucket) sorts as
// (sub-bucket, destination rank).
constexpr int kOrderSigned = 1, kOrderFloat = 2, kOrderDescending = 4, kOrderRotShift = 4;

template <typename K>
__host__ __device__ __forceinline__ K rotate_top_digit(K k, int r) {
    constexpr int SH = sizeof(K) * 8 - 4;
    const uint32_t top = static_cast<uint32_t>(k >> SH);
    const uint32_t rot = ((top << r) | (top >> (4 - r))) & 15u;
    return static_cast<K>((k & ((static_cast<K>(1) << SH) - 1)) | (static_cast<K>(rot) << SH));
}

template <typename K>
__host__ __device__ __forceinline__ K key_to_ordered(K k, int mode) {
    constexpr int TOP = sizeof(K) * 8 - 1;
    constexpr K SIGN = static_cast<K>(1) << TOP;
    if (mode & kOrderFloat) k ^= ((k >> TOP) ? static_cast<K>(~static_cast<K>(0)) : SIGN);
    else if (mode & kOrderSigned) k ^= SIGN;
    if (mode & kOrderDescending) k = static_cast<K>(~k);
    if (const int r = (mode >> kOrderRotShift) & 3) k = rotate_top_digit<K>(k, r);
    return k;
}
template <typename K>
__host__ __device__ __forceinline__ K ordered_to_key(K k, int mode) {
    constexpr int TOP = sizeof(K) * 8 - 1;
    constexpr K SIGN = static_cast<K>(1) << TOP;
    if (const int r = (mode >> kOrderRotShift) & 3) k = rotate_top_digit<K>(k, 4 - r);
    if (mode & kOrderDescending) k = static_cast<K>(~k);
    if (mode & kOrderFloat) k ^= ((k >> TOP) ? SIGN : static_cast<K>(~static_cast<K>(0)));
    else if (mode & kOrderSigned) k ^= SIGN;
    return k;
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

// Keys and payloads are read exactly once per pass: nontemporal loads keep them from displacing
// the output lines that neighbouring tiles are still completing in L2 (histogram kernel +8 %;
// nontemporal STORES lose that merging and cost the scatter 10 %, so stores stay plain).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 stream_load4(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <typename K> __device__ __forceinline__ void unpack4(const uint4& v, K* out);
template <> __device__ __forceinline__ void unpack4<uint32_t>(const uint4& v, uint32_t* out) {
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <> __device__ __forceinline__ void unpack4<uint64_t>(const uint4& v, uint64_t* out) {
    out[0] = (static_cast<uint64_t>(v.y) << 32) | v.x;
    out[1] = (static_cast<uint64_t>(v.w) << 32) | v.z;
}

// Inclusive scan over the 64 lanes of a wave with DPP adds: shifts by 1, 2, 4, 8 inside each row
// of 16 lanes, then the row totals are broadcast into the following rows (row_bcast:15 into rows
// 1 and 3, row_bcast:31 into rows 2 and 3).  Lanes a step does not reach add 0.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false));  // row_shr:1
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false));  // row_shr:2
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false));  // row_shr:4
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false));  // row_shr:8
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false));  // row_bcast:15
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false));  // row_bcast:31
    return v;
}

// Where element `rank` of a rank-ordered LDS array of E-byte elements lives: the index is XORed
// with its own 128-byte-row number (mod the elements per row).  When every key of a tile has the
// same digit, thread t's keys have ranks KPT*t .. KPT*t+KPT-1, so a wave's lanes write with a
// stride of 64 bytes — 2 banks for 64 lanes; the XOR spreads such a stride over all 32 banks,
// and a wave reading 64 consecutive ranks back still touches every bank twice.
template <int E>
__device__ __forceinline__ uint32_t rank_slot(uint32_t rank) {
    constexpr uint32_t PER_ROW = 128u / E;  // 32 (4-byte) or 16 (8-byte) elements
    return rank ^ ((rank / PER_ROW) & (PER_ROW - 1u));
}

// The tile image in LDS used to hand every thread its own 64 consecutive bytes: logical dword
// (row, c), row = owning thread, c = 0..15, lives at row*16 + ((c/4) ^ ((row >> 2) & 3))*4 + c%4.
// The XOR on the 16-byte chunk index makes both the coalesced ds_write_b128 (lanes 4r..4r+3 fill
// row r) and the per-thread ds_read_b128 (lane t reads row t) bank-conflict free without padding.
__device__ __forceinline__ uint32_t tile_image_dword(uint32_t row, uint32_t c) {
    return row * kThreadDwords + ((((c >> 2) ^ (row >> 2)) & 3u) << 2) + (c & 3u);
}

// ---------------------------------------------------------------------------------------------
// histogram: counts[g][d] = number of keys of tile g whose digit is d
// (shaders-sdk/radix/counting.comp:49-116).  Runs for the first pass of a sort only.  LDS bins
// are replicated 32x so that lane l only ever touches bank l&31: no bank conflict and no
// same-address serialisation however skewed the digits are.
// ---------------------------------------------------------------------------------------------
template <typename K>
__global__ __launch_bounds__(kBlock) void hist_kernel(const K* __restrict__ keys, uint32_t n, int bit,
                                                       uint32_t mask, uint32_t* __restrict__ counts,
                                                       uint32_t num_tiles, int order_in,
                                                       uint32_t* __restrict__ zero_counts,
                                                       uint32_t* __restrict__ zero_groups_a,
                                                       uint32_t* __restrict__ zero_groups_b,
                                                       uint32_t* __restrict__ zero_prefix_a,
                                                       uint32_t* __restrict__ zero_prefix_b) {
    constexpr int KPT = KeyTraits<K>::kpt;
    constexpr uint32_t TILE = kBlock * KPT;
    constexpr int PER_VEC = 16 / sizeof(K);
    __shared__ uint32_t bins[kRadices * 32];

    const uint32_t tid = threadIdx.x;
    // a pure read has no output lines to merge: small blocks (64 tiles = 1 MiB per XCD) read 4 % faster
    const uint32_t g = tile_of_workgroup<64u>(blockIdx.x, num_tiles);
    bins[tid] = 0u;
    bins[tid + kBlock] = 0u;
    __syncthreads();

    const uint32_t col = tid & 31u;
    const uint32_t base = g * TILE;
    const uint32_t valid = min(TILE, n - base);
    if (valid == TILE) {
        const uint4* src = reinterpret_cast<const uint4*>(keys + base);
        uint4 v[kThreadDwords / 4];
#pragma unroll
        for (int m = 0; m < kThreadDwords / 4; ++m) v[m] = stream_load4(&src[m * kBlock + tid]);
#pragma unroll
        for (int m = 0; m < kThreadDwords / 4; ++m) {
            K k[PER_VEC];
            unpack4<K>(v[m], k);
            if (order_in) {
#pragma unroll
                for (int j = 0; j < PER_VEC; ++j) k[j] = key_to_ordered<K>(k[j], order_in);
            }
#pragma unroll
            for (int j = 0; j < PER_VEC; ++j) {
                const uint32_t d = digit_of<K>(k[j], bit, mask);
                __hip_atomic_fetch_add(&bins[d * 32u + col], 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    } else {
        for (uint32_t i = tid; i < valid; i += kBlock) {
            const uint32_t d = digit_of<K>(key_to_ordered<K>(keys[base + i], order_in), bit, mask);
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
        counts[static_cast<size_t>(g) * kRadices + d] = lo;
        counts[static_cast<size_t>(g) * kRadices + d + 8u] = hi;
    }
    // scan-free mode: the table the first scatter adds into, and the published words of the first two passes
    if (zero_counts && tid < static_cast<uint32_t>(kRadices)) {
        zero_counts[static_cast<size_t>(g) * kRadices + tid] = 0u;
        zero_prefix_a[static_cast<size_t>(g) * kRadices + tid] = 0u;
        zero_prefix_b[static_cast<size_t>(g) * kRadices + tid] = 0u;
    }
    if (zero_groups_a) {
        const uint32_t w = blockIdx.x * kBlock + tid;
        if (w < group_count(num_tiles) * kRadices) {
            zero_groups_a[w] = 0u; zero_groups_a[kGroupBaseWords + w] = 0u;
            zero_groups_b[w] = 0u; zero_groups_b[kGroupBaseWords + w] = 0u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// scan: partitions[g][r] = sum_{r' < r} total[r'] + sum_{g' < g} counts[g'][r], the exclusive scan
// in (radix, workgroup) order that shaders-sdk/radix/partition.comp:29-68 produces in two phases.
// Kernel 1: every scan workgroup takes 256 consecutive table rows (one per thread) and scans each
// of the 16 radix columns across them, leaving the column totals (and zeroes the same rows of the
// table the following scatter pass accumulates the next pass's counts in).  Kernel 2 (one workgroup) scans
// those totals in (radix, scan-workgroup) order.  The scatter kernel adds the two parts.  When
// the whole table fits one scan workgroup (at most 256 tiles) kernel 1 scans its 16 totals itself
// and kernel 2 is not launched.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void scan_rows_kernel(const uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ offsets,
                                                            uint32_t* __restrict__ group_sums,
                                                            uint32_t num_tiles, uint32_t* __restrict__ clear,
                                                            uint32_t n, uint32_t* __restrict__ digit_starts) {
    __shared__ uint32_t wave_tot[kBlock / 64][kRadices];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t row = blockIdx.x * kScanBlocks + tid;
    uint32_t v[kRadices];
    if (row < num_tiles) {
        const uint4* src = reinterpret_cast<const uint4*>(counts + static_cast<size_t>(row) * kRadices);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 a = src[q];
            v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
        }
    } else {
#pragma unroll
        for (int d = 0; d < kRadices; ++d) v[d] = 0u;
    }
    uint32_t incl[kRadices];
#pragma unroll
    for (int d = 0; d < kRadices; ++d) incl[d] = wave_inclusive_scan(v[d]);
    if (lane == 63u) {
#pragma unroll
        for (int d = 0; d < kRadices; ++d) wave_tot[wave][d] = incl[d];
    }
    __syncthreads();
    uint32_t o[kRadices];
#pragma unroll
    for (int d = 0; d < kRadices; ++d) {
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w)
            if (static_cast<uint32_t>(w) < wave) before += wave_tot[w][d];
        o[d] = incl[d] - v[d] + before;
    }
    if (row < num_tiles) {
        uint4* dst = reinterpret_cast<uint4*>(offsets + static_cast<size_t>(row) * kRadices);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        if (clear) {  // the table the following scatter pass adds the next pass's counts into
            uint4* z = reinterpret_cast<uint4*>(clear + static_cast<size_t>(row) * kRadices);
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if (wave == 0u) {
        uint32_t all = 0;
        if (tid < static_cast<uint32_t>(kRadices)) {
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) all += wave_tot[w][tid];
        }
        if (gridDim.x == 1u) {  // the only scan workgroup: its totals are the digit totals
            const uint32_t start = wave_inclusive_scan(all) - all;
            if (tid < static_cast<uint32_t>(kRadices)) { group_sums[tid] = start; digit_starts[tid] = start; }
            if (tid == 0u) digit_starts[kRadices] = n;
        } else if (tid < static_cast<uint32_t>(kRadices)) {
            group_sums[static_cast<size_t>(tid) * gridDim.x + blockIdx.x] = all;
        }
    }
}

// group_sums[d][j] (radix-major, so the (radix, scan-workgroup) order is the linear order of the
// array) -> exclusive scan in place.  One workgroup; per round every thread owns 16 consecutive
// words, loaded up front (independent loads), summed, block-scanned and rewritten.  One round
// covers 16 Ki words = 2^30 uint32 keys.
__global__ __launch_bounds__(1024) void scan_groups_kernel(uint32_t* __restrict__ group_sums,
                                                            uint32_t num_groups, uint32_t n,
                                                            uint32_t* __restrict__ digit_starts) {
    constexpr int PER = 16;
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t len = num_groups * kRadices;
    uint32_t carry = 0;
    for (uint32_t round = 0; round < len; round += 1024u * PER) {
        const uint32_t first = round + tid * PER;
        uint32_t v[PER];
        if (first + PER <= len) {
#pragma unroll
            for (int q = 0; q < PER / 4; ++q) {
                const uint4 a = *reinterpret_cast<const uint4*>(group_sums + first + 4 * q);
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PER; ++i) v[i] = first + i < len ? group_sums[first + i] : 0u;
        }
        uint32_t total = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) total += v[i];
        const uint32_t incl = wave_inclusive_scan(total);
        __syncthreads();  // wave_tot of the previous round has been read by everyone
        if (lane == 63u) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t run = carry + incl - total, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t t = wave_tot[w];
            if (static_cast<uint32_t>(w) < wave) run += t;
            all += t;
        }
        carry += all;
        // every word is read and rewritten by the same thread only, so in place is safe
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t L = first + i;
            if (L < len) {
                group_sums[L] = run;
                if (L % num_groups == 0u) digit_starts[L / num_groups] = run;  // start of every digit's output run
                run += v[i];
            }
        }
    }
    if (tid == 0u) digit_starts[kRadices] = n;
}

// ---------------------------------------------------------------------------------------------
// scatter: stable partition of one tile by digit, written to partitions[g][d] + rank
// (shaders-sdk/radix/scattering.comp:57-149; out index = localPartitions[d] + earlier keys of
// digit d in address order, :141-143).
//
//   1. coalesced 16-B loads, passed through an XOR-swizzled LDS image so thread t holds the
//      tile's keys [t*KPT, (t+1)*KPT) — consecutive in address order;
//   2. per key one returning LDS add on the thread's own packed counter: the old value is the
//      key's rank among the thread's earlier keys of that digit;
//   3. one block-wide exclusive scan over the 8x256 packed counters in (row, thread) order, the
//      low-half grand total carried into the high halves: counter (d, t) becomes the number of
//      tile keys that sort before thread t's first key of digit d;
//   4. keys go to LDS at their tile rank and come back in rank order, so lane l of a store
//      instruction writes run position s = m*256 + l: consecutive lanes, consecutive addresses;
//   5. FUSE: every stored key adds 1 to an LDS counter for (its digit, its next-pass digit).
//      Counters for keys landing in the first destination tile of their digit's run are plain
//      words, replicated over 8 columns (word = slot * 8 + (lane & 7)): a wave's 64 keys are
//      consecutive in sorted order, so they share their digit and spread over 16 next digits x 8
//      columns — 2-way bank conflicts on average, and 8 lanes per word when all keys are equal.
//      One shift-or and one shift-add make the address and the add is a constant 1: 4 VALU
//      operations per key fewer than byte counters in 32 conflict-free columns, and faster on
//      every distribution.  The counter words reuse the rank counters' LDS, which is dead by
//      then.  Keys landing in the run's second destination
//      tile (a run crosses at most one tile boundary) are rare and use a plain 16x16 table.
//      After the last store the sums are added to next_counts[destination tile][next digit].
// ---------------------------------------------------------------------------------------------
// What only the scan-free mode touches.
struct ScanFreeTables {
    uint32_t* group_words;     // this pass's group sums and bases: cleared before the launch, published by its first workgroups
    uint32_t* prefix_words;    // this pass's [tile][16] sums of the tile's own group in front of it, published likewise
    uint32_t* zero_counts;     // the table the pass after the next accumulates into: cleared row by row here
    uint32_t* zero_group_words;  // and the group words of the pass after the next
    uint32_t* zero_prefix_words; // and its prefix words
    uint32_t* digit_starts;    // the 17 run starts
    int publish;               // tests only: 0 nothing is published, 2 only the bases are (1: everything)
};

// SCANLESS: no scan kernel runs for this pass; the scan is done by a few extra workgroups at the
// front of the launch (their number rounded up to 8, which keeps the XCD of every tile workgroup).
//   * Extra workgroup b < ceil(num_tiles / G) scans the 16 columns of table rows [G b, G b + G)
//     of `counts` (G = 64 rows, 128 above 8192 tiles): it publishes, for every one of its tiles,
//     the 16 sums of the group's rows in front of that tile, and the 16 sums over the whole group.
//   * The next extra workgroup waits for those sums, scans them in (digit, group) order and
//     publishes, per group, the 16 output bases of the group's first tile (and the 17 run starts).
//   * The workgroup of tile g reads the 16 bases of its group and its own 16 in-group sums (two
//     64-byte rows, by 16 lanes), adds them and stores its 16 bases to offsets[g] (the reference's
//     partitions[g][r], partition.comp:26) for readers of the tables.
// A published word carries bit 31 (the values stay below 2^26) and is written with one agent-scope
// store and read with agent-scope loads: value and "is there" travel in one word, so no fences.
// Tile workgroups of the first round look for their words only after their keys are loaded and
// counted in LDS (with 4-byte keys after the block-wide scan as well: the bases are not needed
// before the keys leave); the publishing workgroups have the lowest numbers, so they run first.
// A table of at most kSelfSumTiles rows is not published: every tile workgroup sums it itself.
// Nobody waits without a
// bound: whoever still misses a word after kMaxPolls looks sums the table itself — the scanning
// workgroup its group sums, a tile workgroup its 16 bases — so no workgroup's completion depends
// on another one's progress.  Tables rotate through three buffers: this pass's are read, the
// next pass's are added to, the ones after that are cleared.
// Keys alone run fastest with 5 workgroups per CU, which 4-byte keys reach at 96 registers: the
// scan-free variant would take 98 without the bound (no spills with it).
template <typename K, bool HAS_V> constexpr int scatter_min_waves() { return sizeof(K) == 4 && !HAS_V ? 5 : 1; }

// NT: keys and payloads leave with nontemporal stores, which go to memory as they are issued.
// With ordinary stores a launch whose output fits the L2s ends with all of it still dirty there,
// and the write-back at the end of the kernel runs with nothing to overlap it (2^21 uint32 keys:
// 96 -> 82 us per sort; 2^21 uint32 keys + payloads: 134 -> 96 us).  Larger outputs are evicted
// as the kernel runs and want the L2 to merge the partial lines two tiles' runs share (2^25 keys:
// 23 % slower with NT), so the host picks NT by the bytes a pass writes.
template <typename K, bool HAS_V, bool FUSE, bool SCANLESS, bool NT = false>
__global__ __launch_bounds__(kBlock, (scatter_min_waves<K, HAS_V>())) void scatter_kernel(
    const K* __restrict__ kin, K* __restrict__ kout, const uint32_t* __restrict__ vin,
    uint32_t* __restrict__ vout, uint32_t n, int bit, uint32_t mask,
    const uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets, const uint32_t* __restrict__ group_sums,
    uint32_t num_tiles, int next_bit, uint32_t next_mask, uint32_t* __restrict__ next_counts,
    int order_in, int order_out, ScanFreeTables sf) {
    constexpr int KPT = KeyTraits<K>::kpt;
    constexpr int LOG_TILE = KeyTraits<K>::log_tile;
    constexpr uint32_t TILE = kBlock * KPT;
    constexpr int PER_VEC = 16 / sizeof(K);
    constexpr int NVEC = kThreadDwords / 4;
    constexpr int KEY_DWORDS = sizeof(K) / 4;
#if defined(RADX_EXP_ALIGN_ALL)
    constexpr bool ALIGNED = true;
#elif defined(RADX_EXP_ALIGN_NONE)
    constexpr bool ALIGNED = false;
#else
    constexpr bool ALIGNED = sizeof(K) == 8;
#endif
    // ALIGNED: the rank-ordered LDS array has up to ALIGN_E - 1 empty slots in front of every
    // digit's run, chosen so that slot number and global output index agree modulo ALIGN_E
    // (128 bytes of keys).  A wave of the store loop covers 64 consecutive slots starting at a
    // multiple of 64, so every store instruction writes whole 128-byte lines except at the two
    // ends of a run.
    constexpr uint32_t ALIGN_E = ALIGNED ? 128u / sizeof(K) : 1u;
    constexpr uint32_t SLOTS = ALIGNED ? (TILE + kRadices * (ALIGN_E - 1u) + kBlock - 1u) / kBlock * kBlock : TILE;
    constexpr int NIT = SLOTS / kBlock;  // store-loop iterations: 16 -> 18 (4-byte keys), 8 -> 9 (8-byte keys)

    __shared__ __attribute__((aligned(16))) uint32_t stage[SLOTS * KEY_DWORDS];
    __shared__ __attribute__((aligned(16))) uint32_t ctr[kCtrRows * kBlock];  // FUSE: reused as the next pass's counters [256][8]
    __shared__ uint32_t next_second[FUSE ? kRadices * kRadices : 1];
    // per digit: .x = global base - first slot of the digit, .y = last index of its first destination
    // tile, .z = first slot, .w = number of keys (ALIGNED only)
    __shared__ __attribute__((aligned(16))) uint4 where_s[kRadices];
    __shared__ uint32_t padpack_s[ALIGNED ? kCtrRows : 1];  // empty slots in front of digits r (low half) and r + 8 (high half)
    __shared__ uint32_t wave_tot[kBlock / 64];
    __shared__ uint32_t part_s[SCANLESS ? (kBlock / 64) * 2 * kRadices : 1];  // per wave: 16 digit totals, 16 sums in front of the tile
    __shared__ uint32_t self_sum_s;  // SCANLESS: some group sum never showed up, sum the table here

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // SCANLESS: table rows are read by teams of 16 lanes (team = tid / 16), lane tcol of a team reading column tcol
    const uint32_t tcol = tid & 15u, team = tid >> 4;
    const int log_group = log_group_tiles(num_tiles);
    const uint32_t num_grps = group_count(num_tiles);
    const uint32_t publishers = SCANLESS ? (num_grps + 1u + 7u) & ~7u : 0u;
    if (SCANLESS && blockIdx.x < publishers) {
        const uint32_t b = blockIdx.x;
        if (b > num_grps || sf.publish == 0 || num_tiles <= kSelfSumTiles) return;
        if (b == num_grps && num_grps <= kOneLevelGroups) return;  // nobody reads the bases
        if (b < num_grps) {
            // One group of table rows.  Team t owns rows 4t .. 4t+3 (8t .. 8t+7 in groups of 128) of
            // the group, lane c column c: exclusive scan down every column, then the column sums.
            const uint32_t team_rows = 1u << (log_group - 4);
            const uint32_t first_row = (b << log_group) + team * team_rows;
            uint32_t c[kMaxRowIters], mine = 0;
#pragma unroll
            for (int i = 0; i < kMaxRowIters; ++i) {
                const uint32_t r = first_row + i;
                c[i] = static_cast<uint32_t>(i) < team_rows && r < num_tiles ? counts[static_cast<size_t>(r) * kRadices + tcol] : 0u;
                mine += c[i];
            }
            uint32_t incl = mine;  // over the teams of this wave
            uint32_t up = __shfl_up(incl, 16, 64);
            if (lane >= 16u) incl += up;
            up = __shfl_up(incl, 32, 64);
            if (lane >= 32u) incl += up;
            if (lane >= 48u) part_s[wave * kRadices + tcol] = incl;
            __syncthreads();
            uint32_t run = incl - mine, all = 0;
#pragma unroll
            for (int v = 0; v < kBlock / 64; ++v) {
                const uint32_t t = part_s[v * kRadices + tcol];
                if (static_cast<uint32_t>(v) < wave) run += t;
                all += t;
            }
#pragma unroll
            for (int i = 0; i < kMaxRowIters; ++i) {
                const uint32_t r = first_row + i;
                if (static_cast<uint32_t>(i) < team_rows && r < num_tiles)
                    __hip_atomic_store(&sf.prefix_words[static_cast<size_t>(r) * kRadices + tcol], run | kGroupValid,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                run += c[i];
            }
            if (sf.publish == 1 && tid < static_cast<uint32_t>(kRadices))
                __hip_atomic_store(&sf.group_words[b * kRadices + tid], all | kGroupValid, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // The scan over the groups.  Team t owns groups 8t .. 8t+7, lane c of a team column c.
        uint32_t w[kGroupIters];
        bool have = false;
        for (int looks = 1; looks <= kMaxPolls; ++looks) {
            uint32_t there = kGroupValid;
#pragma unroll
            for (int i = 0; i < kGroupIters; ++i) {
                const uint32_t r = team * kGroupIters + i;
                w[i] = r < num_grps ? __hip_atomic_load(&sf.group_words[r * kRadices + tcol], __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT)
                                    : kGroupValid;
                there &= w[i];
            }
            have = __syncthreads_and(static_cast<int>(there >> 31)) != 0;
            if (have) break;
            __builtin_amdgcn_s_sleep(16);
        }
        if (!have) {  // never in practice: the group sums straight from the table
#pragma unroll  // (w stays in registers only if every index into it is a constant)
            for (int i = 0; i < kGroupIters; ++i) {
                const uint32_t r = team * kGroupIters + i;
                uint32_t sum = 0;
#pragma unroll 1
                for (uint32_t t = r << log_group; t < min((r + 1u) << log_group, num_tiles); ++t)
                    sum += counts[static_cast<size_t>(t) * kRadices + tcol];
                w[i] = sum;
            }
        }
        uint32_t mine = 0;  // this team's groups, this column
#pragma unroll
        for (int i = 0; i < kGroupIters; ++i) { w[i] &= ~kGroupValid; mine += w[i]; }
        uint32_t incl = mine;  // over the teams of this wave
        uint32_t up = __shfl_up(incl, 16, 64);
        if (lane >= 16u) incl += up;
        up = __shfl_up(incl, 32, 64);
        if (lane >= 32u) incl += up;
        if (lane >= 48u) part_s[wave * kRadices + tcol] = incl;
        __syncthreads();
        uint32_t before = incl - mine, all = 0;
#pragma unroll
        for (int v = 0; v < kBlock / 64; ++v) {
            const uint32_t t = part_s[v * kRadices + tcol];
            if (static_cast<uint32_t>(v) < wave) before += t;
            all += t;
        }
        if (wave == 0u) {  // every digit's run start: exclusive scan of the 16 digit totals (lanes 0..15)
            const uint32_t tot = lane < static_cast<uint32_t>(kRadices) ? all : 0u;
            const uint32_t start = wave_inclusive_scan(tot) - tot;
            if (lane < static_cast<uint32_t>(kRadices)) {
                part_s[(kBlock / 64) * kRadices + lane] = start;
                sf.digit_starts[lane] = start;
            }
            if (lane == 0u) sf.digit_starts[kRadices] = n;
        }
        __syncthreads();
        uint32_t run = part_s[(kBlock / 64) * kRadices + tcol] + before;
#pragma unroll
        for (int i = 0; i < kGroupIters; ++i) {
            const uint32_t r = team * kGroupIters + i;
            if (r < num_grps)
                __hip_atomic_store(&sf.group_words[kGroupBaseWords + r * kRadices + tcol], run | kGroupValid,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            run += w[i];
        }
        return;
    }
    // The workgroups that start a launch together on one CU would all load, then all rank, then
    // all store: the memory system gets their loads in one burst and idles while they rank.  The
    // dispatcher deals the first 256 workgroups one to each CU, the next 256 likewise, and so on,
    // so workgroup b is the (b / 256)-th of its CU: it starts that many steps of kStaggerSleep x 64
    // cycles late, which spreads the first round's phases for the whole launch (2^21..2^25 uint32
    // keys: 2-5 % per sort; sleeping once the loads are issued gains nothing).  Speed only.
    if (SCANLESS) {
        constexpr uint32_t slots = HAS_V ? 4u : 5u;  // workgroups per CU in one round
        const uint32_t slot_in_cu = (blockIdx.x - publishers) >> 8;
        if (slot_in_cu < slots)
            for (uint32_t k = 0; k < slot_in_cu; ++k) __builtin_amdgcn_s_sleep(kStaggerSleep);
    }
    const uint32_t g = tile_of_workgroup<xcd_block<K, HAS_V>()>(blockIdx.x - publishers, num_tiles);
    const uint32_t base = g * TILE;
    const uint32_t valid = min(TILE, n - base);

    // Global output base of every digit for this tile; issued first, consumed after the scan.
    uint32_t run_base = 0, run_len = 0;
    const uint32_t my_grp = g >> log_group;
    uint32_t base_word = kGroupValid, prefix_word = kGroupValid, grp_word[SCANLESS ? kOneLevelIters : 1];
    const bool one_level = num_grps <= kOneLevelGroups;
    // Lanes 0..15 of wave 0: the tile's sums over its own group's rows in front of it.  One level:
    // every team's group sums as well (rows past the end re-read the last one and are masked when
    // summed).  Two levels: lanes 0..15 of wave 0 also read the output bases of the tile's group.
    // A table of at most kSelfSumTiles rows is summed by every tile workgroup itself, with ordinary
    // loads that hit the L2: shorter than the trip of a word through a publishing workgroup, its
    // store and a load past the L2 (2^14..2^18 uint32 keys: 4-8 % per sort; from 128 rows on the
    // published words win).
    const bool self_mode = SCANLESS && num_tiles <= kSelfSumTiles;
    auto look_at_group_words = [&]() {
        if (self_mode) return;
        if (tid < static_cast<uint32_t>(kRadices))
            prefix_word = __hip_atomic_load(&sf.prefix_words[static_cast<size_t>(g) * kRadices + tid], __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
        if (one_level) {
#pragma unroll
            for (int i = 0; i < kOneLevelIters; ++i) {
                grp_word[i] = kGroupValid;
                if (16u * i < num_grps)  // uniform
                    grp_word[i] = __hip_atomic_load(&sf.group_words[min(team + 16u * i, num_grps - 1u) * kRadices + tcol],
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (tid < static_cast<uint32_t>(kRadices)) {
            base_word = __hip_atomic_load(&sf.group_words[kGroupBaseWords + my_grp * kRadices + tid], __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // The first round of workgroups starts together with the publishers: they look after loading
    // and counting their keys.  Later workgroups find the words published and look right away.
    const bool late_look = blockIdx.x - publishers < 1024u;
    if (SCANLESS) {
        if (tid == 0u) self_sum_s = self_mode ? 1u : 0u;
        if (!late_look) look_at_group_words();
        if (ALIGNED && tid < static_cast<uint32_t>(kRadices)) run_len = counts[static_cast<size_t>(g) * kRadices + tid];
    } else if (tid < static_cast<uint32_t>(kRadices)) {
        run_base = offsets[static_cast<size_t>(g) * kRadices + tid] +
                   group_sums[static_cast<size_t>(tid) * ((num_tiles + kScanBlocks - 1) / kScanBlocks) + g / kScanBlocks];
        if (ALIGNED) run_len = counts[static_cast<size_t>(g) * kRadices + tid];
    }
    // wave 0, ALIGNED only: the slot layout of the padded rank-ordered array, from the tile's own
    // histogram row and the output bases (lanes 0..15 hold digit d's key count and global base)
    auto place_runs = [&]() {
        // The digit's first slot is its tile rank start plus a pad that makes slot = output index
        // (mod ALIGN_E); pads must not decrease from digit to digit, hence the 16-step recurrence.
        const uint32_t start = wave_inclusive_scan(run_len) - run_len;
        const uint32_t want = (run_base - start) & (ALIGN_E - 1u);
        uint32_t pad = 0, prev = 0;
#pragma unroll
        for (int d = 0; d < kRadices; ++d) {
            const uint32_t mine = prev + ((want - prev) & (ALIGN_E - 1u));
            if (lane == static_cast<uint32_t>(d)) pad = mine;
            prev = __builtin_amdgcn_readlane(mine, d);
        }
        const uint32_t hi = __shfl_down(pad, 8, 64);
        if (tid < static_cast<uint32_t>(kRadices)) {
            const uint32_t first = start + pad;
            where_s[tid] = make_uint4(run_base - first, run_base | (TILE - 1u), first, run_len);
            if (tid < static_cast<uint32_t>(kCtrRows)) padpack_s[tid] = pad | (hi << 16);
        }
    };

    K key[KPT];
    uint32_t val[HAS_V ? KPT : 1];
    if (valid == TILE) {
        const uint4* src = reinterpret_cast<const uint4*>(kin + base);
        uint4 v[NVEC];
#pragma unroll
        for (int m = 0; m < NVEC; ++m) v[m] = stream_load4(&src[m * kBlock + tid]);
        if (HAS_V) {
            // Payloads are loaded the way the keys are — consecutive lanes, consecutive 16 bytes —
            // and handed to their owning threads through an LDS image of KPT dwords per row.
            constexpr uint32_t C = KPT / 4;  // 16-byte chunks per row
            uint32_t* vimg = stage;
            const uint4* vsrc = reinterpret_cast<const uint4*>(vin + base);
            uint4 q[C];
#pragma unroll
            for (uint32_t j = 0; j < C; ++j) q[j] = stream_load4(&vsrc[j * kBlock + tid]);
#pragma unroll
            for (uint32_t j = 0; j < C; ++j) {
                const uint32_t idx = j * kBlock + tid, row = idx / C, chunk = idx % C;
                *reinterpret_cast<uint4*>(&vimg[row * KPT + (((chunk ^ (row >> 2)) & (C - 1u)) << 2)]) = q[j];
            }
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < C; ++j) {
                const uint4 r = *reinterpret_cast<const uint4*>(&vimg[tid * KPT + (((j ^ (tid >> 2)) & (C - 1u)) << 2)]);
                val[4 * j + 0] = r.x; val[4 * j + 1] = r.y; val[4 * j + 2] = r.z; val[4 * j + 3] = r.w;
            }
            __syncthreads();  // the key image goes into the same words
        }
#pragma unroll
        for (int m = 0; m < NVEC; ++m) {
            const uint32_t row = m * (kBlock / 4) + (tid >> 2);
            *reinterpret_cast<uint4*>(&stage[tile_image_dword(row, 4u * (tid & 3u))]) = v[m];
        }
    } else {
        // ragged last tile of the input: element-wise loads into the same LDS image, all-ones
        // keys past the end (they take the highest ranks of the last digit and are not stored)
#pragma unroll 1
        for (uint32_t e = tid; e < TILE; e += kBlock) {
            const K k = e < valid ? kin[base + e] : ordered_to_key<K>(static_cast<K>(~static_cast<K>(0)), order_in);
            const uint32_t at = tile_image_dword(e / KPT, (e % KPT) * KEY_DWORDS);
            stage[at] = static_cast<uint32_t>(k);
            if (KEY_DWORDS == 2) stage[at + 1] = static_cast<uint32_t>(static_cast<uint64_t>(k) >> 32);
        }
        if (HAS_V) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) val[i] = vin[base + min(tid * KPT + i, valid - 1u)];
        }
    }
    if (!SCANLESS && ALIGNED && wave == 0u) place_runs();
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NVEC; ++m) {
        const uint4 q = *reinterpret_cast<const uint4*>(&stage[tile_image_dword(tid, 4u * m)]);
        unpack4<K>(q, &key[m * PER_VEC]);
    }
    if (order_in) {  // first pass of a sort in a non-default key order
#pragma unroll
        for (int i = 0; i < KPT; ++i) key[i] = key_to_ordered<K>(key[i], order_in);
    }

    // 2. thread-private packed counters.
#pragma unroll
    for (int r = 0; r < kCtrRows; ++r) ctr[r * kBlock + tid] = 0u;
    if (SCANLESS && late_look) look_at_group_words();  // in flight during the counting below
    // Ranks (below the tile size) stay in registers until the keys — and, with a payload, the
    // values much later — go to LDS.  With a payload they are packed two to a register to ease
    // the pressure over that long range (uint64 + payload: 83.0 -> 81.1 ms per 2^30 sort);
    // without one the packing arithmetic costs more than it saves (uint32: 13.34 vs 13.28 ms).
    constexpr bool PACK = HAS_V;
    uint32_t rank[PACK ? KPT / 2 : KPT];
    auto rank_of = [&](int i) { return PACK ? (rank[i / 2] >> ((i & 1) << 4)) & 0xffffu : rank[i]; };
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t d = digit_of<K>(key[i], bit, mask);
        const uint32_t sh = (d >> 3) << 4;
        const uint32_t old = __hip_atomic_fetch_add(&ctr[(d & 7u) * kBlock + tid], 1u << sh,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t r = (old >> sh) & 0xffffu;
        if (PACK) rank[i / 2] = (i & 1) ? rank[i / 2] | (r << 16) : r;
        else rank[i] = r;
    }
    // SCANLESS: the tile's 16 output bases.  await_words: wait (with a bound) for the published
    // words and reduce the group sums to one partial pair per wave; take_bases, one barrier
    // later: wave 0 combines them (or, a word having failed to show up, every wave sums the table
    // first).  The bases are not needed before the keys leave for global memory, so where the
    // slot layout does not depend on them (!ALIGNED) both steps come after the block-wide scan:
    // the workgroups that start a launch look for their words about when they are published,
    // and every step the look moves back saves some of them a second look.
    constexpr bool LATE_BASES = SCANLESS && !ALIGNED;
    auto await_words = [&]() {
        if (self_mode) return;
        if (one_level || wave == 0u) {
            for (int looks = 1;; ++looks) {
                uint32_t there = base_word & prefix_word;
#pragma unroll
                for (int i = 0; i < kOneLevelIters; ++i) there &= one_level ? grp_word[i] : kGroupValid;
                if (__all(static_cast<int>(there >> 31))) break;
                if (looks == kMaxPolls) { if (lane == 0u) self_sum_s = 1u; break; }  // every writer writes 1
                __builtin_amdgcn_s_sleep(8);
                look_at_group_words();
            }
        }
        // One level: the group sums, every team's partial column sums, then the 4 teams of a wave.
        // Two levels: the scanning workgroup has done all of that and there is nothing to add up;
        // skipping the empty reduction is worth 2 % to 4-byte keys (2^24..2^26) and costs 8-byte
        // keys 2-5 %, measured — hence by key size.
        if (one_level || !LATE_BASES) {
            uint32_t tot = 0, bef = 0;
#pragma unroll
            for (int i = 0; i < kOneLevelIters; ++i) {
                const uint32_t r = team + 16u * i, w = one_level ? grp_word[i] & ~kGroupValid : 0u;
                tot += r < num_grps ? w : 0u;
                bef += r < my_grp ? w : 0u;
            }
            tot += __shfl_xor(tot, 16, 64); bef += __shfl_xor(bef, 16, 64);
            tot += __shfl_xor(tot, 32, 64); bef += __shfl_xor(bef, 32, 64);
            if (lane < static_cast<uint32_t>(kRadices)) {
                part_s[wave * 2 * kRadices + lane] = tot;
                part_s[wave * 2 * kRadices + kRadices + lane] = bef;
            }
        }
    };
    auto take_bases = [&]() {
        if (self_sum_s) {  // uniform.  Small tables, or (never in practice) a word that failed to show up: the column sums straight from the table.
            uint32_t tot = 0, bef = 0;
            // 8 rows per team in flight at a time: the loads hit the L2 and are independent
            for (uint32_t r0 = team; r0 < num_tiles; r0 += 16u * 8u) {
                uint32_t w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t r = r0 + 16u * j;
                    w[j] = r < num_tiles ? counts[static_cast<size_t>(r) * kRadices + tcol] : 0u;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    tot += w[j];
                    bef += r0 + 16u * j < g ? w[j] : 0u;
                }
            }
            tot += __shfl_xor(tot, 16, 64); bef += __shfl_xor(bef, 16, 64);
            tot += __shfl_xor(tot, 32, 64); bef += __shfl_xor(bef, 32, 64);
            if (lane < static_cast<uint32_t>(kRadices)) {
                part_s[wave * 2 * kRadices + lane] = tot;
                part_s[wave * 2 * kRadices + kRadices + lane] = bef;
            }
            __syncthreads();
        }
        if (wave == 0u) {
            if (one_level || self_sum_s || !LATE_BASES) {  // uniform
                uint32_t tot = 0, bef = 0;
                if (lane < static_cast<uint32_t>(kRadices)) {
#pragma unroll
                    for (int w = 0; w < kBlock / 64; ++w) {
                        tot += part_s[w * 2 * kRadices + lane];
                        bef += part_s[w * 2 * kRadices + kRadices + lane];
                    }
                }
                // the digit starts from the totals (lanes past 15 add 0), then what lies in front of the tile
                const uint32_t scanned = wave_inclusive_scan(tot) - tot;
                const uint32_t start = (one_level || self_sum_s) ? scanned : base_word & ~kGroupValid;
                run_base = start + bef + (self_sum_s ? 0u : prefix_word & ~kGroupValid);
            } else {
                run_base = (base_word & ~kGroupValid) + (prefix_word & ~kGroupValid);  // the group's bases + the tile's place in its group
            }
            if (lane < static_cast<uint32_t>(kRadices)) {
                offsets[static_cast<size_t>(g) * kRadices + lane] = run_base;
                sf.zero_counts[static_cast<size_t>(g) * kRadices + lane] = 0u;
                sf.zero_prefix_words[static_cast<size_t>(g) * kRadices + lane] = 0u;
                if (g < num_grps) {
                    sf.zero_group_words[g * kRadices + lane] = 0u;
                    sf.zero_group_words[kGroupBaseWords + g * kRadices + lane] = 0u;
                }
                if (g == 0u && (one_level || sf.publish == 0)) sf.digit_starts[lane] = run_base;  // nobody scanned
            }
            if (g == 0u && lane == 0u && (one_level || sf.publish == 0)) sf.digit_starts[kRadices] = n;
            if (ALIGNED) place_runs();  // read after the next barrier
        }
    };
    if (SCANLESS && !LATE_BASES) await_words();
    __syncthreads();
    if (SCANLESS && !LATE_BASES) take_bases();

    // 3. block-wide exclusive scan, thread t owning words 8t..8t+7 of the row-major counter array.
    {
        const uint4 a = *reinterpret_cast<const uint4*>(&ctr[tid * 8u]);
        const uint4 b = *reinterpret_cast<const uint4*>(&ctr[tid * 8u + 4u]);
        const uint32_t c[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t total = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) total += c[i];
        const uint32_t incl = wave_inclusive_scan(total);
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
        if (ALIGNED) run += padpack_s[tid >> 5];  // thread t owns words 8t..8t+7: counter row t / 32
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { o[i] = run; run += c[i]; }
        *reinterpret_cast<uint4*>(&ctr[tid * 8u]) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(&ctr[tid * 8u + 4u]) = make_uint4(o[4], o[5], o[6], o[7]);
    }
    if (LATE_BASES) await_words();
    __syncthreads();
    if (LATE_BASES) take_bases();

    // 4. tile ranks; digit d's tile run starts at the counter of (d, thread 0).
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t d = digit_of<K>(key[i], bit, mask);
        const uint32_t sh = (d >> 3) << 4;
        const uint32_t before = (ctr[(d & 7u) * kBlock + tid] >> sh) & 0xffffu;
        if (PACK) rank[i / 2] += before << ((i & 1) << 4);  // stays below the tile size: no carry into the other half
        else rank[i] += before;
    }
    if (!ALIGNED && tid < static_cast<uint32_t>(kRadices)) {
        const uint32_t start = (ctr[(tid & 7u) * kBlock] >> ((tid >> 3) << 4)) & 0xffffu;
        where_s[tid] = make_uint4(run_base - start, run_base | (TILE - 1u), 0u, 0u);
    }
    if (FUSE) {
        // The rank counters are dead once read: every thread clears the column only it read
        // (column 0 is also read by lanes 1..15 of thread 0's own wave, earlier in program order),
        // so the words can serve as the zeroed next-pass counters after the next barrier.
#pragma unroll
        for (int r = 0; r < kCtrRows; ++r) ctr[r * kBlock + tid] = 0u;
        next_second[tid] = 0u;
    }
    K* skeys = reinterpret_cast<K*>(stage);
#pragma unroll
    for (int i = 0; i < KPT; ++i) skeys[rank_slot<sizeof(K)>(rank_of(i))] = key[i];
    __syncthreads();

    // Full tiles in the default key order (all but at most one workgroup per launch, and all but
    // the first / last pass of a sort in another order) take a path without any branch in it.
    // Keys are handled kGroup at a time: the group's LDS reads are independent and issue back to
    // back instead of one read -> wait -> read -> wait -> store chain per key.  Measured at 2^30:
    // uint32 keys 13.45 -> 13.05 ms per sort with groups of 4..16; uint64 + payload 84.5 -> 83.2 ms
    // with 8; uint64 keys alone are fastest one at a time.
    constexpr int kGroup = ALIGNED ? (sizeof(K) == 4 ? 6 : (HAS_V ? 3 : 1)) : (sizeof(K) == 8 && !HAS_V) ? 1 : 8;
    static_assert(NIT % kGroup == 0, "the store loop handles whole groups");
    // FUSE: the next pass's counters of keys that land in their run's first destination tile,
    // ctr reused as uint32[256 slots = (digit, next digit)][kNextCols]: a wave's 64 keys are
    // consecutive in sorted order, so they share their digit and spread over 16 next digits x 8 columns
    constexpr uint32_t kNextCols = kCtrRows * kBlock / (kRadices * kRadices);  // 8
    const uint32_t col = tid & (kNextCols - 1u);
    // rank_slot(m * 256 + tid) = rank_slot((m & 3) * 256 + tid) + (m >> 2) * 1024: the XOR only
    // sees index bits 5..9 and only changes bits 0..4, so four bases serve all the reads.
    uint32_t read_base[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) read_base[q] = rank_slot<sizeof(K)>(q * kBlock + tid);
    uint32_t where[HAS_V ? NIT : 1];
    // FUSE: the width of the next digit and the constant 1 the counters are bumped by, each held in a
    // register (opaque to the compiler, which otherwise re-creates both from scalars for every key)
    uint32_t next_width = __builtin_popcount(next_mask), one = 1u;
    if (FUSE && sizeof(K) == 4) asm volatile("" : "+v"(next_width), "+v"(one));
    auto emit = [&](int m0, bool guarded, int restore) {
        K k[kGroup];
        uint4 w[kGroup];
#pragma unroll
        for (int j = 0; j < kGroup; ++j) k[j] = skeys[read_base[(m0 + j) & 3] + ((m0 + j) >> 2) * (4 * kBlock)];
        uint32_t d16[kGroup];  // 16 x digit: the byte offset into where_s and the high part of the counter slot
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            d16[j] = digit_of<K>(k[j], bit, mask) << 4;
            w[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(where_s) + d16[j]);
        }
        uint32_t second_bits = 0;  // bit j: key j of the group lands in its run's second destination tile
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            const uint32_t s = (m0 + j) * kBlock + tid;
            const uint32_t at = w[j].x + s;
            // ALIGNED: a slot holds a key of digit d only inside d's run of slots; what the empty
            // slots hold (stale words) never passes this test, whatever digit it decodes to
            const bool live = ALIGNED ? (s - w[j].z) < w[j].w : (!guarded || s < valid);
            if (HAS_V) where[m0 + j] = ALIGNED && !live ? 0xffffffffu : at;
            const K out_key = restore ? ordered_to_key<K>(k[j], restore) : k[j];  // the last pass restores the key's own bits
#if defined(RADX_SAFE_STORES)  // development builds: a wrong output base shows up as a parity failure, not as a memory fault
            if (live && at < n) {
#else
            if (live) {
#endif
                if (NT) __builtin_nontemporal_store(out_key, &kout[at]);
                else kout[at] = out_key;
            }
            if (FUSE) {
                const uint32_t next_digit = sizeof(K) == 4
                    ? __builtin_amdgcn_ubfe(static_cast<uint32_t>(k[j]), static_cast<uint32_t>(next_bit), next_width)
                    : digit_of<K>(k[j], next_bit, next_mask);
                const uint32_t slot = d16[j] | next_digit;
                const bool in_first = at <= w[j].y;
                if (live && in_first)
                    __hip_atomic_fetch_add(&ctr[slot * kNextCols + col], one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                second_bits |= (live && !in_first ? 1u : 0u) << j;
            }
        }
        // Keys for a run's second destination tile share one counter per slot.  A wave's keys are
        // consecutive in sorted order, so with skewed keys most of its lanes can want the same
        // counter: the lanes that agree with the first such lane are counted with one add, the
        // others add for themselves.  Rare for uniform keys (1 run in 16 crosses a tile boundary),
        // hence the wave-uniform test per group first.
        if (FUSE && __ballot(second_bits != 0u)) {
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                const bool in_second = (second_bits >> j) & 1u;
                const uint64_t second_lanes = __ballot(in_second);
                if (second_lanes) {
                    const uint32_t slot = (digit_of<K>(k[j], bit, mask) << 4) | digit_of<K>(k[j], next_bit, next_mask);
                    const int leader = __ffsll(static_cast<unsigned long long>(second_lanes)) - 1;
                    const uint32_t leader_slot = __builtin_amdgcn_readlane(slot, leader);
                    const uint64_t agree = __ballot(in_second && slot == leader_slot);
                    if (static_cast<int>(lane) == leader)
                        __hip_atomic_fetch_add(&next_second[leader_slot], static_cast<uint32_t>(__popcll(agree)),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else if (in_second && slot != leader_slot)
                        __hip_atomic_fetch_add(&next_second[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    };
    if (valid == TILE && order_out == 0) {
#pragma unroll
        for (int m = 0; m < NIT; m += kGroup) emit(m, false, 0);
    } else {
#pragma unroll
        for (int m = 0; m < NIT; m += kGroup) emit(m, true, order_out);
    }

    if (FUSE) {
        __syncthreads();
        // Thread t sums the columns of slot t = (digit t >> 4, next digit t & 15).
        const uint4 a = *reinterpret_cast<const uint4*>(&ctr[tid * kNextCols]);
        const uint4 b = *reinterpret_cast<const uint4*>(&ctr[tid * kNextCols + 4u]);
        const uint32_t first_sum = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        const uint32_t second_sum = next_second[tid];
        const uint32_t dest = where_s[tid >> 4].y >> LOG_TILE;
        uint32_t* row = next_counts + static_cast<size_t>(dest) * kRadices + (tid & 15u);
#if defined(RADX_SAFE_STORES)
        if (first_sum && dest < num_tiles) atomicAdd(row, first_sum);
        if (second_sum && dest + 1u < num_tiles) atomicAdd(row + kRadices, second_sum);
#else
        if (first_sum) atomicAdd(row, first_sum);
        if (second_sum) atomicAdd(row + kRadices, second_sum);
#endif
    }

    if (HAS_V) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) stage[rank_slot<4>(rank_of(i))] = val[i];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < NIT; ++m) {
            const uint32_t s = m * kBlock + tid;
#if defined(RADX_SAFE_STORES)
            if ((ALIGNED ? where[m] != 0xffffffffu : s < valid) && where[m] < n)
#else
            if (ALIGNED ? where[m] != 0xffffffffu : s < valid)
#endif
            {
                const uint32_t out_val = stage[rank_slot<4>((m & 3) * kBlock + tid) + (m >> 2) * (4 * kBlock)];
                if (NT) __builtin_nontemporal_store(out_val, &vout[where[m]]);
                else vout[where[m]] = out_val;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Inputs of at most one tile (4096 4-byte / 2048 8-byte keys): the whole pass loop of
// include/radx/radx_implement.inl:427-444 in ONE launch of one workgroup.  Nothing has to travel
// between workgroups, so no table leaves the CU: per pass the same thread-private packed
// counters, the same block-wide scan and the same rank-ordered LDS array as in scatter_kernel
// (its steps 2-4), after which thread t takes back the keys of ranks [t*KPT, (t+1)*KPT) and the
// next pass ranks those.  Keys past n are all-ones in the ordered image: they sort behind every
// real key (ties keep address order, and they come last) and are not stored.
// ---------------------------------------------------------------------------------------------
template <typename K, bool HAS_V>
__global__ __launch_bounds__(kBlock) void sort_tile_kernel(const K* __restrict__ kin, K* __restrict__ kout,
                                                            const uint32_t* __restrict__ vin, uint32_t* __restrict__ vout,
                                                            uint32_t n, int begin_bit, int end_bit, int order,
                                                            int padding_pass, uint32_t* __restrict__ digit_starts) {
    constexpr int KPT = KeyTraits<K>::kpt;
    constexpr uint32_t TILE = kBlock * KPT;
    __shared__ K skeys[TILE];
    __shared__ uint32_t svals[HAS_V ? TILE : 1];
    __shared__ __attribute__((aligned(16))) uint32_t ctr[kCtrRows * kBlock];
    __shared__ uint32_t wave_tot[kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    K key[KPT];
    uint32_t val[HAS_V ? KPT : 1];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t e = tid * KPT + i;
        key[i] = e < n ? key_to_ordered<K>(kin[e], order) : static_cast<K>(~static_cast<K>(0));
        if (HAS_V) val[i] = e < n ? vin[e] : 0u;
    }
    for (int bit = begin_bit; bit < end_bit; bit += kDigitBits) {
        const int width = end_bit - bit < kDigitBits ? end_bit - bit : kDigitBits;
        const uint32_t mask = (1u << width) - 1u;
#pragma unroll
        for (int r = 0; r < kCtrRows; ++r) ctr[r * kBlock + tid] = 0u;
        __syncthreads();  // also: every thread has taken the previous pass's keys back out of skeys
        uint32_t rank[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t d = digit_of<K>(key[i], bit, mask);
            const uint32_t sh = (d >> 3) << 4;
            const uint32_t old = __hip_atomic_fetch_add(&ctr[(d & 7u) * kBlock + tid], 1u << sh, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_WORKGROUP);
            rank[i] = (old >> sh) & 0xffffu;
        }
        __syncthreads();
        {   // block-wide exclusive scan, thread t owning words 8t..8t+7 of the row-major counter array
            const uint4 a = *reinterpret_cast<const uint4*>(&ctr[tid * 8u]);
            const uint4 b = *reinterpret_cast<const uint4*>(&ctr[tid * 8u + 4u]);
            const uint32_t c[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t total = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) total += c[i];
            const uint32_t incl = wave_inclusive_scan(total);
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
        // the 17 run starts of the last pass, as the multi-launch passes leave them: thread 0's
        // counter of digit d is where d's run begins (the pads all follow the last real key)
        if (bit + kDigitBits >= end_bit && !padding_pass && tid <= kRadices) {
            uint32_t start = n;
            if (tid < kRadices) start = min(start, (ctr[(tid & 7u) * kBlock] >> ((tid >> 3) << 4)) & 0xffffu);
            digit_starts[tid] = start;
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t d = digit_of<K>(key[i], bit, mask);
            const uint32_t sh = (d >> 3) << 4;
            rank[i] += (ctr[(d & 7u) * kBlock + tid] >> sh) & 0xffffu;  // below TILE: the counters add up to TILE
            skeys[rank_slot<sizeof(K)>(rank[i])] = key[i];
            if (HAS_V) svals[rank_slot<4>(rank[i])] = val[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            key[i] = skeys[rank_slot<sizeof(K)>(tid * KPT + i)];
            if (HAS_V) val[i] = svals[rank_slot<4>(tid * KPT + i)];
        }
    }
    // a padding pass (zero-width digit, a stable copy) moves nothing: only its run starts remain
    if (padding_pass && tid <= kRadices) digit_starts[tid] = tid == 0 ? 0u : n;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t e = tid * KPT + i;
        if (e < n) {
            kout[e] = ordered_to_key<K>(key[i], order);
            if (HAS_V) vout[e] = val[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// index path: the reference's indiction / permutation kernels (shaders-sdk/radix/indiction.comp:18-29,
// permutation.comp:18-29; built by Radix::initialize, never dispatched by Radix::command).  They
// exist so that wide payloads need not ride through every pass: sort (key, index) pairs, then
// gather the payload rows once through the sorted index.  Rows are elem_words units of T
// (T = 16-byte or 4-byte words); consecutive lanes copy consecutive words, so writes are
// coalesced and each row is read as one contiguous burst.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void indiction_kernel(uint32_t* __restrict__ ptrs, T* __restrict__ cache,
                                                            const T* __restrict__ in, uint64_t n,
                                                            uint32_t elem_words) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t first = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (uint64_t i = first; i < n; i += stride) ptrs[i] = static_cast<uint32_t>(i);     // ptrs[id] = id
    if (cache) {
        const uint64_t words = n * elem_words;
        for (uint64_t f = first; f < words; f += stride) cache[f] = in[f];               // keyExtra[id] = keyIn[id]
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void permutation_kernel(T* __restrict__ in, const T* __restrict__ cache,
                                                              const uint32_t* __restrict__ ptrs, uint64_t n,
                                                              uint32_t elem_words, int log2_words) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t words = n * elem_words;
    for (uint64_t f = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; f < words; f += stride) {
        const uint64_t i = log2_words >= 0 ? f >> log2_words : f / elem_words;
        const uint64_t w = f - i * elem_words;
        in[f] = cache[static_cast<uint64_t>(ptrs[i]) * elem_words + w];                   // keyIn[id] = keyExtra[ptrs[id]]
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

// get_blocks_info (shaders-sdk/radix/includes.glsl:162-173) with grid = ceil(n / tile): every
// workgroup's chunk is one tile.
struct Geometry {
    uint32_t tile;        // keys per tile (16 KiB of keys)
    uint32_t num_tiles;   // grid
    uint32_t num_groups;  // scan workgroups: ceil(num_tiles / 256)
};
constexpr Geometry geometry_for(uint64_t n, int key_bytes) {
    Geometry g{};
    g.tile = static_cast<uint32_t>(kTileDwords * 4 / key_bytes);
    const uint64_t nn = n ? n : 1;
    g.num_tiles = static_cast<uint32_t>((nn + g.tile - 1) / g.tile);
    g.num_groups = (g.num_tiles + kScanBlocks - 1) / kScanBlocks;
    return g;
}
constexpr size_t align256(size_t b) { return (b + 255u) & ~static_cast<size_t>(255u); }

// counts[0..1] alternate between "scanned this pass" and "accumulated for the next pass".  The
// scan-free mode (at most kScanlessMaxTiles tiles) rotates through three tables, so counts[2]
// only has that many rows, and keeps one small table of group sums and one [tile][16] table of
// published in-group prefixes beside each.
struct ScratchLayout {
    size_t counts[3], group_tot[3], prefix_words[3], offsets, group_sums, digit_starts, total;
};
constexpr size_t kGroupTotBytes = align256(2 * kGroupBaseWords * 4);  // group sums, then group bases
constexpr ScratchLayout layout_for(uint64_t max_n, int key_bytes) {
    const Geometry g = geometry_for(max_n, key_bytes);
    const size_t table = align256(static_cast<size_t>(g.num_tiles) * kRadices * 4);
    const uint32_t small_tiles = g.num_tiles < kScanlessMaxTiles ? g.num_tiles : kScanlessMaxTiles;
    ScratchLayout l{};
    size_t at = 0;
    l.counts[0] = at;    at += table;
    l.group_tot[0] = at; at += kGroupTotBytes;
    l.group_tot[1] = at; at += kGroupTotBytes;
    l.counts[1] = at;    at += table;
    l.group_tot[2] = at; at += kGroupTotBytes;
    l.counts[2] = at;    at += align256(static_cast<size_t>(small_tiles) * kRadices * 4);
    for (int i = 0; i < 3; ++i) { l.prefix_words[i] = at; at += align256(static_cast<size_t>(small_tiles) * kRadices * 4); }
    l.offsets = at;      at += table;
    l.group_sums = at;   at += align256(static_cast<size_t>(g.num_groups) * kRadices * 4);
    l.digit_starts = at; at += align256(32 * 4);
    l.total = at;
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
    bool fuse = true;
    bool scan_kernels = false;  // run the scan kernels even where the scan-free mode applies
    int publish = 1;            // tests: 0 scan-free workgroups get no bases and sum the table themselves, 2 no group sums
    int key_order = 0;      // kOrderSigned | kOrderFloat | kOrderDescending
    char* scratch = nullptr;
    ScratchLayout layout{};
    uint32_t* counts[3] = {nullptr, nullptr, nullptr};
    uint32_t* group_tot[3] = {nullptr, nullptr, nullptr};
    uint32_t* prefix_words[3] = {nullptr, nullptr, nullptr};
    bool tables_scanfree = false;  // the last pass ran scan-free: offsets holds full output bases
    uint32_t* offsets = nullptr;
    uint32_t* group_sums = nullptr;
    uint32_t* digit_starts = nullptr;
    int cur = 0;            // counts[cur] holds the histogram of the last pass run
    uint64_t tables_n = 0;  // element count of the pass whose tables the scratch holds (0: none yet)
    bool chained = false;   // counts[cur ^ 1] holds a fused histogram for the next pass
    uint64_t chained_n = 0;
    int chained_bit = 0, chained_bits = 0;
    bool profile = false;
    std::vector<Timed> pending;
    std::vector<Timed> pool;
    double ms[3] = {0, 0, 0};
    uint64_t launches[3] = {0, 0, 0};
};

namespace {

// Points the sorter's table pointers into `base`, laid out by layout_for.
void bind_scratch(radx_hip_sorter* s, char* base) {
    s->scratch = base;
    for (int i = 0; i < 3; ++i) {
        s->counts[i] = reinterpret_cast<uint32_t*>(base + s->layout.counts[i]);
        s->group_tot[i] = reinterpret_cast<uint32_t*>(base + s->layout.group_tot[i]);
        s->prefix_words[i] = reinterpret_cast<uint32_t*>(base + s->layout.prefix_words[i]);
    }
    s->offsets = reinterpret_cast<uint32_t*>(base + s->layout.offsets);
    s->group_sums = reinterpret_cast<uint32_t*>(base + s->layout.group_sums);
    s->digit_starts = reinterpret_cast<uint32_t*>(base + s->layout.digit_starts);
}

// Folds finished launch timings into the per-kind sums and recycles their events.  wait: block
// on launches still running; otherwise only launches whose stop event has fired are taken.
hipError_t harvest_timings(radx_hip_sorter* s, bool wait) {
    size_t kept = 0;
    hipError_t first_error = hipSuccess;
    for (auto& t : s->pending) {
        hipError_t e = wait ? hipEventSynchronize(t.stop) : hipEventQuery(t.stop);
        if (e == hipErrorNotReady) {
            // the query's own "not ready" must not surface as a later launch's error; nothing else is cleared
            if (hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();
            s->pending[kept++] = t;
            continue;
        }
        float f = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&f, t.start, t.stop);
        if (e == hipSuccess) { s->ms[t.kind] += f; s->launches[t.kind] += 1; }
        else if (first_error == hipSuccess) first_error = e;
        s->pool.push_back(t);
    }
    s->pending.resize(kept);
    return first_error;
}
constexpr size_t kMaxPendingTimings = 1024;  // ~40 sorts' worth; beyond it finished ones are folded in

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

struct PassArgs {
    const void* kin; void* kout; const void* vin; void* vout;
    uint32_t n;
    int bit, num_bits;            // this pass's digit
    bool have_counts;             // counts[cur ^ 1] already holds this pass's histogram (fused by the previous pass)
    bool fuse_next;               // accumulate the next pass's histogram while scattering
    int next_bit, next_num_bits;  // the next pass's digit
    int order_in, order_out;      // key-order mapping applied as keys are loaded / stored (first / last pass)
};

template <typename K, bool HAS_V>
int run_pass_typed(radx_hip_sorter* s, const PassArgs& a, hipStream_t stream) {
    const Geometry g = geometry_for(a.n, sizeof(K));
    const uint32_t mask = (1u << a.num_bits) - 1u;
    const uint32_t next_mask = (1u << a.next_num_bits) - 1u;
    const K* kin = static_cast<const K*>(a.kin);
    K* kout = static_cast<K*>(a.kout);
    const uint32_t* vin = static_cast<const uint32_t*>(a.vin);
    uint32_t* vout = static_cast<uint32_t*>(a.vout);

    // Small inputs run scan-free (see scatter_kernel): a pass is one launch, the tables rotate
    // through three buffers.  Otherwise: scan kernels, two buffers.
    const bool scanfree = s->fuse && !s->scan_kernels && g.num_tiles <= kScanlessMaxTiles;
    if (a.have_counts) {
        // the table the previous scatter filled becomes this pass's histogram
        s->cur = scanfree ? (s->cur + 1) % 3 : s->cur ^ 1;
    } else {
        s->cur = 0;
        LaunchTimer t(s, 0, stream);
        uint32_t* const none = nullptr;
        // scan-free: it also clears the table the first scatter adds into and the group sums of the first two passes
        hipLaunchKernelGGL(hist_kernel<K>, dim3(g.num_tiles), dim3(kBlock), 0, stream, kin, a.n, a.bit,
                           mask, s->counts[0], g.num_tiles, a.order_in, scanfree ? s->counts[1] : none,
                           scanfree ? s->group_tot[0] : none, scanfree ? s->group_tot[1] : none,
                           scanfree ? s->prefix_words[0] : none, scanfree ? s->prefix_words[1] : none);
    }
    uint32_t* counts = s->counts[s->cur];
    const int next = scanfree ? (s->cur + 1) % 3 : s->cur ^ 1;
    uint32_t* next_counts = s->counts[next];
    // Workgroups per CU are part of the tuning: every resident tile is 16 more write streams for
    // the memory system.  4-byte keys alone are fastest at the 5 the registers allow, and so are
    // 8-byte keys alone; with a payload 4 beat 5 (2^30 uint64 + payload: 77.9 -> 76.5 ms), which
    // the 4-byte-key variant reaches through its registers and the 8-byte-key variant through
    // LDS that is requested at launch and never touched.
    constexpr size_t hold_back = (HAS_V && sizeof(K) == 8) ? 6144 : 0;
    if (scanfree) {
        const int after = (s->cur + 2) % 3;
        const ScanFreeTables sf{s->group_tot[s->cur], s->prefix_words[s->cur], s->counts[after], s->group_tot[after],
                                s->prefix_words[after], s->digit_starts, s->publish};
        // the workgroups that publish the group sums come first, rounded up to the number of XCDs
        const uint32_t grid = g.num_tiles + ((group_count(g.num_tiles) + 1u + 7u) & ~7u);
        // what a pass writes stays in the L2s to the end of the kernel if it fits: nontemporal stores then
        const uint64_t pass_bytes = static_cast<uint64_t>(a.n) * (sizeof(K) + (HAS_V ? 4u : 0u));
        const bool nt = pass_bytes >= kNtStoreMinBytes && pass_bytes <= kNtStoreMaxBytes;
        LaunchTimer t(s, 2, stream);
        if (a.fuse_next && nt)
            hipLaunchKernelGGL((scatter_kernel<K, HAS_V, true, true, true>), dim3(grid), dim3(kBlock), hold_back, stream,
                               kin, kout, vin, vout, a.n, a.bit, mask, counts, s->offsets, s->group_sums, g.num_tiles,
                               a.next_bit, next_mask, next_counts, a.order_in, a.order_out, sf);
        else if (a.fuse_next)
            hipLaunchKernelGGL((scatter_kernel<K, HAS_V, true, true>), dim3(grid), dim3(kBlock), hold_back, stream,
                               kin, kout, vin, vout, a.n, a.bit, mask, counts, s->offsets, s->group_sums, g.num_tiles,
                               a.next_bit, next_mask, next_counts, a.order_in, a.order_out, sf);
        else if (nt)
            hipLaunchKernelGGL((scatter_kernel<K, HAS_V, false, true, true>), dim3(grid), dim3(kBlock), hold_back, stream,
                               kin, kout, vin, vout, a.n, a.bit, mask, counts, s->offsets, s->group_sums, g.num_tiles,
                               0, 0u, static_cast<uint32_t*>(nullptr), a.order_in, a.order_out, sf);
        else
            hipLaunchKernelGGL((scatter_kernel<K, HAS_V, false, true>), dim3(grid), dim3(kBlock), hold_back, stream,
                               kin, kout, vin, vout, a.n, a.bit, mask, counts, s->offsets, s->group_sums, g.num_tiles,
                               0, 0u, static_cast<uint32_t*>(nullptr), a.order_in, a.order_out, sf);
    } else {
        {
            LaunchTimer t(s, 1, stream);
            hipLaunchKernelGGL(scan_rows_kernel, dim3(g.num_groups), dim3(kBlock), 0, stream, counts, s->offsets,
                               s->group_sums, g.num_tiles, a.fuse_next ? next_counts : static_cast<uint32_t*>(nullptr),
                               a.n, s->digit_starts);
            if (g.num_groups > 1)
                hipLaunchKernelGGL(scan_groups_kernel, dim3(1), dim3(1024), 0, stream, s->group_sums, g.num_groups,
                                   a.n, s->digit_starts);
        }
        const ScanFreeTables sf{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        LaunchTimer t(s, 2, stream);
        if (a.fuse_next)
            hipLaunchKernelGGL((scatter_kernel<K, HAS_V, true, false>), dim3(g.num_tiles), dim3(kBlock), hold_back, stream,
                               kin, kout, vin, vout, a.n, a.bit, mask, counts, s->offsets, s->group_sums, g.num_tiles,
                               a.next_bit, next_mask, next_counts, a.order_in, a.order_out, sf);
        else
            hipLaunchKernelGGL((scatter_kernel<K, HAS_V, false, false>), dim3(g.num_tiles), dim3(kBlock), hold_back, stream,
                               kin, kout, vin, vout, a.n, a.bit, mask, counts, s->offsets, s->group_sums, g.num_tiles,
                               0, 0u, static_cast<uint32_t*>(nullptr), a.order_in, a.order_out, sf);
    }
    s->tables_scanfree = scanfree;
    RADX_TRY(hipGetLastError());
    s->tables_n = a.n;
    s->chained = a.fuse_next;
    s->chained_n = a.n;
    s->chained_bit = a.next_bit;
    s->chained_bits = a.next_num_bits;
    return 0;
}

int run_pass(radx_hip_sorter* s, const PassArgs& a, hipStream_t stream) {
    // A caller that profiles for a long time without reading the times must not grow the list
    // forever.  Folded in here, in front of the pass's launches, so that a launch error of this
    // pass is still there for the check behind them.
    if (s->pending.size() >= kMaxPendingTimings) (void)harvest_timings(s, s->pending.size() >= 4 * kMaxPendingTimings);
    const bool has_v = s->value_bytes != 0;
    if (s->key_bytes == 4)
        return has_v ? run_pass_typed<uint32_t, true>(s, a, stream) : run_pass_typed<uint32_t, false>(s, a, stream);
    return has_v ? run_pass_typed<uint64_t, true>(s, a, stream) : run_pass_typed<uint64_t, false>(s, a, stream);
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

int check_digit(const radx_hip_sorter* s, int bit, int num_bits) {
    if (num_bits < 0 || num_bits > kDigitBits) return bad_arg("num_bits must be 0..4");
    if (bit < 0 || bit + num_bits > s->key_bytes * 8) return bad_arg("digit outside the key");
    return 0;
}

// A pass over no elements launches nothing, but it is still "the last pass": every digit's run is
// empty, so the 17 run starts become zeros (in stream order) and no tables are left to read.
int empty_pass(radx_hip_sorter* s, hipStream_t stream) {
    RADX_TRY(hipMemsetAsync(s->digit_starts, 0, (kRadices + 1) * sizeof(uint32_t), stream));
    s->tables_n = 0;
    s->chained = false;
    return 0;
}

}  // namespace

extern "C" {

int radx_hip_version(void) { return 400; }

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
    bind_scratch(s, s->scratch);
    // tables read back before any pass has run are zeros, never uninitialised memory
    const hipError_t z = hipMemset(s->scratch, 0, s->layout.total);
    if (z != hipSuccess) { (void)hipFree(s->scratch); delete s; return fail(z, "hipMemset(scratch)"); }
    *out = s;
    return 0;
}

int radx_hip_destroy(radx_hip_sorter* s) {
    if (!s) return 0;
    DeviceGuard guard(s->device);
    (void)harvest_timings(s, true);
    for (auto& t : s->pool) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    const hipError_t e = s->scratch ? hipFree(s->scratch) : hipSuccess;
    delete s;
    if (e != hipSuccess) return fail(e, "hipFree(scratch)");
    return 0;
}

int radx_hip_tune(radx_hip_sorter* s, int flags) {
    if (!s) return bad_arg("sorter is NULL");
    if (flags < 0 || flags > 15 || (flags & 12) == 12) return bad_arg("flags must be 0..11");
    s->fuse = (flags & 1) != 0;
    s->scan_kernels = (flags & 2) != 0;
    s->publish = (flags & 4) ? 0 : (flags & 8) ? 2 : 1;
    s->chained = false;
    return 0;
}

int radx_hip_set_key_order(radx_hip_sorter* s, int key_kind, int descending) {
    if (!s) return bad_arg("sorter is NULL");
    if (key_kind < 0 || key_kind > 2) return bad_arg("key_kind must be 0 (unsigned), 1 (signed) or 2 (float)");
    s->key_order = (key_kind == 1 ? kOrderSigned : key_kind == 2 ? kOrderFloat : 0) | (descending ? kOrderDescending : 0);
    return 0;
}

int radx_hip_pass(radx_hip_sorter* s, const void* keys_in, void* keys_out, const void* vals_in,
                  void* vals_out, uint64_t n, int bit, int num_bits, void* stream) {
    return radx_hip_pass_fused(s, keys_in, keys_out, vals_in, vals_out, n, bit, num_bits, -1, 0, 0, stream);
}

int radx_hip_pass_fused(radx_hip_sorter* s, const void* keys_in, void* keys_out, const void* vals_in,
                        void* vals_out, uint64_t n, int bit, int num_bits, int next_bit, int next_num_bits,
                        int use_fused_counts, void* stream) {
    if (const int rc = check_buffers(s, keys_in, keys_out, vals_in, vals_out, n)) return rc;
    if (const int rc = check_digit(s, bit, num_bits)) return rc;
    const bool fuse_next = next_bit >= 0;
    if (fuse_next) if (const int rc = check_digit(s, next_bit, next_num_bits)) return rc;
    if (use_fused_counts && !(s->chained && s->chained_n == n && s->chained_bit == bit && s->chained_bits == num_bits))
        return bad_arg("no fused histogram for this digit and n: the previous pass did not produce one");
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    if (n == 0) return empty_pass(s, static_cast<hipStream_t>(stream));
    PassArgs a{keys_in, keys_out, vals_in, vals_out, static_cast<uint32_t>(n), bit, num_bits,
               use_fused_counts != 0, fuse_next, fuse_next ? next_bit : 0, fuse_next ? next_num_bits : 0, 0, 0};
    return run_pass(s, a, static_cast<hipStream_t>(stream));
}

namespace {
// The pass loop behind radx_hip_sort_bits / radx_hip_sort_bits_into.  keys_out == nullptr: the pass
// count is rounded up to even (a zero-width digit pass is a stable copy) so that the result ends
// in `keys`.  Otherwise exactly ceil(bits / 4) passes (at least one) run and the last one stores
// to keys_out / vals_out instead of its ping-pong partner.
int sort_bits_impl(radx_hip_sorter* s, void* keys, void* keys_swap, void* vals, void* vals_swap, uint64_t n,
                   int begin_bit, int end_bit, void* keys_out, void* vals_out, hipStream_t stream) {
    // At most one tile: all passes in one launch of one workgroup, the result straight to its
    // place (keys_swap is not needed).  Not in the modes that ask for the reference's pass
    // structure (histogram kernel every pass, or scan kernels).
    if (geometry_for(n, s->key_bytes).num_tiles == 1 && s->fuse && !s->scan_kernels) {
        {
            LaunchTimer t(s, 2, stream);
            const uint32_t n32 = static_cast<uint32_t>(n);
            // what the pass loop below would add: a zero-width pass to make the count even, or
            // the one pass a sort over no bits into keys_out still runs
            const int real_passes = (end_bit - begin_bit + kDigitBits - 1) / kDigitBits;
            const int padding_pass = keys_out ? real_passes == 0 : (real_passes & 1);
            const uint32_t* vin = static_cast<const uint32_t*>(vals);
            uint32_t* vout = static_cast<uint32_t*>(keys_out ? vals_out : vals);
            if (s->key_bytes == 4) {
                const uint32_t* kin = static_cast<const uint32_t*>(keys);
                uint32_t* kout = static_cast<uint32_t*>(keys_out ? keys_out : keys);
                if (s->value_bytes)
                    hipLaunchKernelGGL((sort_tile_kernel<uint32_t, true>), dim3(1), dim3(kBlock), 0, stream, kin, kout, vin, vout,
                                       n32, begin_bit, end_bit, s->key_order, padding_pass, s->digit_starts);
                else
                    hipLaunchKernelGGL((sort_tile_kernel<uint32_t, false>), dim3(1), dim3(kBlock), 0, stream, kin, kout, vin, vout,
                                       n32, begin_bit, end_bit, s->key_order, padding_pass, s->digit_starts);
            } else {
                const uint64_t* kin = static_cast<const uint64_t*>(keys);
                uint64_t* kout = static_cast<uint64_t*>(keys_out ? keys_out : keys);
                if (s->value_bytes)
                    hipLaunchKernelGGL((sort_tile_kernel<uint64_t, true>), dim3(1), dim3(kBlock), 0, stream, kin, kout, vin, vout,
                                       n32, begin_bit, end_bit, s->key_order, padding_pass, s->digit_starts);
                else
                    hipLaunchKernelGGL((sort_tile_kernel<uint64_t, false>), dim3(1), dim3(kBlock), 0, stream, kin, kout, vin, vout,
                                       n32, begin_bit, end_bit, s->key_order, padding_pass, s->digit_starts);
            }
        }
        RADX_TRY(hipGetLastError());
        s->tables_n = 0;  // no tables were made
        s->chained = false;
        return 0;
    }
    int passes = (end_bit - begin_bit + kDigitBits - 1) / kDigitBits;
    if (keys_out) passes = passes ? passes : 1;
    else passes += passes & 1;  // even, so the result ends in `keys` (scattering.comp:40,143)
    void* kbuf[2] = {keys, keys_swap};
    void* vbuf[2] = {vals, vals_swap};
    auto digit_at = [&](int p, int* bit, int* width) {
        int b = begin_bit + p * kDigitBits, w = end_bit - b;
        if (w > kDigitBits) w = kDigitBits;
        if (w <= 0) { w = 0; b = 0; }  // padding pass: zero-width digit, a stable copy
        *bit = b; *width = w;
    };
    for (int p = 0; p < passes; ++p) {
        PassArgs a{};
        digit_at(p, &a.bit, &a.num_bits);
        const int src = p & 1, dstb = 1 - src;
        const bool last = p == passes - 1;
        a.kin = kbuf[src]; a.vin = vbuf[src];
        a.kout = last && keys_out ? keys_out : kbuf[dstb];
        a.vout = last && keys_out ? vals_out : vbuf[dstb];
        a.n = static_cast<uint32_t>(n);
        a.have_counts = s->fuse && p > 0;
        a.fuse_next = s->fuse && !last;
        if (a.fuse_next) digit_at(p + 1, &a.next_bit, &a.next_num_bits);
        a.order_in = p == 0 ? s->key_order : 0;
        a.order_out = last ? s->key_order : 0;
        if (const int rc = run_pass(s, a, stream)) return rc;
    }
    s->chained = false;
    return 0;
}

bool ranges_overlap(const void* a, const void* b, uint64_t bytes) {
    const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
    return x < y + bytes && y < x + bytes;
}
}  // namespace

int radx_hip_sort_bits(radx_hip_sorter* s, void* keys, void* keys_swap, void* vals, void* vals_swap,
                       uint64_t n, int begin_bit, int end_bit, void* stream) {
    if (const int rc = check_buffers(s, keys, keys_swap, vals, vals_swap, n)) return rc;
    if (begin_bit < 0 || end_bit > s->key_bytes * 8 || begin_bit > end_bit)
        return bad_arg("need 0 <= begin_bit <= end_bit <= key bits");
    if (n == 0) return 0;
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    return sort_bits_impl(s, keys, keys_swap, vals, vals_swap, n, begin_bit, end_bit, nullptr, nullptr,
                          static_cast<hipStream_t>(stream));
}

int radx_hip_sort_bits_into(radx_hip_sorter* s, void* keys, void* keys_swap, void* vals, void* vals_swap,
                            uint64_t n, int begin_bit, int end_bit, void* keys_out, void* vals_out,
                            void* stream) {
    if (const int rc = check_buffers(s, keys, keys_swap, vals, vals_swap, n)) return rc;
    if (begin_bit < 0 || end_bit > s->key_bytes * 8 || begin_bit > end_bit)
        return bad_arg("need 0 <= begin_bit <= end_bit <= key bits");
    if (n == 0) return 0;
    if (!keys_out) return bad_arg("keys_out is NULL");
    if (reinterpret_cast<uintptr_t>(keys_out) % s->key_bytes) return bad_arg("keys_out is not aligned to the key size");
    if (s->value_bytes) {
        if (!vals_out) return bad_arg("vals_out is NULL for a key-value sorter");
        if (reinterpret_cast<uintptr_t>(vals_out) % 4) return bad_arg("vals_out is not 4-byte aligned");
    }
    int passes = (end_bit - begin_bit + kDigitBits - 1) / kDigitBits;
    if (passes == 0) passes = 1;
    // the last pass reads keys (odd pass count) or keys_swap (even): its output must not touch its input
    const void* last_src = (passes & 1) ? keys : keys_swap;
    if (ranges_overlap(keys_out, last_src, n * s->key_bytes))
        return bad_arg("keys_out overlaps the buffer the last pass reads");
    if (s->value_bytes && ranges_overlap(vals_out, (passes & 1) ? vals : vals_swap, n * 4))
        return bad_arg("vals_out overlaps the buffer the last pass reads");
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    return sort_bits_impl(s, keys, keys_swap, vals, vals_swap, n, begin_bit, end_bit, keys_out, vals_out,
                          static_cast<hipStream_t>(stream));
}

int radx_hip_msd_split(radx_hip_sorter* s, const void* keys_in, void* keys_out, const void* vals_in,
                       void* vals_out, uint64_t n, int log2_ranks, void* stream) {
    if (const int rc = check_buffers(s, keys_in, keys_out, vals_in, vals_out, n)) return rc;
    if (log2_ranks < 0 || log2_ranks > kDigitBits) return bad_arg("log2_ranks must be 0..4");
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    if (n == 0) return empty_pass(s, static_cast<hipStream_t>(stream));
    const int order = s->key_order | ((log2_ranks & 3) << kOrderRotShift);  // rotating by 0 or 4 is the identity
    PassArgs a{keys_in, keys_out, vals_in, vals_out, static_cast<uint32_t>(n), s->key_bytes * 8 - kDigitBits,
               kDigitBits, false, false, 0, 0, order, order};
    return run_pass(s, a, static_cast<hipStream_t>(stream));
}

int radx_hip_sort(radx_hip_sorter* s, void* keys, void* keys_swap, void* vals, void* vals_swap,
                  uint64_t n, void* stream) {
    if (!s) return bad_arg("sorter is NULL");
    return radx_hip_sort_bits(s, keys, keys_swap, vals, vals_swap, n, 0, s->key_bytes * 8, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// The sort for callers that cannot allocate (code under HIP graph capture, serving loops and
// harnesses that forbid hipMalloc): the same pass loop as radx_hip_sort with the tables in a
// device array that is part of the code object instead of a sorter's hipMalloc'ed scratch.
//
// Every GPU the library runs on holds its own copy of the array (sized for the largest input the
// entry point accepts) at its own address, from module load until the process ends.  In host code
// the array's name is only a handle for hipGetSymbolAddress — it is not a device address and must
// never be handed to a kernel or a memset.  The address is looked up once per device, with that
// device current, and every use goes through the looked-up pointer.
//
// All calls on one GPU share that GPU's array and the array keeps the last call's contents.
// Nothing in it has to be cleared by the caller: the histogram kernel stores every table row the
// scan later reads, the scan zeroes the accumulation table in front of every scatter, and it
// rewrites the run starts.  Two calls that overlap in time on one GPU race on the tables.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr uint64_t kNoAllocMaxN = RADX_HIP_NOALLOC_MAX_N;
constexpr size_t kNoAllocScratchBytes = layout_for(kNoAllocMaxN, 8).total;  // 8-byte keys have the smaller tile: more rows
static_assert(layout_for(kNoAllocMaxN, 4).total <= kNoAllocScratchBytes, "the 4-byte-key layout must fit as well");
static_assert(kNoAllocScratchBytes <= (11u << 20), "keep the resident array modest");
constexpr int kMaxDevices = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

namespace {
std::mutex g_scratch_mutex;
char* g_scratch_ptr[kMaxDevices] = {};  // per device: where that device's copy of `scratch` lives

// Address of `device`'s copy of the array; looked up on first use, with that device current.
int static_scratch_of(int device, char** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        RADX_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(hipErrorInvalidSymbol, "hipGetSymbolAddress(scratch) returned NULL");
        g_scratch_ptr[device] = static_cast<char*>(ptr);
    }
    *out = g_scratch_ptr[device];
    return 0;
}
}  // namespace

extern "C" {

int radx_hip_sort_noalloc(int device, void* keys, void* keys_swap, void* vals, void* vals_swap, uint64_t n,
                          int key_bytes, int value_bytes, void* stream) {
    if (key_bytes != 4 && key_bytes != 8) return bad_arg("key_bytes must be 4 or 8");
    if (value_bytes != 0 && value_bytes != 4) return bad_arg("value_bytes must be 0 or 4");
    if (device < 0 || device >= kMaxDevices) return bad_arg("device must be 0..63");
    if (n > kNoAllocMaxN) {
        snprintf(g_error, sizeof(g_error),
                 "invalid argument: radx_hip_sort_noalloc sorts at most %llu elements (its tables live in a fixed "
                 "%zu-byte device array); use a sorter from radx_hip_create for %llu",
                 static_cast<unsigned long long>(kNoAllocMaxN), kNoAllocScratchBytes, static_cast<unsigned long long>(n));
        return static_cast<int>(hipErrorInvalidValue);
    }
    radx_hip_sorter s;  // on the stack; with profiling off its (empty) vectors never allocate
    s.device = device;
    s.max_n = kNoAllocMaxN;
    s.key_bytes = key_bytes;
    s.value_bytes = value_bytes;
    if (const int rc = check_buffers(&s, keys, keys_swap, vals, vals_swap, n)) return rc;
    if (n == 0) return 0;
    // the tables used are `device`'s copy of the array: every buffer has to live on that GPU
    const void* bufs[4] = {keys, keys_swap, vals, vals_swap};
    for (const void* b : bufs) {
        if (!b) continue;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, b) != hipSuccess) {
            (void)hipGetLastError();
            return bad_arg("buffer is not memory HIP knows (hipMalloc / hipHostMalloc)");
        }
        if (attr.type == hipMemoryTypeDevice && attr.device != device) return bad_arg("buffer lives on another GPU than `device`");
    }
    DeviceGuard guard(device);
    RADX_TRY(guard.status());
    char* base = nullptr;
    if (const int rc = static_scratch_of(device, &base)) return rc;
    s.layout = layout_for(n, key_bytes);  // tables sized for this call: always within the array
    bind_scratch(&s, base);
    return sort_bits_impl(&s, keys, keys_swap, vals, vals_swap, n, 0, key_bytes * 8, nullptr, nullptr,
                          static_cast<hipStream_t>(stream));
}

int radx_hip_geometry(const radx_hip_sorter* s, uint64_t n, uint32_t* tile_keys, uint32_t* num_tiles) {
    if (!s) return bad_arg("sorter is NULL");
    if (n > kMaxElements) return bad_arg("n must be < 2^32");
    const Geometry g = geometry_for(n, s->key_bytes);
    if (tile_keys) *tile_keys = g.tile;
    if (num_tiles) *num_tiles = g.num_tiles;
    return 0;
}

int radx_hip_copy_tables(radx_hip_sorter* s, uint64_t n, uint32_t* counts, uint32_t* offsets, void* stream) {
    if (!s) return bad_arg("sorter is NULL");
    if (n == 0 || n > s->max_n) return bad_arg("n out of range");
    if (s->tables_n != n) return bad_arg("the scratch holds no tables for this n: the last pass ran on a different element count (or none ran)");
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Geometry g = geometry_for(n, s->key_bytes);
    const size_t words = static_cast<size_t>(g.num_tiles) * kRadices;
    RADX_TRY(hipStreamSynchronize(st));
    if (counts) RADX_TRY(hipMemcpy(counts, s->counts[s->cur], words * 4, hipMemcpyDeviceToHost));
    if (offsets) {
        std::vector<uint32_t> sums(static_cast<size_t>(g.num_groups) * kRadices);
        RADX_TRY(hipMemcpy(offsets, s->offsets, words * 4, hipMemcpyDeviceToHost));
        if (!s->tables_scanfree) {  // scan kernels leave the table in two parts
            RADX_TRY(hipMemcpy(sums.data(), s->group_sums, sums.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < words; ++i)
                offsets[i] += sums[(i % kRadices) * g.num_groups + i / kRadices / kScanBlocks];
        }
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

int radx_hip_digit_starts_async(radx_hip_sorter* s, void* dst, void* stream) {
    if (!s || !dst) return bad_arg("NULL argument");
    DeviceGuard guard(s->device);
    RADX_TRY(guard.status());
    RADX_TRY(hipMemcpyAsync(dst, s->digit_starts, (kRadices + 1) * sizeof(uint32_t), hipMemcpyDefault,
                            static_cast<hipStream_t>(stream)));
    return 0;
}

int radx_hip_device_tables(radx_hip_sorter* s, uint64_t n, const uint32_t** counts, const uint32_t** offsets,
                           const uint32_t** group_bases, uint32_t* tiles_per_group) {
    if (!s) return bad_arg("sorter is NULL");
    if (n == 0 || n > s->max_n) return bad_arg("n out of range");
    if (s->tables_n != n) return bad_arg("the scratch holds no tables for this n: the last pass ran on a different element count (or none ran)");
    if (counts) *counts = s->counts[s->cur];
    if (offsets) *offsets = s->offsets;
    if (group_bases) *group_bases = s->tables_scanfree ? nullptr : s->group_sums;
    if (tiles_per_group) *tiles_per_group = kScanBlocks;
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
    RADX_TRY(harvest_timings(s, true));
    for (int i = 0; i < 3; ++i) {
        if (ms) ms[i] = s->ms[i];
        if (launches) launches[i] = s->launches[i];
        s->ms[i] = 0;
        s->launches[i] = 0;
    }
    return 0;
}

namespace {
// The free functions below take raw device pointers and no sorter handle: each launches on the
// device that owns its buffer, whatever device is current on the calling thread.
int device_of(const void* p, int* device) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return bad_arg("buffer is not memory HIP knows (hipMalloc / hipHostMalloc)");
    }
    *device = attr.device;
    return 0;
}
#define RADX_ON_DEVICE_OF(ptr)                                     \
    int dev_ = 0;                                                  \
    if (const int rc_ = device_of((ptr), &dev_)) return rc_;       \
    DeviceGuard guard_(dev_);                                      \
    RADX_TRY(guard_.status())

int log2_or_minus1(uint32_t v) {
    if (v == 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1u << l) < v) ++l;
    return l;
}
}  // namespace

int radx_hip_indiction(uint32_t* ptrs, void* cache, const void* in, uint64_t n, size_t elem_bytes, void* stream) {
    if (n == 0) return 0;
    if (!ptrs) return bad_arg("ptrs is NULL");
    if (n > kMaxElements) return bad_arg("n must be < 2^32");
    if ((cache == nullptr) != (in == nullptr)) return bad_arg("cache and in go together");
    if (cache && (elem_bytes == 0 || elem_bytes % 4)) return bad_arg("elem_bytes must be a positive multiple of 4");
    RADX_ON_DEVICE_OF(ptrs);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool wide = cache && elem_bytes % 16 == 0 && aligned16(cache) && aligned16(in);
    if (wide)
        hipLaunchKernelGGL(indiction_kernel<uint4>, dim3(8192), dim3(kBlock), 0, st, ptrs, static_cast<uint4*>(cache),
                           static_cast<const uint4*>(in), n, static_cast<uint32_t>(elem_bytes / 16));
    else
        hipLaunchKernelGGL(indiction_kernel<uint32_t>, dim3(8192), dim3(kBlock), 0, st, ptrs,
                           static_cast<uint32_t*>(cache), static_cast<const uint32_t*>(in), n,
                           static_cast<uint32_t>(elem_bytes / 4));
    RADX_TRY(hipGetLastError());
    return 0;
}

int radx_hip_permutation(void* in, const void* cache, const uint32_t* ptrs, uint64_t n, size_t elem_bytes,
                         void* stream) {
    if (n == 0) return 0;
    if (!in || !cache || !ptrs) return bad_arg("NULL buffer");
    if (in == cache) return bad_arg("in and cache are the same buffer");
    if (n > kMaxElements) return bad_arg("n must be < 2^32");
    if (elem_bytes == 0 || elem_bytes % 4) return bad_arg("elem_bytes must be a positive multiple of 4");
    RADX_ON_DEVICE_OF(in);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (elem_bytes % 16 == 0 && aligned16(in) && aligned16(cache)) {
        const uint32_t words = static_cast<uint32_t>(elem_bytes / 16);
        hipLaunchKernelGGL(permutation_kernel<uint4>, dim3(8192), dim3(kBlock), 0, st, static_cast<uint4*>(in),
                           static_cast<const uint4*>(cache), ptrs, n, words, log2_or_minus1(words));
    } else {
        const uint32_t words = static_cast<uint32_t>(elem_bytes / 4);
        hipLaunchKernelGGL(permutation_kernel<uint32_t>, dim3(8192), dim3(kBlock), 0, st, static_cast<uint32_t*>(in),
                           static_cast<const uint32_t*>(cache), ptrs, n, words, log2_or_minus1(words));
    }
    RADX_TRY(hipGetLastError());
    return 0;
}

int radx_hip_generate_u32(void* keys, uint64_t n, uint32_t seed, void* stream) {
    if (n == 0) return 0;
    if (!keys) return bad_arg("keys is NULL");
    RADX_ON_DEVICE_OF(keys);
    hipLaunchKernelGGL(generate_u32_kernel, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint32_t*>(keys), n, seed);
    RADX_TRY(hipGetLastError());
    return 0;
}

int radx_hip_generate_u64(void* keys, void* vals, uint64_t n, uint32_t seed, void* stream) {
    if (n == 0) return 0;
    if (!keys) return bad_arg("keys is NULL");
    RADX_ON_DEVICE_OF(keys);
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
    RADX_ON_DEVICE_OF(keys);
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
    RADX_ON_DEVICE_OF(keys);
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
