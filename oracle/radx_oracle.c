/*
 * radx_oracle.c — CPU restatement of the reference's counting / partition / scattering pass
 * loop.  TEST INFRASTRUCTURE ONLY; see radx_oracle.h (including why parity is unpinned by the
 * reference and what it is pinned to instead).
 *
 * The GLSL runs one invocation per lane; here every wave-row is walked lane by lane and the
 * subgroup operations (ballot, exclusive bit count, exclusive add) are evaluated on 64-bit
 * lane masks, which is what they compute on a Wave_Size <= 64 device.
 */
#include "radx_oracle.h"

#include <stdlib.h>
#include <string.h>

#define MAX_WAVE 64u
#define MAX_RADICES 256u

/* tiled — include/radx/radx_utils.hpp:10-14: ceil division, 0 for 0. */
static uint64_t tiled(uint64_t sz, uint64_t tile) { return sz == 0 ? 0 : sz / tile + (sz % tile != 0); }

static uint64_t load_key(const void* keys, int key_bytes, uint64_t i) {
    return key_bytes == 8 ? ((const uint64_t*)keys)[i] : (uint64_t)((const uint32_t*)keys)[i];
}
static void store_key(void* keys, int key_bytes, uint64_t i, uint64_t k) {
    if (key_bytes == 8) ((uint64_t*)keys)[i] = k;
    else ((uint32_t*)keys)[i] = (uint32_t)k;
}
static unsigned popcount64(uint64_t m) { return (unsigned)__builtin_popcountll(m); }

/* shaders-sdk/radix/includes.glsl:162-173 */
radx_ref_blocks radx_ref_blocks_info(const radx_ref_config* cfg, uint32_t n, uint32_t group) {
    const uint64_t block_tile = (uint64_t)cfg->wave * cfg->vec;              /* :165 */
    const uint64_t per_work = tiled(n, cfg->grid);                           /* :166 */
    const uint64_t block_size = tiled(per_work, block_tile) * block_tile;    /* :167 */
    const uint64_t block_offset = block_size * group;                        /* :168 */
    const uint64_t block_limit = block_offset + block_size;                  /* :169 */
    radx_ref_blocks b;
    b.count = (uint32_t)tiled(block_size, block_tile);                       /* :170 */
    b.limit = (uint32_t)(block_limit < n ? block_limit : n);                 /* :172 min(limit, n) */
    b.offset = (uint32_t)(block_offset < n ? block_offset : n);
    if (b.count > 1048576u) b.count = 1048576u;                              /* counting.comp:56 */
    return b;
}

/* shaders-sdk/radix/includes.glsl:122 (32-bit) and :134 (64-bit BFE) */
uint32_t radx_ref_digit(uint64_t key, uint32_t shift, uint32_t bits) {
    if (bits == 0) return 0;
    return (uint32_t)((key >> (shift * bits)) & ((1ull << bits) - 1ull));
}

/*
 * One wave-row of a tile: the lanes that hold a valid element, their digits, and for every
 * radix the ballot of lanes holding it (counting.comp:65-96, scattering.comp:75-114).
 * Returns the number of valid lanes.
 */
static unsigned load_row(const radx_ref_config* cfg, const void* keys, int key_bytes, uint64_t first,
                         uint32_t limit, uint32_t shift, uint32_t digit[MAX_WAVE], uint64_t key[MAX_WAVE],
                         uint64_t mask[MAX_RADICES]) {
    const uint32_t radices = 1u << cfg->bits;
    unsigned valid = 0;
    memset(mask, 0, radices * sizeof(uint64_t));
    for (uint32_t lane = 0; lane < cfg->wave; ++lane) {
        const uint64_t addr = first + lane;
        if (addr >= limit) continue;                       /* predicate = lessThan(addrW, limit) */
        key[lane] = load_key(keys, key_bytes, addr);
        digit[lane] = radx_ref_digit(key[lane], shift, cfg->bits);
        mask[digit[lane]] |= 1ull << lane;                 /* sgrblt(true) among lanes with keyW == r */
        ++valid;
    }
    return valid;
}

/* shaders-sdk/radix/counting.comp:49-116 */
void radx_ref_counting(const radx_ref_config* cfg, const void* keys, int key_bytes, uint32_t n,
                       uint32_t shift, uint32_t* counts) {
    const uint32_t radices = 1u << cfg->bits;
    uint32_t digit[MAX_WAVE];
    uint64_t key[MAX_WAVE], mask[MAX_RADICES];
    for (uint32_t g = 0; g < cfg->grid; ++g) {
        uint32_t local_counts[MAX_RADICES];
        memset(local_counts, 0, sizeof(local_counts));                        /* :53-55 */
        const radx_ref_blocks blocks = radx_ref_blocks_info(cfg, n, g);       /* :56 */
        for (uint32_t wk = 0; wk < blocks.count; ++wk) {                      /* :64 */
            unsigned any = 0;
            for (uint32_t w = 0; w < cfg->vec; ++w) {
                const uint64_t first = (uint64_t)blocks.offset + ((uint64_t)wk * cfg->vec + w) * cfg->wave;
                const unsigned valid = load_row(cfg, keys, key_bytes, first, blocks.limit, shift, digit, key, mask);
                any |= valid;
                /* :93-101 — the first lane of each radix's ballot adds the ballot's bit count */
                for (uint32_t r = 0; r < radices; ++r) local_counts[r] += popcount64(mask[r]);
            }
            if (!any) break;                                                  /* :67 */
        }
        for (uint32_t r = 0; r < radices; ++r) counts[(size_t)g * radices + r] = local_counts[r];  /* :112-115 */
    }
}

/*
 * Exclusive add over one strip of up to `wave` lanes (subgroupExclusiveAdd) and the strip's
 * sum taken at the last active lane (partition.comp:42-43, 61-62).
 */
static uint32_t strip_scan(const uint32_t* histv, uint32_t lanes, uint32_t* scan) {
    uint32_t run = 0;
    for (uint32_t l = 0; l < lanes; ++l) { scan[l] = run; run += histv[l]; }
    return run;
}

/* shaders-sdk/radix/partition.comp:29-68, WG_COUNT taken equal to the dispatched grid */
void radx_ref_partition(const radx_ref_config* cfg, const uint32_t* counts, uint32_t* partitions) {
    const uint32_t radices = 1u << cfg->bits, wg_count = cfg->grid, wave = cfg->wave;
    uint32_t local_partitions[MAX_RADICES];
    uint32_t histv[MAX_WAVE], scan[MAX_WAVE];
    memset(local_partitions, 0, sizeof(local_partitions));                    /* :30 */

    /* :33-48 — per radix, strips of `wave` workgroups: cross-workgroup exclusive scan */
    for (uint32_t radice = 0; radice < radices; ++radice) {
        for (uint32_t gp = 0; gp < wg_count; gp += wave) {
            const uint32_t lanes = wg_count - gp < wave ? wg_count - gp : wave;
            for (uint32_t l = 0; l < lanes; ++l) histv[l] = counts[(size_t)(gp + l) * radices + radice];
            const uint32_t sum = strip_scan(histv, lanes, scan);
            const uint32_t pref = local_partitions[radice];                   /* :45 */
            local_partitions[radice] += sum;
            for (uint32_t l = 0; l < lanes; ++l)
                partitions[(size_t)(gp + l) * radices + radice] = pref + scan[l];  /* :46 */
        }
    }
    /* :51-67 — per workgroup, strips of `wave` radices: exclusive scan of the radix totals, added in */
    for (uint32_t workgroup = 0; workgroup < wg_count; ++workgroup) {
        uint32_t partsum = 0;                                                 /* :55 */
        for (uint32_t rk = 0; rk < radices; rk += wave) {
            const uint32_t lanes = radices - rk < wave ? radices - rk : wave;
            for (uint32_t l = 0; l < lanes; ++l) histv[l] = local_partitions[rk + l];
            const uint32_t sum = strip_scan(histv, lanes, scan);
            const uint32_t pref = partsum;                                    /* :64 */
            partsum += sum;
            for (uint32_t l = 0; l < lanes; ++l)
                partitions[(size_t)workgroup * radices + rk + l] += pref + scan[l];  /* :65 */
        }
    }
}

/* shaders-sdk/radix/scattering.comp:57-149 */
void radx_ref_scattering(const radx_ref_config* cfg, const void* keys_in, void* keys_out,
                         const uint32_t* vals_in, uint32_t* vals_out, int key_bytes, uint32_t n,
                         uint32_t shift, const uint32_t* partitions) {
    const uint32_t radices = 1u << cfg->bits;
    uint32_t digit[MAX_WAVE];
    uint64_t key[MAX_WAVE], mask[MAX_RADICES];
    for (uint32_t g = 0; g < cfg->grid; ++g) {
        uint32_t local_counts[MAX_RADICES], local_partitions[MAX_RADICES];
        for (uint32_t r = 0; r < radices; ++r) {                              /* :63-65 */
            local_partitions[r] = partitions[(size_t)g * radices + r];
            local_counts[r] = 0;
        }
        const radx_ref_blocks blocks = radx_ref_blocks_info(cfg, n, g);       /* :66 */
        for (uint32_t wk = 0; wk < blocks.count; ++wk) {                      /* :74 */
            unsigned any = 0;
            /* :119-123 — wave 0 walks the tile's wave-rows in address order */
            for (uint32_t w = 0; w < cfg->vec; ++w) {
                const uint64_t first = (uint64_t)blocks.offset + ((uint64_t)wk * cfg->vec + w) * cfg->wave;
                const unsigned valid = load_row(cfg, keys_in, key_bytes, first, blocks.limit, shift, digit, key, mask);
                any |= valid;
                uint32_t prtsum[MAX_WAVE];
                for (uint32_t lane = 0; lane < cfg->wave; ++lane)             /* :120 count = localCounts[keyW] */
                    if (first + lane < blocks.limit) prtsum[lane] = local_counts[digit[lane]];
                for (uint32_t r = 0; r < radices; ++r)                        /* :121 leader adds the ballot's bit count */
                    local_counts[r] += popcount64(mask[r]);
                for (uint32_t lane = 0; lane < cfg->wave; ++lane) {           /* :141-143 */
                    if (first + lane >= blocks.limit) continue;
                    const uint64_t below = mask[digit[lane]] & ((1ull << lane) - 1ull);
                    const uint32_t prtscn = popcount64(below);                /* subgroupBallotExclusiveBitCount */
                    const uint64_t where = (uint64_t)prtscn + prtsum[lane] + local_partitions[digit[lane]];
                    store_key(keys_out, key_bytes, where, key[lane]);
                    if (vals_in && vals_out) vals_out[where] = vals_in[first + lane];
                }
            }
            if (!any) break;                                                  /* :77 */
        }
    }
}

/* include/radx/radx_implement.inl:421-447 */
int radx_ref_sort(const radx_ref_config* cfg, void* keys, void* swap, uint32_t* vals,
                  uint32_t* vals_swap, int key_bytes, uint32_t n, uint32_t passes,
                  uint32_t* counts_log, uint32_t* partitions_log) {
    const uint32_t radices = 1u << cfg->bits;
    const size_t table = (size_t)cfg->grid * radices;
    if (passes == 0) passes = (uint32_t)(key_bytes * 8) / cfg->bits;          /* stageCount, :426 */
    uint32_t* counts = (uint32_t*)calloc(table ? table : 1, sizeof(uint32_t));
    uint32_t* parts = (uint32_t*)calloc(table ? table : 1, sizeof(uint32_t));
    void* kbuf[2] = {keys, swap};
    uint32_t* vbuf[2] = {vals, vals_swap};
    for (uint32_t shift = 0; shift < passes; ++shift) {                       /* :427 */
        const int in = (int)(shift & 1u), out = 1 - in;                       /* scattering.comp:40,143 */
        radx_ref_counting(cfg, kbuf[in], key_bytes, n, shift, counts);        /* :432-434 */
        radx_ref_partition(cfg, counts, parts);                               /* :436-438 */
        radx_ref_scattering(cfg, kbuf[in], kbuf[out], vbuf[in], vbuf[out], key_bytes, n, shift, parts);  /* :440-442 */
        if (counts_log) memcpy(counts_log + shift * table, counts, table * sizeof(uint32_t));
        if (partitions_log) memcpy(partitions_log + shift * table, parts, table * sizeof(uint32_t));
    }
    free(counts);
    free(parts);
    return (int)(passes & 1u);
}

/* shaders-sdk/radix/indiction.comp:18-29 */
void radx_ref_indiction(uint32_t* ptrs, void* cache, const void* in, uint64_t n, size_t elem_bytes) {
    for (uint64_t id = 0; id < n; ++id) {
        ptrs[id] = (uint32_t)id;                                              /* :27 ptrs[0].data[globalID] = globalID */
        if (cache && in) memcpy((char*)cache + id * elem_bytes, (const char*)in + id * elem_bytes, elem_bytes);
    }
}

/* shaders-sdk/radix/permutation.comp:18-29 */
void radx_ref_permutation(void* in, const void* cache, const uint32_t* ptrs, uint64_t n, size_t elem_bytes) {
    for (uint64_t id = 0; id < n; ++id)                                       /* :27 keyIn[id] = keyExtra[ptrs[id]] */
        memcpy((char*)in + id * elem_bytes, (const char*)cache + (uint64_t)ptrs[id] * elem_bytes, elem_bytes);
}

/* src/test/sort.cpp:226-234 */
uint32_t radx_ref_hash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

void radx_ref_generate_u32(uint32_t* keys, uint64_t n, uint32_t seed) {
    for (uint64_t i = 0; i < n; ++i) keys[i] = radx_ref_hash((uint32_t)i ^ seed);
}

void radx_ref_generate_u64(uint64_t* keys, uint32_t* vals, uint64_t n, uint32_t seed) {
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t lo = (uint32_t)i;
        keys[i] = ((uint64_t)radx_ref_hash(lo ^ seed) << 32) | radx_ref_hash(~lo ^ seed);
        if (vals) vals[i] = lo;
    }
}

void radx_ref_checksum(const void* keys, uint64_t n, int key_bytes, uint64_t out[3]) {
    uint64_t sum = 0, x = 0, mix = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t k = load_key(keys, key_bytes, i);
        sum += k;
        x ^= k;
        if (key_bytes == 8) {
            const uint32_t lo = radx_ref_hash((uint32_t)k);
            const uint32_t hi = radx_ref_hash((uint32_t)(k >> 32) ^ lo);
            mix += ((uint64_t)hi << 32) | lo;
        } else {
            mix += radx_ref_hash((uint32_t)k);
        }
    }
    out[0] = sum; out[1] = x; out[2] = mix;
}

uint64_t radx_ref_violations(const void* keys, const uint32_t* vals, uint64_t n, int key_bytes) {
    uint64_t bad = 0;
    for (uint64_t i = 1; i < n; ++i) {
        const uint64_t a = load_key(keys, key_bytes, i - 1), b = load_key(keys, key_bytes, i);
        if (a > b) ++bad;
        else if (vals && a == b && vals[i - 1] > vals[i]) ++bad;
    }
    return bad;
}
