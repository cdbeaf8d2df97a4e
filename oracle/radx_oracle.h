/*
 * radx_oracle.h — CPU restatement of the reference's LSD radix-sort pass loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under radx_amd/, include/ or harness/'s GPU path may
 * call this; it exists so tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
 * check the HIP engine.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference holds no golden vectors, known-answer tests
 * or asserts for this path (its only executable, src/test/sort.cpp:429-469, sorts on GPU and
 * CPU and never compares), and its GLSL/Vulkan implementation cannot be built or run here.
 * The pin is definitional instead: the reference's scatter is stable
 * (shaders-sdk/radix/scattering.comp:119-123,141-143), so its result is the ascending unsigned
 * stable sort, and this restatement is cross-checked against std::stable_sort (oracle/cpu_sort.cpp,
 * the reference's own CPU comparator, src/test/sort.cpp:218-219) in tests/test_oracle.py.
 *
 * Each function cites the reference lines it follows (paths relative to the reference tree).
 */
#ifndef RADX_ORACLE_H
#define RADX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Launch geometry of the reference kernels.
 *   bits  BITS_PER_PASS                 shaders-sdk/radix/includes.glsl:21-39 (HEAD: 2, Turing: 8)
 *   grid  gl_NumWorkGroups.x            include/radx/radx_internal.hpp:143 (108)
 *   wave  Wave_Size                     shaders-sdk/include/ballotlib.glsl (64 on AMD)
 *   vec   VEC_SIZE, wave-rows per tile  shaders-sdk/radix/includes.glsl:53
 * block tile = wave*vec keys (includes.glsl:165).  The HIP engine's decomposition is the same
 * algorithm with grid = ceil(n / tile) and wave*vec = its tile (4096 uint32 / 2048 uint64).
 */
typedef struct {
    uint32_t bits;
    uint32_t grid;
    uint32_t wave;
    uint32_t vec;
} radx_ref_config;

typedef struct {
    uint32_t count;  /* tiles in this workgroup's chunk */
    uint32_t limit;  /* one past the last element      */
    uint32_t offset; /* first element                  */
} radx_ref_blocks;

/* get_blocks_info — shaders-sdk/radix/includes.glsl:162-173 (tiled(): include/radx/radx_utils.hpp:10-14) */
radx_ref_blocks radx_ref_blocks_info(const radx_ref_config* cfg, uint32_t n, uint32_t group);

/* extractKey / BFE — shaders-sdk/radix/includes.glsl:117-123,128-135 */
uint32_t radx_ref_digit(uint64_t key, uint32_t shift, uint32_t bits);

/* counting — shaders-sdk/radix/counting.comp:49-116.  counts is [grid][1<<bits]. */
void radx_ref_counting(const radx_ref_config* cfg, const void* keys, int key_bytes, uint32_t n,
                       uint32_t shift, uint32_t* counts);

/* partition — shaders-sdk/radix/partition.comp:29-68 with WG_COUNT == grid.  Both tables [grid][1<<bits]. */
void radx_ref_partition(const radx_ref_config* cfg, const uint32_t* counts, uint32_t* partitions);

/*
 * scattering — shaders-sdk/radix/scattering.comp:57-149.  vals_in/vals_out may be NULL; when
 * given, each uint32 payload is written to the same output index as its key (the declared,
 * unused valueIn binding: includes.glsl:151).
 */
void radx_ref_scattering(const radx_ref_config* cfg, const void* keys_in, void* keys_out,
                         const uint32_t* vals_in, uint32_t* vals_out, int key_bytes, uint32_t n,
                         uint32_t shift, const uint32_t* partitions);

/*
 * Radix::command — include/radx/radx_implement.inl:421-447: for Shift in [0, passes):
 * counting, partition, scattering, reading buffer Shift&1 and writing buffer 1-(Shift&1)
 * (scattering.comp:40,143).  passes == 0 means key bits / cfg->bits.  passes must be even for
 * the result to end in `keys`; the return value is the index (0 = keys, 1 = swap) that holds it.
 * counts_log / partitions_log, when not NULL, receive every pass's tables back to back
 * ([passes][grid][1<<bits]).
 */
int radx_ref_sort(const radx_ref_config* cfg, void* keys, void* swap, uint32_t* vals,
                  uint32_t* vals_swap, int key_bytes, uint32_t n, uint32_t passes,
                  uint32_t* counts_log, uint32_t* partitions_log);

/* indiction — shaders-sdk/radix/indiction.comp:18-29: ptrs[i] = i; cache[i] = in[i] (rows of elem_bytes) */
void radx_ref_indiction(uint32_t* ptrs, void* cache, const void* in, uint64_t n, size_t elem_bytes);
/* permutation — shaders-sdk/radix/permutation.comp:18-29: in[i] = cache[ptrs[i]] */
void radx_ref_permutation(void* in, const void* cache, const uint32_t* ptrs, uint64_t n, size_t elem_bytes);

/* hash — src/test/sort.cpp:226-234 */
uint32_t radx_ref_hash(uint32_t a);

/* Seeded generators shared with the HIP harness kernels (same formulas as include/radx_hip.h). */
void radx_ref_generate_u32(uint32_t* keys, uint64_t n, uint32_t seed);
void radx_ref_generate_u64(uint64_t* keys, uint32_t* vals, uint64_t n, uint32_t seed);

/* sum, xor, sum of mixed keys — same definition as radx_hip_checksum. */
void radx_ref_checksum(const void* keys, uint64_t n, int key_bytes, uint64_t out[3]);

/* Number of i in [1,n) with keys[i-1] > keys[i], or equal keys with vals[i-1] > vals[i]. */
uint64_t radx_ref_violations(const void* keys, const uint32_t* vals, uint64_t n, int key_bytes);

#ifdef __cplusplus
}
#endif
#endif
