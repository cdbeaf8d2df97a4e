/*
 * radx_hip.h — C ABI of the gfx950 LSD radix-sort engine.
 *
 * This is the boundary the header-only C++ layer in include/radx/ binds to, and the only
 * thing a replacement engine has to export.  Plain pointers and sizes; no C++ or torch
 * types.  Every function returns 0 on success or the failing hipError_t value
 * (argument errors are reported as hipErrorInvalidValue == 1); radx_hip_error_string()
 * gives the text of the last failure on the calling thread.
 *
 * What each entry point replaces in the reference (paths relative to the reference tree):
 *   radx_hip_create / radx_hip_scratch_bytes / radx_hip_destroy
 *       radx::Radix::createInternalMemory      include/radx/radx_implement.inl:450-490
 *       (one scratch allocation holding the histogram and prefix-scan tables) and
 *       radx::Radix::initialize                include/radx/radx_implement.inl:391-418
 *       (pipeline creation; here the kernels are part of the code object).
 *   radx_hip_sort
 *       radx::Radix::command                   include/radx/radx_implement.inl:421-447
 *       reached through radx::Sort<T>::command include/radx/radx_internal.hpp:129-132.
 *       Work is enqueued on the caller's stream exactly as the reference records into the
 *       caller's command buffer; nothing has executed when the call returns.
 *   radx_hip_pass / radx_hip_pass_fused / radx_hip_copy_tables
 *       one iteration of the loop at radx_implement.inl:427-444 (counting, partition,
 *       scattering: shaders-sdk/radix/{counting,partition,scattering}.comp) with the
 *       histogram (binding 3) and prefix-sum (binding 4) tables readable afterwards.
 *   radx_hip_indiction / radx_hip_permutation
 *       shaders-sdk/radix/indiction.comp:18-29, permutation.comp:18-29 (index + gather path).
 *   radx_hip_generate_* / radx_hip_check_sorted / radx_hip_checksum
 *       device-side counterparts of the harness's key generation and of the result
 *       comparison the harness omits     src/test/sort.cpp:226-234,342-350,452-469.
 *
 * Ownership: the caller owns keys / swap (and vals / vals_swap); each must hold at least
 * n elements and be 16-byte aligned.  The sorter owns only its scratch tables.  The sorted
 * result is left in `keys` (and `vals`); the swap buffers are clobbered
 * (shaders-sdk/radix/scattering.comp:40,143 ping-pong by pass parity, pass count even).
 * One sort may be in flight per sorter handle; use one handle per concurrent stream.
 */
#ifndef RADX_HIP_H
#define RADX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct radx_hip_sorter radx_hip_sorter;

/* Digit width of one pass and the number of bins, fixed at build time. */
#define RADX_HIP_BITS_PER_PASS 4
#define RADX_HIP_RADICES 16

/* Library version as major*10000 + minor*100 + patch. */
int radx_hip_version(void);

/* Text of the last error raised on this thread ("" when none). */
const char* radx_hip_error_string(void);

/* Bytes of device scratch a sorter for up to max_n elements of key_bytes (4 or 8) holds. */
size_t radx_hip_scratch_bytes(uint64_t max_n, int key_bytes);

/*
 * Create a sorter on HIP device `device` for at most max_n elements.
 * key_bytes: 4 (uint32 keys) or 8 (uint64 keys).  value_bytes: 0 (keys only) or 4
 * (uint32 payload that travels with its key).  max_n must be < 2^32: the reference's
 * element count is a 32-bit uniform (include/radx/radx_implement.inl:381-383).
 */
int radx_hip_create(int device, uint64_t max_n, int key_bytes, int value_bytes,
                    radx_hip_sorter** out);

int radx_hip_destroy(radx_hip_sorter* s);

/*
 * Enqueue a full ascending, unsigned, stable sort of n elements on `stream`
 * (a hipStream_t passed as void*; NULL is the default stream).
 * vals / vals_swap must be NULL for a keys-only sorter and non-NULL otherwise.
 * An input of at most one tile (radx_hip_geometry: 4096 4-byte or 2048 8-byte keys) is sorted by
 * one launch of one workgroup that keeps the keys in LDS through all passes — same digits, same
 * result, same 17 run starts (radx_hip_digit_starts), but no histogram / prefix-scan tables are
 * made: radx_hip_copy_tables and radx_hip_device_tables then return hipErrorInvalidValue.  The
 * same holds for radx_hip_sort_bits, radx_hip_sort_bits_into and radx_hip_sort_noalloc;
 * radx_hip_tune bit 1 (or a cleared bit 0) keeps the pass structure, tables included, at every size.
 */
int radx_hip_sort(radx_hip_sorter* s, void* keys, void* keys_swap, void* vals,
                  void* vals_swap, uint64_t n, void* stream);

/*
 * radx_hip_sort without a sorter handle and without any allocation.  It has no counterpart in
 * the reference (whose scratch is always a buffer allocated by Radix::createInternalMemory); it is
 * for callers that cannot call hipMalloc — code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  The histogram / prefix-scan tables live in a fixed device
 * array inside the library's code object (about 10 MiB, resident on every GPU the library is loaded
 * on until the process ends), so n is limited to RADX_HIP_NOALLOC_MAX_N elements; larger n fails
 * with hipErrorInvalidValue and a message.  Unsigned ascending order; key_bytes 4 or 8,
 * value_bytes 0 or 4, buffers as for radx_hip_sort.  The call makes no hipMalloc / hipFree and
 * enqueues the same kernels as radx_hip_sort on `stream`.  All calls on one GPU share that GPU's
 * array: calls that overlap in time on the same GPU (two streams, two threads) race on it, so
 * serialise them (one stream, or events between streams).  Nothing carries over between calls.
 * keys / keys_swap / vals / vals_swap must be memory of GPU `device`.  The first call on a GPU
 * looks up that GPU's copy of the array (which makes the runtime load the code object there): make
 * it outside stream capture; every later call only enqueues kernels and can be captured.
 */
#define RADX_HIP_NOALLOC_MAX_N (1ull << 26)
int radx_hip_sort_noalloc(int device, void* keys, void* keys_swap, void* vals, void* vals_swap,
                          uint64_t n, int key_bytes, int value_bytes, void* stream);

/*
 * As radx_hip_sort, restricted to key bits [begin_bit, end_bit).  The number of passes,
 * ceil((end_bit-begin_bit)/4), is rounded up to an even count so the result still lands
 * in `keys`; a padding pass sorts on a zero-width digit and is a stable copy.
 */
int radx_hip_sort_bits(radx_hip_sorter* s, void* keys, void* keys_swap, void* vals,
                       void* vals_swap, uint64_t n, int begin_bit, int end_bit, void* stream);

/*
 * As radx_hip_sort_bits, with exactly ceil((end_bit-begin_bit)/4) passes (at least one; no
 * padding pass) and the last pass storing to keys_out / vals_out instead of its ping-pong
 * partner.  keys / keys_swap (16-byte aligned) are both overwritten.  keys_out needs only the
 * alignment of one key and may lie anywhere that does not overlap the buffer the last pass
 * reads (keys for an odd pass count, keys_swap for an even one) — e.g. inside a larger array
 * at an arbitrary element offset.  The reference's passes always alternate between its two
 * fixed buffers (shaders-sdk/radix/scattering.comp:40,143); this is what a caller that sorts
 * several independent segments into one packed array needs (the multi-GPU path).
 */
int radx_hip_sort_bits_into(radx_hip_sorter* s, void* keys, void* keys_swap, void* vals,
                            void* vals_swap, uint64_t n, int begin_bit, int end_bit,
                            void* keys_out, void* vals_out, void* stream);

/*
 * The split step of a sort spread over 2^log2_ranks GPUs (no counterpart in the single-device
 * reference; it is one histogram -> scan -> scatter pass of the same kernels, as dispatched by
 * include/radx/radx_implement.inl:427-444).  Stable partition of n elements on the key's top
 * 4-bit digit d = (destination rank r << (4 - log2_ranks)) | sub-bucket j, written in
 * (j, r) order: bucket (j, r) is the run at position p = j * 2^log2_ranks + r, so that for
 * every j the runs for all destination ranks are adjacent, in rank order — one contiguous
 * all-to-all send buffer per sub-bucket.  radx_hip_digit_starts then returns the 16 run starts
 * indexed by p.  The sorter's key order (radx_hip_set_key_order) applies to the digit; the keys
 * are stored with their own bits.
 */
int radx_hip_msd_split(radx_hip_sorter* s, const void* keys_in, void* keys_out, const void* vals_in,
                       void* vals_out, uint64_t n, int log2_ranks, void* stream);

/*
 * Enqueue ONE pass (histogram -> scan -> scatter) reading keys_in / vals_in and writing
 * keys_out / vals_out, on the digit of `num_bits` (0..4) bits starting at bit `bit`.
 * The tables stay in the sorter's scratch until the next call.
 */
int radx_hip_pass(radx_hip_sorter* s, const void* keys_in, void* keys_out, const void* vals_in,
                  void* vals_out, uint64_t n, int bit, int num_bits, void* stream);

/*
 * One pass as radx_hip_sort chains them.  While scattering, the pass also accumulates the
 * histogram table of the NEXT pass (digit next_num_bits bits wide at next_bit; next_bit < 0
 * disables this), which is the table counting would produce on this pass's output.  With
 * use_fused_counts != 0 the pass takes its own histogram from what the previous
 * radx_hip_pass_fused call accumulated (same n, same digit) instead of running the histogram
 * kernel.  radx_hip_copy_tables shows the tables either way.
 */
int radx_hip_pass_fused(radx_hip_sorter* s, const void* keys_in, void* keys_out, const void* vals_in,
                        void* vals_out, uint64_t n, int bit, int num_bits, int next_bit,
                        int next_num_bits, int use_fused_counts, void* stream);

/*
 * Key order of radx_hip_sort / radx_hip_sort_bits on this sorter (default: unsigned ascending).
 * key_kind 0: unsigned integer keys; 1: two's-complement signed integers; 2: IEEE-754 floats
 * (binary32 for 4-byte keys, binary64 for 8-byte keys) in the total order -NaN < -inf < ... <
 * -0 < +0 < ... < +inf < +NaN.  descending != 0 reverses the order; ties still keep input order.
 * These are the options the reference declares as RadicePropStruct{Descending, IsSigned}
 * (shaders-sdk/radix/includes.glsl:137) and never reads.  The mapping is applied as the first
 * pass loads the keys and undone as the last pass stores them: same passes, same traffic.
 * radx_hip_pass / radx_hip_pass_fused always work on the raw unsigned bits.
 */
int radx_hip_set_key_order(radx_hip_sorter* s, int key_kind, int descending);

/*
 * Tuning knobs, as a bit set (default 1).  The sorted result, the run starts and the tables
 * radx_hip_copy_tables returns do not depend on any of them (with the default, a whole sort of at
 * most one tile leaves no tables at all, see radx_hip_sort).
 *   bit 0  fuse the histogram: every scatter pass of a sort also produces the next pass's
 *          histogram; clear, the histogram kernel runs every pass (three launches per pass, the
 *          reference's counting -> partition -> scattering, radx_implement.inl:427-444).
 *   bit 1  always run the scan kernels.  Clear (default), inputs of at most 16384 tiles sorted with
 *          bit 0 set run scan-free: the partition step is done by a few extra workgroups inside
 *          the scatter launch, one launch per pass; and a sort of at most one tile is one launch.
 *   bit 2 / bit 3 (tests; at most one of them)  scan-free passes publish nothing / no group sums,
 *          which makes every tile workgroup / the scanning workgroup sum the table itself — the
 *          bounded-wait fallbacks that normal runs never take.
 */
int radx_hip_tune(radx_hip_sorter* s, int flags);

/*
 * Geometry of the engine for n elements, i.e. get_blocks_info
 * (shaders-sdk/radix/includes.glsl:162-173) for a grid of ceil(n / tile) workgroups: keys per
 * tile (16 KiB of keys) and number of tiles.  Workgroup g owns elements
 * [g*tile, min((g+1)*tile, n)).
 */
int radx_hip_geometry(const radx_hip_sorter* s, uint64_t n, uint32_t* tile_keys, uint32_t* num_tiles);

/*
 * Synchronise `stream` and copy the tables of the last pass to host memory, both laid out
 * [num_tiles][RADX_HIP_RADICES] like the reference's bindings 3 and 4:
 *   counts[g][r]  = keys of tile g whose digit is r           (counting.comp:45)
 *   offsets[g][r] = first output index for digit r of tile g (partition.comp:26)
 * Either pointer may be NULL.
 */
int radx_hip_copy_tables(radx_hip_sorter* s, uint64_t n, uint32_t* counts, uint32_t* offsets,
                         void* stream);

/*
 * Synchronise `stream` and copy 17 uint32 to host: the start index of each digit's output run
 * of the last pass, followed by n.  out[d+1]-out[d] = number of keys with digit d.
 */
int radx_hip_digit_starts(radx_hip_sorter* s, uint32_t out[17], void* stream);

/*
 * The same 17 words, copied on `stream` without synchronising: dst is device memory (or pinned
 * host memory) the caller reads in stream order — e.g. the send buffer of the all-gather with
 * which the ranks of a multi-GPU sort tell each other their run lengths, so that the host waits
 * once per sort instead of once for its own lengths and once for everybody's.
 */
int radx_hip_digit_starts_async(radx_hip_sorter* s, void* dst, void* stream);

/*
 * Device-side views of the tables of the last pass over n elements — the counterparts of
 * InternalInterface::histogramBufferInfo / prefixScansBufferInfo (include/radx/radx_internal.hpp:25-28,
 * 37-38; bindings 3 and 4 of the reference's shaders), valid until the next call on this sorter:
 *   counts[g * 16 + r]   keys of tile g whose digit is r                         (counting.comp:45)
 *   offsets[g * 16 + r]  first output index for digit r of tile g (partition.comp:26) — complete
 *                        when *group_bases comes back NULL (scan-free passes); otherwise the
 *                        scan kernels leave it in two parts and the index is
 *                        offsets[g * 16 + r] + group_bases[r * ceil(tiles / T) + g / T], T = *tiles_per_group.
 * Any out-pointer may be NULL.  No synchronisation: read them in stream order after the pass.
 */
int radx_hip_device_tables(radx_hip_sorter* s, uint64_t n, const uint32_t** counts, const uint32_t** offsets,
                           const uint32_t** group_bases, uint32_t* tiles_per_group);

/*
 * Per-kernel timing.  When enabled, every launch made through this sorter is bracketed by
 * HIP events on the launch stream; radx_hip_kernel_times synchronises those events and
 * returns, per kernel kind (0 histogram, 1 scan, 2 scatter), the summed milliseconds and
 * the number of launches since the last reset.
 */
int radx_hip_profile(radx_hip_sorter* s, int enable);
int radx_hip_kernel_times(radx_hip_sorter* s, double ms[3], uint64_t launches[3]);

/*
 * The reference's index path (shaders-sdk/radix/indiction.comp:18-29, permutation.comp:18-29;
 * pipelines 3 and 4 of Radix::initialize, include/radx/radx_implement.inl:413-414, never
 * dispatched by Radix::command): sort (key, index) pairs, then move payload rows of any width once.
 *   indiction:   ptrs[i] = i;  cache[i] = in[i]        (cache and in may both be NULL: index only)
 *   permutation: in[i] = cache[ptrs[i]]
 * Rows are elem_bytes wide (a positive multiple of 4); 16-byte multiples on 16-byte aligned
 * buffers move as 16-byte words.  Enqueued on `stream`.
 */
int radx_hip_indiction(uint32_t* ptrs, void* cache, const void* in, uint64_t n, size_t elem_bytes,
                       void* stream);
int radx_hip_permutation(void* in, const void* cache, const uint32_t* ptrs, uint64_t n,
                         size_t elem_bytes, void* stream);

/* ---- harness utilities (device side of src/test/sort.cpp's generation and checking) ---- */

/* keys[i] = H(i ^ seed), H = the 32-bit integer hash of src/test/sort.cpp:226-234. */
int radx_hip_generate_u32(void* keys, uint64_t n, uint32_t seed, void* stream);
/* keys[i] = (uint64)H(i ^ seed) << 32 | H(~i ^ seed);  vals[i] = i (vals may be NULL). */
int radx_hip_generate_u64(void* keys, void* vals, uint64_t n, uint32_t seed, void* stream);

/*
 * Synchronous.  *violations = number of i in [1,n) with keys[i-1] > keys[i]
 * (for key_bytes 8 with vals != NULL also counts equal keys whose payloads decrease,
 * i.e. stability violations when payload = original index).
 */
int radx_hip_check_sorted(const void* keys, const void* vals, uint64_t n, int key_bytes,
                          uint64_t* violations, void* stream);

/* Synchronous.  out[0] = sum of keys mod 2^64, out[1] = xor of keys, out[2] = sum of H(key). */
int radx_hip_checksum(const void* keys, uint64_t n, int key_bytes, uint64_t out[3],
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RADX_HIP_H */
