// cpu_sort.cpp — the reference's CPU comparator, restated, plus the parallel std::sort baseline.
//
// TEST AND CPU-BASELINE INFRASTRUCTURE ONLY (see radx_oracle.h): used by the tests and by bench.py's
// cpu_baseline leg; never by the HIP engine or the harness (which carries its own CPU timing leg).
//
// The reference compares its GPU sort against
//     std::stable_sort(std::execution::par, begin, end)        src/test/sort.cpp:218-219
// on uint32_t.  libstdc++'s parallel policies need TBB; where its headers are absent the
// policy silently runs serially, so radx_cpu_pstl_backend() reports which one was compiled in
// and radx_cpu_parallel_sort_u32 (__gnu_parallel::sort over OpenMP) is the figure that really
// uses every host core.
#include <algorithm>
#include <cstdint>
#include <execution>
#include <numeric>
#include <parallel/algorithm>
#include <vector>

#include <omp.h>

extern "C" {

// "tbb" when the parallel policy is backed by TBB, "serial" when it degrades to one thread.
const char* radx_cpu_pstl_backend(void) {
#if defined(_PSTL_PAR_BACKEND_TBB)
    return "tbb";
#else
    return "serial";
#endif
}

int radx_cpu_threads(void) { return omp_get_max_threads(); }

// src/test/sort.cpp:218-219 (radixSortCPU), verbatim call on uint32_t.
void radx_cpu_stable_sort_u32(uint32_t* keys, uint64_t n) {
    std::stable_sort(std::execution::par, keys, keys + n);
}

void radx_cpu_stable_sort_u64(uint64_t* keys, uint64_t n) {
    std::stable_sort(std::execution::par, keys, keys + n);
}

// "parallel std::sort on the host cores": libstdc++ parallel mode, called explicitly.
void radx_cpu_parallel_sort_u32(uint32_t* keys, uint64_t n) {
    __gnu_parallel::sort(keys, keys + n);
}

void radx_cpu_parallel_sort_u64(uint64_t* keys, uint64_t n) {
    __gnu_parallel::sort(keys, keys + n);
}

// Stable sort of (key, payload) pairs by key only: ties keep input order, the semantics the
// reference's scatter gives a payload that travels with its key
// (shaders-sdk/radix/scattering.comp:119-123,141-143; includes.glsl:151).
void radx_cpu_stable_sort_kv_u64(uint64_t* keys, uint32_t* vals, uint64_t n) {
    struct Pair { uint64_t k; uint32_t v; };
    std::vector<Pair> p(n);
    for (uint64_t i = 0; i < n; ++i) p[i] = {keys[i], vals[i]};
    std::stable_sort(p.begin(), p.end(), [](const Pair& a, const Pair& b) { return a.k < b.k; });
    for (uint64_t i = 0; i < n; ++i) { keys[i] = p[i].k; vals[i] = p[i].v; }
}

void radx_cpu_stable_sort_kv_u32(uint32_t* keys, uint32_t* vals, uint64_t n) {
    struct Pair { uint32_t k; uint32_t v; };
    std::vector<Pair> p(n);
    for (uint64_t i = 0; i < n; ++i) p[i] = {keys[i], vals[i]};
    std::stable_sort(p.begin(), p.end(), [](const Pair& a, const Pair& b) { return a.k < b.k; });
    for (uint64_t i = 0; i < n; ++i) { keys[i] = p[i].k; vals[i] = p[i].v; }
}

}  // extern "C"
