// Development comparator: the vendor library's device radix sort (rocPRIM: 8-bit digits, one
// sweep per digit with decoupled look-back) on the same GPU and key generator, as a data point
// for what a look-back pass costs here.  Not part of the product and not linked into it.
//   ./rocprim_sort_bench [log2n=30] [iters=5]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__host__ __device__ inline uint32_t mix(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12); a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);  a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);  a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
__global__ void fill(uint32_t* k, uint64_t n, uint32_t seed) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        k[i] = mix((uint32_t)i ^ seed);
}
__global__ void check(const uint32_t* k, uint64_t n, unsigned long long* bad) {
    for (uint64_t i = 1 + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (k[i - 1] > k[i]) atomicAdd(bad, 1ull);
}

int main(int argc, char** argv) {
    const int log2n = argc > 1 ? atoi(argv[1]) : 30, iters = argc > 2 ? atoi(argv[2]) : 5;
    const size_t n = size_t(1) << log2n;
    uint32_t *in, *out; unsigned long long* bad; void* tmp = nullptr; size_t tmp_bytes = 0;
    CHECK(hipMalloc(&in, n * 4)); CHECK(hipMalloc(&out, n * 4)); CHECK(hipMalloc(&bad, 8));
    CHECK(rocprim::radix_sort_keys(nullptr, tmp_bytes, in, out, n));
    CHECK(hipMalloc(&tmp, tmp_bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int it = 0; it < iters; ++it) {
        fill<<<4096, 256>>>(in, n, it + 1);
        CHECK(hipEventRecord(e0));
        CHECK(rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, n));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemset(bad, 0, 8));
        check<<<4096, 256>>>(out, n, bad);
        unsigned long long h = 0; CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("rocprim radix_sort_keys n=2^%d iter %d: %.3f ms  %.2f Gkeys/s  (temp %.1f MiB, violations %llu)\n",
               log2n, it, ms, n / ms / 1e6, tmp_bytes / 1048576.0, h);
    }
    return 0;
}
