// Development micro-benchmark: HBM copy rate by access width and block shape (4 GiB in, 4 GiB out).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t remap(uint32_t bid, uint32_t nwg) {
    const uint32_t q = nwg >> 3, r = nwg & 7u, x = bid & 7u, k = bid >> 3;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + k;
}
// one 16 KiB tile per workgroup, 4 x dwordx4 per thread in and out
template <bool REMAP> __global__ __launch_bounds__(256) void copy_x4(const uint4* in, uint4* out, uint32_t tiles) {
    const uint32_t t = REMAP ? remap(blockIdx.x, tiles) : blockIdx.x;
    const uint4* s = in + (size_t)t * 1024; uint4* d = out + (size_t)t * 1024;
    uint4 v[4];
    for (int m = 0; m < 4; ++m) v[m] = s[m * 256 + threadIdx.x];
    for (int m = 0; m < 4; ++m) d[m * 256 + threadIdx.x] = v[m];
}
// dwordx4 loads, dword stores (16 per thread), consecutive lanes -> consecutive dwords
template <bool REMAP> __global__ __launch_bounds__(256) void copy_x4_d1(const uint4* in, uint32_t* out, uint32_t tiles) {
    __shared__ uint4 lds[1024];
    const uint32_t t = REMAP ? remap(blockIdx.x, tiles) : blockIdx.x;
    const uint4* s = in + (size_t)t * 1024; uint32_t* d = out + (size_t)t * 4096;
    for (int m = 0; m < 4; ++m) lds[m * 256 + threadIdx.x] = s[m * 256 + threadIdx.x];
    __syncthreads();
    const uint32_t* l = (const uint32_t*)lds;
    for (int m = 0; m < 16; ++m) d[m * 256 + threadIdx.x] = l[m * 256 + threadIdx.x];
}
// dwordx4 loads, dword stores shifted by a misalignment and split in 16 runs of 256 keys whose
// destinations are 16 different regions (the scatter kernel's store shape for uniform digits)
__global__ __launch_bounds__(256) void copy_runs(const uint4* in, uint32_t* out, uint32_t tiles, uint32_t shift) {
    __shared__ uint4 lds[1024];
    const uint32_t t = remap(blockIdx.x, tiles);
    const uint4* s = in + (size_t)t * 1024;
    for (int m = 0; m < 4; ++m) lds[m * 256 + threadIdx.x] = s[m * 256 + threadIdx.x];
    __syncthreads();
    const uint32_t* l = (const uint32_t*)lds;
    const size_t region = (size_t)tiles * 256;  // elements per digit region
    for (int m = 0; m < 16; ++m) {
        size_t where = (size_t)m * region + (size_t)t * 256 + threadIdx.x + shift;
        if (where >= (size_t)tiles * 4096) where -= (size_t)tiles * 4096;
        out[where] = l[m * 256 + threadIdx.x];
    }
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load4(const uint4* p) { u32x4 v = __builtin_nontemporal_load((const u32x4*)p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_store4(uint4 q, uint4* p) { u32x4 v = {q.x, q.y, q.z, q.w}; __builtin_nontemporal_store(v, (u32x4*)p); }
// variants: nontemporal loads / stores, persistent grid-stride
template <bool NT_LD, bool NT_ST> __global__ __launch_bounds__(256) void copy_nt(const uint4* in, uint4* out, uint32_t tiles) {
    const uint32_t t = remap(blockIdx.x, tiles);
    const uint4* s = in + (size_t)t * 1024; uint4* d = out + (size_t)t * 1024;
    uint4 v[4];
    for (int m = 0; m < 4; ++m) v[m] = NT_LD ? nt_load4(&s[m * 256 + threadIdx.x]) : s[m * 256 + threadIdx.x];
    for (int m = 0; m < 4; ++m) { if (NT_ST) nt_store4(v[m], &d[m * 256 + threadIdx.x]); else d[m * 256 + threadIdx.x] = v[m]; }
}
__global__ __launch_bounds__(256) void copy_persist(const uint4* in, uint4* out, uint32_t tiles) {
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint4* s = in + (size_t)t * 1024; uint4* d = out + (size_t)t * 1024;
        uint4 v[4];
        for (int m = 0; m < 4; ++m) v[m] = s[m * 256 + threadIdx.x];
        for (int m = 0; m < 4; ++m) d[m * 256 + threadIdx.x] = v[m];
    }
}
template <int T> __global__ __launch_bounds__(T) void copy_big(const uint4* in, uint4* out, uint32_t blocks) {
    const uint32_t t = remap(blockIdx.x, blocks);
    const uint4* s = in + (size_t)t * T * 4; uint4* d = out + (size_t)t * T * 4;
    uint4 v[4];
    for (int m = 0; m < 4; ++m) v[m] = s[m * T + threadIdx.x];
    for (int m = 0; m < 4; ++m) d[m * T + threadIdx.x] = v[m];
}
template <bool NT_ST> __global__ __launch_bounds__(256) void copy_runs_nt(const uint4* in, uint32_t* out, uint32_t tiles, uint32_t shift) {
    __shared__ uint4 lds[1024];
    const uint32_t t = remap(blockIdx.x, tiles);
    const uint4* s = in + (size_t)t * 1024;
    for (int m = 0; m < 4; ++m) lds[m * 256 + threadIdx.x] = nt_load4(&s[m * 256 + threadIdx.x]);
    __syncthreads();
    const uint32_t* l = (const uint32_t*)lds;
    const size_t region = (size_t)tiles * 256;
    for (int m = 0; m < 16; ++m) {
        size_t where = (size_t)m * region + (size_t)t * 256 + threadIdx.x + shift;
        if (where >= (size_t)tiles * 4096) where -= (size_t)tiles * 4096;
        if (NT_ST) __builtin_nontemporal_store(l[m * 256 + threadIdx.x], &out[where]); else out[where] = l[m * 256 + threadIdx.x];
    }
}
// dwordx4 loads, LDS staging, dword stores, T threads per workgroup and 16 dwords per thread
template <int T> __global__ __launch_bounds__(T) void copy_staged(const uint4* in, uint32_t* out, uint32_t blocks) {
    __shared__ uint4 lds[T * 4];
    const uint32_t t = remap(blockIdx.x, blocks);
    const uint4* s = in + (size_t)t * T * 4; uint32_t* d = out + (size_t)t * T * 16;
    for (int m = 0; m < 4; ++m) lds[m * T + threadIdx.x] = nt_load4(&s[m * T + threadIdx.x]);
    __syncthreads();
    const uint32_t* l = (const uint32_t*)lds;
    for (int m = 0; m < 16; ++m) d[m * T + threadIdx.x] = l[m * T + threadIdx.x];
}
// two adjacent tiles per 256-thread workgroup, second tile's loads issued before the first is stored
__global__ __launch_bounds__(256) void copy_two(const uint4* in, uint32_t* out, uint32_t tiles) {
    __shared__ uint4 lds[1024];
    const uint32_t t = remap(blockIdx.x, tiles / 2) * 2;
    const uint4* s = in + (size_t)t * 1024; uint32_t* d = out + (size_t)t * 4096;
    uint4 a[4], b[4];
    for (int m = 0; m < 4; ++m) a[m] = nt_load4(&s[m * 256 + threadIdx.x]);
    for (int m = 0; m < 4; ++m) b[m] = nt_load4(&s[1024 + m * 256 + threadIdx.x]);
    const uint32_t* l = (const uint32_t*)lds;
    for (int m = 0; m < 4; ++m) lds[m * 256 + threadIdx.x] = a[m];
    __syncthreads();
    for (int m = 0; m < 16; ++m) d[m * 256 + threadIdx.x] = l[m * 256 + threadIdx.x];
    __syncthreads();
    for (int m = 0; m < 4; ++m) lds[m * 256 + threadIdx.x] = b[m];
    __syncthreads();
    for (int m = 0; m < 16; ++m) d[4096 + m * 256 + threadIdx.x] = l[m * 256 + threadIdx.x];
}
// The 16-run store shape again, but every wave instruction writes one 256-byte-aligned window of the
// destination: run r is laid out in a slot space of 320 slots at offset (destination start mod 64),
// so slot windows of 64 coincide with aligned destination windows; lanes outside the run are idle.
__global__ __launch_bounds__(256) void copy_runs_aligned(const uint4* in, uint32_t* out, uint32_t tiles, uint32_t shift) {
    __shared__ uint4 lds[1024];
    const uint32_t t = remap(blockIdx.x, tiles);
    const uint4* s = in + (size_t)t * 1024;
    for (int m = 0; m < 4; ++m) lds[m * 256 + threadIdx.x] = nt_load4(&s[m * 256 + threadIdx.x]);
    __syncthreads();
    const uint32_t* l = (const uint32_t*)lds;
    const size_t region = (size_t)tiles * 256;
    for (int it = 0; it < 20; ++it) {
        const uint32_t p = it * 256 + threadIdx.x, r = p / 320, q = p % 320;
        size_t start = (size_t)r * region + (size_t)t * 256 + shift + r * 7 + (size_t)r * 790528;  // every region its own misalignment; region starts not a power of two apart
        const uint32_t pad = (uint32_t)(start & 63);
        if (q >= pad && q < pad + 256) {
            size_t where = start - pad + q;
            if (where >= (size_t)tiles * 4096) where -= (size_t)tiles * 4096;
            out[where] = l[r * 256 + q - pad];
        }
    }
}
__global__ __launch_bounds__(256) void copy_runs_misaligned(const uint4* in, uint32_t* out, uint32_t tiles, uint32_t shift) {
    __shared__ uint4 lds[1024];
    const uint32_t t = remap(blockIdx.x, tiles);
    const uint4* s = in + (size_t)t * 1024;
    for (int m = 0; m < 4; ++m) lds[m * 256 + threadIdx.x] = nt_load4(&s[m * 256 + threadIdx.x]);
    __syncthreads();
    const uint32_t* l = (const uint32_t*)lds;
    const size_t region = (size_t)tiles * 256;
    for (int r = 0; r < 16; ++r) {
        size_t where = (size_t)r * region + (size_t)t * 256 + shift + r * 7 + (size_t)r * 790528 + threadIdx.x;
        if (where >= (size_t)tiles * 4096) where -= (size_t)tiles * 4096;
        out[where] = l[r * 256 + threadIdx.x];
    }
}
int main(int argc, char** argv) {
    const size_t n = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 30); const uint32_t tiles = n / 4096;
    uint32_t *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 5; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-28s %.3f ms  %.0f GB/s\n", name, best, 2.0 * n * 4 / best / 1e6);
    };
    run("x4->x4", [&] { copy_x4<false><<<tiles, 256>>>((uint4*)a, (uint4*)b, tiles); });
    run("x4->x4 xcd-remap", [&] { copy_x4<true><<<tiles, 256>>>((uint4*)a, (uint4*)b, tiles); });
    run("x4->dword", [&] { copy_x4_d1<false><<<tiles, 256>>>((uint4*)a, b, tiles); });
    run("x4->dword xcd-remap", [&] { copy_x4_d1<true><<<tiles, 256>>>((uint4*)a, b, tiles); });
    run("x4->16 runs aligned", [&] { copy_runs<<<tiles, 256>>>((uint4*)a, b, tiles, 0); });
    run("x4->16 runs shift 13", [&] { copy_runs<<<tiles, 256>>>((uint4*)a, b, tiles, 13); });
    run("nt load", [&] { copy_nt<true, false><<<tiles, 256>>>((uint4*)a, (uint4*)b, tiles); });
    run("nt store", [&] { copy_nt<false, true><<<tiles, 256>>>((uint4*)a, (uint4*)b, tiles); });
    run("nt load+store", [&] { copy_nt<true, true><<<tiles, 256>>>((uint4*)a, (uint4*)b, tiles); });
    run("persistent 2048 blocks", [&] { copy_persist<<<2048, 256>>>((uint4*)a, (uint4*)b, tiles); });
    run("persistent 1024 blocks", [&] { copy_persist<<<1024, 256>>>((uint4*)a, (uint4*)b, tiles); });
    run("512-thread blocks", [&] { copy_big<512><<<tiles / 2, 512>>>((uint4*)a, (uint4*)b, tiles / 2); });
    run("1024-thread blocks", [&] { copy_big<1024><<<tiles / 4, 1024>>>((uint4*)a, (uint4*)b, tiles / 4); });
    run("runs shift13 nt-load", [&] { copy_runs_nt<false><<<tiles, 256>>>((uint4*)a, b, tiles, 13); });
    run("runs shift13 nt-load+store", [&] { copy_runs_nt<true><<<tiles, 256>>>((uint4*)a, b, tiles, 13); });
    run("staged 256 threads", [&] { copy_staged<256><<<tiles, 256>>>((uint4*)a, b, tiles); });
    run("staged 512 threads", [&] { copy_staged<512><<<tiles / 2, 512>>>((uint4*)a, b, tiles / 2); });
    run("staged 1024 threads", [&] { copy_staged<1024><<<tiles / 4, 1024>>>((uint4*)a, b, tiles / 4); });
    run("two tiles per 256-thread wg", [&] { copy_two<<<tiles / 2, 256>>>((uint4*)a, b, tiles); });
    run("16 runs misaligned (r*7+13)", [&] { copy_runs_misaligned<<<tiles, 256>>>((uint4*)a, b, tiles, 13); });
    run("16 runs, aligned windows", [&] { copy_runs_aligned<<<tiles, 256>>>((uint4*)a, b, tiles, 13); });
    run("16 runs misaligned (r*7+0)", [&] { copy_runs_misaligned<<<tiles, 256>>>((uint4*)a, b, tiles, 0); });
    run("16 runs, aligned windows +0", [&] { copy_runs_aligned<<<tiles, 256>>>((uint4*)a, b, tiles, 0); });
    run("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
