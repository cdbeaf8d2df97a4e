// dispatch_bench.hip — how long does the GPU take to start and retire a grid of workgroups that do
// (almost) nothing, as a function of grid size and of what a workgroup occupies (threads, LDS,
// registers)?  The floor under every mid-size pass of the sort.
//   hipcc -O3 --offload-arch=gfx950 -o dispatch_bench dispatch_bench.hip && ./dispatch_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LDS_BYTES, int REGS>
__global__ __launch_bounds__(1024) void idle_kernel(unsigned* out, unsigned spin) {
    __shared__ unsigned lds[LDS_BYTES / 4];
    unsigned r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = threadIdx.x * (i + 1) + spin;
    lds[threadIdx.x] = r[0];
    __syncthreads();
    // keep the registers alive without doing real work
    unsigned acc = lds[(threadIdx.x + 1) % blockDim.x];
    if (spin == 0xffffffffu) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) acc += r[i] * r[(i + 7) % REGS];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

template <int LDS_BYTES, int REGS>
int sweep(const char* name, int threads, unsigned* out, hipStream_t st) {
    hipEvent_t a, b;
    OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    printf("%-44s", name);
    for (int grid : {8, 264, 1048, 2072, 4120, 8200, 16392}) {
        std::vector<float> t;
        for (int rep = 0; rep < 40; ++rep) {
            // a chain of 8 launches between two events: per-launch time in steady state, no idle gaps
            OK(hipEventRecord(a, st));
            for (int k = 0; k < 8; ++k) hipLaunchKernelGGL((idle_kernel<LDS_BYTES, REGS>), dim3(grid), dim3(threads), 0, st, out, 1u);
            OK(hipEventRecord(b, st));
            OK(hipEventSynchronize(b));
            float ms;
            OK(hipEventElapsedTime(&ms, a, b));
            if (rep >= 10) t.push_back(ms / 8 * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf(" %6d:%6.2f", grid, t[t.size() / 2]);
    }
    printf("   us per launch\n");
    return 0;
}

int main() {
    unsigned* out;
    OK(hipMalloc(&out, 1 << 20));
    hipStream_t st;
    OK(hipStreamCreate(&st));
    printf("workgroup shape                               grid: us per back-to-back launch (median)\n");
    if (sweep<1024, 8>("256 threads,  1 KiB LDS, few registers", 256, out, st)) return 1;
    if (sweep<26624, 8>("256 threads, 26 KiB LDS, few registers", 256, out, st)) return 1;
    if (sweep<26624, 80>("256 threads, 26 KiB LDS, ~96 registers", 256, out, st)) return 1;
    if (sweep<1024, 80>("256 threads,  1 KiB LDS, ~96 registers", 256, out, st)) return 1;
    if (sweep<53248, 80>("512 threads, 52 KiB LDS, ~96 registers", 512, out, st)) return 1;
    if (sweep<65536, 80>("1024 threads, 64 KiB LDS, ~96 registers", 1024, out, st)) return 1;
    if (sweep<1024, 8>("64 threads,  1 KiB LDS, few registers", 64, out, st)) return 1;
    return 0;
}
