// sort_test.cpp — the harness: this build's counterpart of the reference's only executable
// (reference: src/test/sort.cpp:246-483 rad::TestSort, src/test/sort.hpp:170-197).
//
// Same sequence through the same host API — PhysicalDeviceHelper, Device, Radix, Sort<Radix>,
// InputInterface, one device allocation holding keys and swap back to back, host-visible
// staging buffers, upload, timed sort, download, timed CPU sort — plus what the reference
// leaves out: seeded inputs and a bit-exact comparison of the two results.
//
//   sort_test [--n N | --log2n L] [--type u32|u64kv] [--dist uniform|shuffle|hash] [--seed S]
//             [--iters I] [--device D] [--batches B] [--cpu-only]
//
// --batches B sorts B independent host-resident arrays through pinned staging buffers with upload,
// sort and download overlapped on three streams, and reports the PCIe-inclusive rate.
//
// --cpu-only runs the plumbing without touching a GPU: generation, the reference's CPU
// comparator std::stable_sort(std::execution::par, ...) (src/test/sort.cpp:218-219) and the
// OpenMP parallel std::sort, checked against each other.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execution>
#include <iostream>
#include <numeric>
#include <parallel/algorithm>
#include <random>
#include <string>
#include <vector>

#include <omp.h>

#include <radx/radx.hpp>

namespace rad {

struct Options {
    size_t n = size_t(1) << 23;  // the reference's hard-coded size (src/test/sort.hpp:184)
    std::string type = "u32", dist = "shuffle";
    uint64_t seed = 1;
    int iters = 3, device = 0, batches = 0;
    bool cpuOnly = false;
};

// src/test/sort.cpp:226-234
static uint32_t hash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

static void generate(uint32_t* keys, size_t n, const Options& o) {
    std::mt19937_64 eng(o.seed);  // the reference seeds from std::random_device (sort.cpp:342-343)
    if (o.dist == "uniform") {    // the draw the reference leaves commented out (sort.cpp:344,349)
        std::uniform_int_distribution<uint32_t> distr;
        for (size_t i = 0; i < n; ++i) keys[i] = distr(eng);
    } else if (o.dist == "hash") {
        for (size_t i = 0; i < n; ++i) keys[i] = hash(uint32_t(i) ^ uint32_t(o.seed));
    } else {                      // what the reference runs: 0..n-1 shuffled (sort.cpp:349-350)
        std::iota(keys, keys + n, 0u);
        std::shuffle(keys, keys + n, eng);
    }
}

// src/test/sort.cpp:218-224
static void radixSortCPU(uint32_t* begin, uint32_t* end) { std::stable_sort(std::execution::par, begin, end); }
static void parallelSortCPU(uint32_t* begin, uint32_t* end) { __gnu_parallel::sort(begin, end); }

static const char* pstlBackend() {
#if defined(_PSTL_PAR_BACKEND_TBB)
    return "tbb";
#else
    return "serial";
#endif
}

template <class F>
static double bestOf(int reps, F&& f) {
    double best = 1e300;
    for (int r = 0; r < reps; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        f(r);
        auto t1 = std::chrono::steady_clock::now();
        best = std::min(best, std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
    return best;
}

static int runCpuOnly(const Options& o) {
    std::vector<uint32_t> keys(o.n), a, b;
    generate(keys.data(), o.n, o);
    const double msStable = bestOf(1, [&](int) { a = keys; radixSortCPU(a.data(), a.data() + a.size()); });
    const double msPar = bestOf(o.iters, [&](int) { b = keys; parallelSortCPU(b.data(), b.data() + b.size()); });
    const bool same = a == b && std::is_sorted(a.begin(), a.end());
    std::cout << "CPU sort measured in " << msStable << "ms (std::stable_sort(par), PSTL backend " << pstlBackend() << ")\n";
    std::cout << "CPU parallel sort measured in " << msPar << "ms (" << omp_get_max_threads() << " threads)\n";
    std::cout << "Sorting Finished: " << (same ? "PASS" : "FAIL") << std::endl;
    std::printf("{\"harness\":\"cpu-only\",\"n\":%zu,\"dist\":\"%s\",\"stable_ms\":%.3f,\"parallel_ms\":%.3f,"
                "\"threads\":%d,\"pstl\":\"%s\",\"pass\":%s}\n",
                o.n, o.dist.c_str(), msStable, msPar, omp_get_max_threads(), pstlBackend(), same ? "true" : "false");
    return same ? 0 : 1;
}

class TestSort {
protected:
    std::shared_ptr<radx::Device> device;
    std::shared_ptr<radx::Radix> program;
    std::shared_ptr<radx::Sort<radx::Radix>> radixSort;
    std::shared_ptr<radx::PhysicalDeviceHelper> physicalHelper;
    std::shared_ptr<radx::InputInterface> inputInterface;
    std::shared_ptr<radx::VmaAllocatedBuffer> vmaDeviceBuffer, vmaToHostBuffer, vmaHostBuffer;

public:
    int runU32(const Options& o) {
        const size_t elementCount = o.n;
        // reference: src/test/sort.cpp:304-308
        physicalHelper = std::make_shared<radx::PhysicalDeviceHelper>(radx::PhysicalDevice(o.device));
        device = std::make_shared<radx::Device>()->initialize(radx::LogicalDevice(o.device), physicalHelper);
        program = std::make_shared<radx::Radix>(); program->initialize(device);
        radixSort = std::make_shared<radx::Sort<radx::Radix>>(); radixSort->initialize(device, program, elementCount);
        inputInterface = std::make_shared<radx::InputInterface>(device);

        // reference: src/test/sort.cpp:310-330 — keys and swap back to back in one allocation
        // (each region padded to 16 bytes, the alignment the kernels' vector loads need)
        const size_t keysSize = (elementCount * sizeof(uint32_t) + 15) & ~size_t(15), keysBackupSize = keysSize;
        const size_t keysOffset = 0, keysBackupOffset = keysOffset + keysSize;
        const size_t memorySize = std::max<size_t>(keysBackupOffset + keysBackupSize, 16);
        vmaDeviceBuffer = std::make_shared<radx::VmaAllocatedBuffer>(device, memorySize, 0, radx::MEMORY_USAGE_GPU_ONLY);
        vmaHostBuffer = std::make_shared<radx::VmaAllocatedBuffer>(device, memorySize, 0, radx::MEMORY_USAGE_GPU_TO_CPU);
        vmaToHostBuffer = std::make_shared<radx::VmaAllocatedBuffer>(device, memorySize, 0, radx::MEMORY_USAGE_GPU_TO_CPU);
        radx::Vector<uint32_t>
            keysHostVector(vmaHostBuffer, elementCount, keysOffset),
            keysToHostVector(vmaToHostBuffer, elementCount, keysOffset),
            keysDeviceVector(vmaDeviceBuffer, elementCount, keysOffset),
            swapDeviceVector(vmaDeviceBuffer, elementCount, keysBackupOffset);

        // reference: src/test/sort.cpp:333-338
        inputInterface->setElementCount(keysDeviceVector.size());
        inputInterface->setKeysBufferInfo(keysDeviceVector);
        inputInterface->setSwapBufferInfo(swapDeviceVector);
        inputInterface->buildDescriptorSet();

        generate(keysHostVector.map(), elementCount, o);
        std::vector<uint32_t> sortedNumbers(elementCount);

        hipStream_t stream;
        radx::check_hip(hipStreamCreate(&stream), "hipStreamCreate");
        hipEvent_t e0, e1;
        radx::check_hip(hipEventCreate(&e0), "hipEventCreate");
        radx::check_hip(hipEventCreate(&e1), "hipEventCreate");
        double gpuMs = 1e300;
        for (int it = 0; it < o.iters; ++it) {
            // upload / sort (bracketed by two timestamps) / download: src/test/sort.cpp:396-423,429-438
            radx::check_hip(hipMemcpyAsync(keysDeviceVector.deviceData(), keysHostVector.data(), keysHostVector.range(),
                                           hipMemcpyHostToDevice, stream), "upload");
            radx::check_hip(hipEventRecord(e0, stream), "timestamp 0");
            radixSort->command(radx::CommandBuffer(stream), inputInterface);
            radx::check_hip(hipEventRecord(e1, stream), "timestamp 1");
            radx::check_hip(hipMemcpyAsync(keysToHostVector.data(), keysDeviceVector.deviceData(), keysDeviceVector.range(),
                                           hipMemcpyDeviceToHost, stream), "download");
            radx::check_hip(hipStreamSynchronize(stream), "submitCommandWithSync");
            float ms = 0.f;
            radx::check_hip(hipEventElapsedTime(&ms, e0, e1), "timestamps");
            gpuMs = std::min<double>(gpuMs, ms);
        }
        std::cout << "GPU sort measured in " << gpuMs << "ms" << std::endl;  // sort.cpp:449

        std::vector<uint32_t> ref;
        const double cpuMs = bestOf(1, [&](int) {
            ref.assign(keysHostVector.begin(), keysHostVector.end());
            radixSortCPU(ref.data(), ref.data() + ref.size());                 // sort.cpp:454-459
        });
        std::cout << "CPU sort measured in " << cpuMs << "ms (PSTL backend " << pstlBackend() << ")" << std::endl;
        std::vector<uint32_t> par;
        const double parMs = bestOf(2, [&](int) {
            par.assign(keysHostVector.begin(), keysHostVector.end());
            parallelSortCPU(par.data(), par.data() + par.size());
        });
        std::cout << "CPU parallel sort measured in " << parMs << "ms (" << omp_get_max_threads() << " threads)" << std::endl;

        if (elementCount) std::memcpy(sortedNumbers.data(), keysToHostVector.map(), keysToHostVector.range());  // sort.cpp:465
        const bool pass = sortedNumbers == ref && par == ref;
        std::cout << "Sorting Finished: " << (pass ? "PASS" : "FAIL") << std::endl;
        const double gk = elementCount / gpuMs / 1e6;
        std::printf("{\"harness\":\"gpu\",\"type\":\"u32\",\"n\":%zu,\"dist\":\"%s\",\"gpu_ms\":%.4f,\"gkeys_per_s\":%.3f,"
                    "\"hbm_GBps\":%.1f,\"roofline_frac\":%.4f,\"cpu_stable_ms\":%.3f,\"cpu_parallel_ms\":%.3f,"
                    "\"threads\":%d,\"pstl\":\"%s\",\"driver\":\"%s\",\"pass\":%s}\n",
                    elementCount, o.dist.c_str(), gpuMs, gk, gk * 64.0, gk * 64.0 / 8000.0, cpuMs, parMs,
                    omp_get_max_threads(), pstlBackend(), device->getDriverName().c_str(), pass ? "true" : "false");
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(stream);
        return pass ? 0 : 1;
    }

    // Staging for callers whose data starts and ends on the host (the upload / download command
    // buffers around the sort, src/test/sort.cpp:396-401,418-423): `batches` independent arrays in
    // pinned host memory, two device slots, three streams.  Batch i+1 uploads and batch i-1
    // downloads while batch i sorts; the reported rate includes PCIe both ways.
    int runStaged(const Options& o) {
        const size_t n = o.n;
        const int B = o.batches;
        physicalHelper = std::make_shared<radx::PhysicalDeviceHelper>(radx::PhysicalDevice(o.device));
        device = std::make_shared<radx::Device>()->initialize(radx::LogicalDevice(o.device), physicalHelper);
        program = std::make_shared<radx::Radix>(); program->initialize(device);
        radixSort = std::make_shared<radx::Sort<radx::Radix>>(); radixSort->initialize(device, program, n);

        const size_t keysSize = (n * sizeof(uint32_t) + 15) & ~size_t(15);
        std::shared_ptr<radx::VmaAllocatedBuffer> slotBuffer[2];
        std::shared_ptr<radx::InputInterface> slotInput[2];
        radx::Vector<uint32_t> slotKeys[2];
        for (int s = 0; s < 2; ++s) {
            slotBuffer[s] = std::make_shared<radx::VmaAllocatedBuffer>(device, 2 * keysSize, 0, radx::MEMORY_USAGE_GPU_ONLY);
            slotKeys[s] = radx::Vector<uint32_t>(slotBuffer[s], n, 0);
            radx::Vector<uint32_t> swap(slotBuffer[s], n, keysSize);
            slotInput[s] = std::make_shared<radx::InputInterface>(device);
            slotInput[s]->setElementCount(n).setKeysBufferInfo(slotKeys[s]).setSwapBufferInfo(swap).buildDescriptorSet();
        }
        auto hostIn = std::make_shared<radx::VmaAllocatedBuffer>(device, size_t(B) * keysSize, 0, radx::MEMORY_USAGE_CPU_TO_GPU);
        auto hostOut = std::make_shared<radx::VmaAllocatedBuffer>(device, size_t(B) * keysSize, 0, radx::MEMORY_USAGE_GPU_TO_CPU);
        std::vector<radx::Vector<uint32_t>> in, out;
        for (int b = 0; b < B; ++b) {
            in.emplace_back(hostIn, n, size_t(b) * keysSize);
            out.emplace_back(hostOut, n, size_t(b) * keysSize);
            Options ob = o; ob.seed = o.seed + b;
            generate(in[b].map(), n, ob);
        }

        hipStream_t up, run, down;
        radx::check_hip(hipStreamCreate(&up), "hipStreamCreate");
        radx::check_hip(hipStreamCreate(&run), "hipStreamCreate");
        radx::check_hip(hipStreamCreate(&down), "hipStreamCreate");
        hipEvent_t uploaded[2], sorted[2], downloaded[2];
        for (int s = 0; s < 2; ++s) {
            radx::check_hip(hipEventCreateWithFlags(&uploaded[s], hipEventDisableTiming), "hipEventCreate");
            radx::check_hip(hipEventCreateWithFlags(&sorted[s], hipEventDisableTiming), "hipEventCreate");
            radx::check_hip(hipEventCreateWithFlags(&downloaded[s], hipEventDisableTiming), "hipEventCreate");
        }
        auto pipeline = [&](bool overlap) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int b = 0; b < B; ++b) {
                const int s = b & 1;
                if (b >= 2) radx::check_hip(hipStreamWaitEvent(up, downloaded[s], 0), "slot free");
                radx::check_hip(hipMemcpyAsync(slotKeys[s].deviceData(), in[b].data(), n * sizeof(uint32_t),
                                               hipMemcpyHostToDevice, up), "upload");
                radx::check_hip(hipEventRecord(uploaded[s], up), "record");
                radx::check_hip(hipStreamWaitEvent(run, uploaded[s], 0), "wait upload");
                radixSort->command(radx::CommandBuffer(run), slotInput[s]);
                radx::check_hip(hipEventRecord(sorted[s], run), "record");
                radx::check_hip(hipStreamWaitEvent(down, sorted[s], 0), "wait sort");
                radx::check_hip(hipMemcpyAsync(out[b].data(), slotKeys[s].deviceData(), n * sizeof(uint32_t),
                                               hipMemcpyDeviceToHost, down), "download");
                radx::check_hip(hipEventRecord(downloaded[s], down), "record");
                if (!overlap) radx::check_hip(hipStreamSynchronize(down), "serialise");
            }
            radx::check_hip(hipStreamSynchronize(down), "drain");
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        };
        pipeline(true);  // warm-up (first touch of the pinned pages, code object load)
        const double serialMs = pipeline(false);
        const double overlapMs = pipeline(true);

        bool pass = true;
        std::vector<uint32_t> ref;
        for (int b = 0; b < B && pass; ++b) {
            ref.assign(in[b].begin(), in[b].end());
            parallelSortCPU(ref.data(), ref.data() + ref.size());
            pass = std::equal(ref.begin(), ref.end(), out[b].begin());
        }
        std::cout << "Staged sort of " << B << " batches: " << overlapMs << "ms overlapped, " << serialMs << "ms serialised" << std::endl;
        std::cout << "Sorting Finished: " << (pass ? "PASS" : "FAIL") << std::endl;
        const double keys = double(n) * B;
        std::printf("{\"harness\":\"staged\",\"type\":\"u32\",\"n\":%zu,\"batches\":%d,\"overlapped_ms\":%.3f,\"serialised_ms\":%.3f,"
                    "\"gkeys_per_s_incl_pcie\":%.3f,\"gkeys_per_s_serialised\":%.3f,\"pcie_GBps_each_way\":%.2f,\"pass\":%s}\n",
                    n, B, overlapMs, serialMs, keys / overlapMs / 1e6, keys / serialMs / 1e6,
                    keys * 4 / overlapMs / 1e6, pass ? "true" : "false");
        for (int s = 0; s < 2; ++s) { (void)hipEventDestroy(uploaded[s]); (void)hipEventDestroy(sorted[s]); (void)hipEventDestroy(downloaded[s]); }
        (void)hipStreamDestroy(up); (void)hipStreamDestroy(run); (void)hipStreamDestroy(down);
        return pass ? 0 : 1;
    }

    // uint64 keys + uint32 payload (= original index): the key-value path the reference declares
    // but never wires (shaders-sdk/radix/includes.glsl:128-135,151).
    int runU64KV(const Options& o) {
        const size_t n = o.n;
        physicalHelper = std::make_shared<radx::PhysicalDeviceHelper>(radx::PhysicalDevice(o.device));
        device = std::make_shared<radx::Device>()->initialize(radx::LogicalDevice(o.device), physicalHelper);
        program = std::make_shared<radx::Radix>(); program->setKeyBytes(8).setValueBytes(4); program->initialize(device);
        radixSort = std::make_shared<radx::Sort<radx::Radix>>(); radixSort->initialize(device, program, n);
        inputInterface = std::make_shared<radx::InputInterface>(device);

        const size_t kBytes = (n * 8 + 15) & ~size_t(15), vBytes = (n * 4 + 15) & ~size_t(15);
        auto devKeys = std::make_shared<radx::VmaAllocatedBuffer>(device, std::max<size_t>(2 * kBytes, 16), 0, radx::MEMORY_USAGE_GPU_ONLY);
        auto devVals = std::make_shared<radx::VmaAllocatedBuffer>(device, std::max<size_t>(2 * vBytes, 16), 0, radx::MEMORY_USAGE_GPU_ONLY);
        radx::Vector<uint64_t> keysDev(devKeys, n, 0), swapDev(devKeys, n, kBytes);
        radx::Vector<uint32_t> valsDev(devVals, n, 0), valsSwapDev(devVals, n, vBytes);
        inputInterface->setElementCount(n).setKeysBufferInfo(keysDev).setSwapBufferInfo(swapDev)
            .setValuesBufferInfo(valsDev).setValuesSwapBufferInfo(valsSwapDev).buildDescriptorSet();

        std::vector<uint64_t> keys(n), gotK(n);
        std::vector<uint32_t> vals(n), gotV(n);
        std::mt19937_64 eng(o.seed);
        for (size_t i = 0; i < n; ++i) {
            keys[i] = o.dist == "hash" ? (uint64_t(hash(uint32_t(i) ^ uint32_t(o.seed))) << 32) | hash(~uint32_t(i) ^ uint32_t(o.seed))
                                       : (o.dist == "uniform" ? eng() : eng() & 0xFFFFull);  // "shuffle": few distinct keys -> ties
            vals[i] = uint32_t(i);
        }
        hipStream_t stream;
        radx::check_hip(hipStreamCreate(&stream), "hipStreamCreate");
        hipEvent_t e0, e1;
        radx::check_hip(hipEventCreate(&e0), "hipEventCreate");
        radx::check_hip(hipEventCreate(&e1), "hipEventCreate");
        double gpuMs = 1e300;
        for (int it = 0; it < o.iters; ++it) {
            radx::check_hip(hipMemcpyAsync(keysDev.deviceData(), keys.data(), n * 8, hipMemcpyHostToDevice, stream), "upload keys");
            radx::check_hip(hipMemcpyAsync(valsDev.deviceData(), vals.data(), n * 4, hipMemcpyHostToDevice, stream), "upload vals");
            radx::check_hip(hipEventRecord(e0, stream), "timestamp 0");
            radixSort->command(radx::CommandBuffer(stream), inputInterface);
            radx::check_hip(hipEventRecord(e1, stream), "timestamp 1");
            radx::check_hip(hipMemcpyAsync(gotK.data(), keysDev.deviceData(), n * 8, hipMemcpyDeviceToHost, stream), "download keys");
            radx::check_hip(hipMemcpyAsync(gotV.data(), valsDev.deviceData(), n * 4, hipMemcpyDeviceToHost, stream), "download vals");
            radx::check_hip(hipStreamSynchronize(stream), "sync");
            float ms = 0.f;
            radx::check_hip(hipEventElapsedTime(&ms, e0, e1), "timestamps");
            gpuMs = std::min<double>(gpuMs, ms);
        }
        std::cout << "GPU sort measured in " << gpuMs << "ms" << std::endl;
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        const double cpuMs = bestOf(1, [&](int) {
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
        });
        std::cout << "CPU sort measured in " << cpuMs << "ms" << std::endl;
        bool pass = true;
        for (size_t i = 0; i < n && pass; ++i) pass = gotV[i] == order[i] && gotK[i] == keys[order[i]];
        std::cout << "Sorting Finished: " << (pass ? "PASS" : "FAIL") << std::endl;
        const double gp = n / gpuMs / 1e6;
        std::printf("{\"harness\":\"gpu\",\"type\":\"u64kv\",\"n\":%zu,\"dist\":\"%s\",\"gpu_ms\":%.4f,\"gpairs_per_s\":%.3f,"
                    "\"hbm_GBps\":%.1f,\"roofline_frac\":%.4f,\"cpu_stable_ms\":%.3f,\"pass\":%s}\n",
                    n, o.dist.c_str(), gpuMs, gp, gp * 384.0, gp * 384.0 / 8000.0, cpuMs, pass ? "true" : "false");
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(stream);
        return pass ? 0 : 1;
    }
};

}  // namespace rad

int main(int argc, char** argv) {
    rad::Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << a << " needs a value\n"; std::exit(2); } return argv[++i]; };
        if (a == "--n") o.n = std::stoull(next());
        else if (a == "--log2n") o.n = size_t(1) << std::stoi(next());
        else if (a == "--type") o.type = next();
        else if (a == "--dist") o.dist = next();
        else if (a == "--seed") o.seed = std::stoull(next());
        else if (a == "--iters") o.iters = std::max(1, std::stoi(next()));
        else if (a == "--device") o.device = std::stoi(next());
        else if (a == "--batches") o.batches = std::max(1, std::stoi(next()));
        else if (a == "--cpu-only") o.cpuOnly = true;
        else { std::cerr << "unknown option " << a << "\n"; return 2; }
    }
    if (o.type != "u32" && o.type != "u64kv") { std::cerr << "--type must be u32 or u64kv\n"; return 2; }
    if (o.dist != "uniform" && o.dist != "shuffle" && o.dist != "hash") { std::cerr << "--dist must be uniform, shuffle or hash\n"; return 2; }
    try {
        if (o.cpuOnly) return rad::runCpuOnly(o);
        rad::TestSort test;
        if (o.batches > 0 && o.type == "u32") return test.runStaged(o);
        return o.type == "u32" ? test.runU32(o) : test.runU64KV(o);
    } catch (const std::exception& e) {
        std::cerr << "sort_test: " << e.what() << std::endl;
        return 3;
    }
}
