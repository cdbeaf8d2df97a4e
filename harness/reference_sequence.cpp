// reference_sequence.cpp — the reference harness's use of the host API, transcribed statement for
// statement (reference: src/test/sort.cpp:304-338 set-up, :411 the sort command) with only the
// handle types swapped: fw->getPhysicalDevice(0) -> radx::PhysicalDevice(0), fw->createDevice() ->
// radx::LogicalDevice(0), vk::BufferUsageFlagBits::... -> 0, vma::VMA_MEMORY_USAGE_* ->
// radx::MEMORY_USAGE_*, the vk::CommandBuffer -> radx::CommandBuffer(stream).  It exists to show
// that code written against the reference's headers compiles against these, including the
// overloads and public members the reference's own harness does not use: the rvalue
// std::shared_ptr<radx::Algorithm> form of Sort<T>::initialize (radx_internal.hpp:122-125), a class
// derived from InputInterface overriding its virtual setters (:65-68), and the histogram /
// prefix-scan views of InternalInterface (:25-28).  Run on a GPU it also checks the result and
// reads the two tables back through those views.
#include <algorithm>
#include <cstdio>
#include <memory>
#include <numeric>
#include <random>
#include <vector>

#include <radx/radx.hpp>

namespace rad {

// overrides a virtual setter the way a user of the reference could (radx_internal.hpp:65-68)
class CountingInput : public radx::InputInterface {
public:
    using radx::InputInterface::InputInterface;
    int keysSet = 0;
    radx::InputInterface& setKeysBufferInfo(const radx::DescriptorBufferInfo& keys = {}) override {
        ++keysSet;
        return radx::InputInterface::setKeysBufferInfo(keys);
    }
};

// a Sort that lets its owner look at the scratch views
class OpenSort : public radx::Sort<radx::Radix> {
public:
    radx::InternalInterface& internal() { return *internalInterface; }
};

class TestSort {
protected:
    std::shared_ptr<radx::Device> device;
    std::shared_ptr<radx::Radix> program;
    std::shared_ptr<OpenSort> radixSort;
    std::shared_ptr<radx::PhysicalDeviceHelper> physicalHelper;
    std::shared_ptr<radx::InputInterface> inputInterface;
    std::shared_ptr<radx::VmaAllocatedBuffer> vmaDeviceBuffer, vmaToHostBuffer, vmaHostBuffer;
    const size_t elementCount = (1 << 20) + 16;  // a multiple of 4 elements: swap starts 16-byte aligned

public:
    int run() {
        size_t keysSize, keysBackupSize, keysOffset, keysBackupOffset;

        // create radix sort application (RadX C++)                      src/test/sort.cpp:304-308
        physicalHelper = std::make_shared<radx::PhysicalDeviceHelper>(radx::PhysicalDevice(0));
        device = std::make_shared<radx::Device>()->initialize(radx::LogicalDevice(0), physicalHelper);
        program = std::make_shared<radx::Radix>(), program->initialize(device);
        // the "right-based link" overload                              radx_internal.hpp:122-125
        radixSort = std::make_shared<OpenSort>(), radixSort->initialize(device, std::shared_ptr<radx::Algorithm>(program), elementCount);
        auto counting = std::make_shared<CountingInput>(device);
        inputInterface = counting;

        { // sizes of offsets                                           src/test/sort.cpp:310-313
            keysSize = elementCount * sizeof(uint32_t), keysBackupSize = keysSize;
            keysOffset = 0, keysBackupOffset = keysOffset + keysSize;
        };

        // get memory size and set max element count                    src/test/sort.cpp:316-321
        size_t memorySize = keysBackupOffset + keysBackupSize;
        {
            vmaDeviceBuffer = std::make_shared<radx::VmaAllocatedBuffer>(this->device, memorySize, 0, radx::MEMORY_USAGE_GPU_ONLY);
            vmaHostBuffer = std::make_shared<radx::VmaAllocatedBuffer>(this->device, memorySize, 0, radx::MEMORY_USAGE_GPU_TO_CPU);
            vmaToHostBuffer = std::make_shared<radx::VmaAllocatedBuffer>(this->device, memorySize, 0, radx::MEMORY_USAGE_GPU_TO_CPU);
        };

        //                                                              src/test/sort.cpp:324-330
        radx::Vector<uint32_t>
            // in-host buffers
            keysHostVector = radx::Vector<uint32_t>(vmaHostBuffer, elementCount, keysOffset),
            keysToHostVector = radx::Vector<uint32_t>(vmaToHostBuffer, elementCount, keysOffset),
            keysDeviceVector = radx::Vector<uint32_t>(vmaDeviceBuffer, elementCount, keysOffset),
            swapDeviceVector = radx::Vector<uint32_t>(vmaDeviceBuffer, elementCount, keysBackupOffset)
            ;

        // on deprecation                                               src/test/sort.cpp:333-335
        inputInterface->setElementCount(keysDeviceVector.size());
        inputInterface->setKeysBufferInfo(keysDeviceVector);
        inputInterface->setSwapBufferInfo(swapDeviceVector);

        // build descriptor set                                         src/test/sort.cpp:338
        inputInterface->buildDescriptorSet();

        // random engine (seeded here; the reference seeds from std::random_device, :342-343)
        std::mt19937_64 eng(1);
        std::iota(keysHostVector.begin(), keysHostVector.end(), 0u);        // src/test/sort.cpp:349
        std::shuffle(keysHostVector.begin(), keysHostVector.end(), eng);    // src/test/sort.cpp:350

        hipStream_t stream;
        radx::check_hip(hipStreamCreate(&stream), "hipStreamCreate");
        radx::check_hip(hipMemcpyAsync(keysDeviceVector.deviceData(), keysHostVector.data(), keysSize, hipMemcpyHostToDevice, stream), "upload");
        radx::CommandBuffer sortCmdBuf(stream);

        // sorting command                                              src/test/sort.cpp:411
        radixSort->command(sortCmdBuf, inputInterface);

        radx::check_hip(hipMemcpyAsync(keysToHostVector.data(), keysDeviceVector.deviceData(), keysSize, hipMemcpyDeviceToHost, stream), "download");
        radx::check_hip(hipStreamSynchronize(stream), "sync");

        bool pass = counting->keysSet == 1;
        for (size_t i = 0; i < elementCount && pass; ++i) pass = keysToHostVector[i] == uint32_t(i);

        // the scratch views: the last pass's histogram sums to n, and its prefix table starts every
        // digit's tile-0 run where the digit's run starts (the table is complete at this size)
        radx::InternalInterface& internal = radixSort->internal();
        const size_t words = internal.histogramBufferInfo.range / sizeof(uint32_t);
        std::vector<uint32_t> counts(words), prefix(words);
        pass = pass && words > 0 && internal.prefixScansBufferInfo.range == internal.histogramBufferInfo.range &&
               internal.prefixGroupBasesBufferInfo.buffer == nullptr;
        if (pass) {
            radx::check_hip(hipMemcpy(counts.data(), internal.histogramBufferInfo.address(), words * 4, hipMemcpyDeviceToHost), "counts");
            radx::check_hip(hipMemcpy(prefix.data(), internal.prefixScansBufferInfo.address(), words * 4, hipMemcpyDeviceToHost), "prefix");
            uint64_t total = 0, digitTotal[16] = {0};
            for (size_t i = 0; i < words; ++i) { total += counts[i]; digitTotal[i % 16] += counts[i]; }
            pass = total == elementCount;
            uint64_t start = 0;
            for (int d = 0; d < 16 && pass; ++d) { pass = prefix[d] == start; start += digitTotal[d]; }
            // partitions[g][r] = partitions[g-1][r] + counts[g-1][r]       partition.comp:33-48
            for (size_t i = 16; i < words && pass; ++i) pass = prefix[i] == prefix[i - 16] + counts[i - 16];
        }
        radx::check_hip(hipStreamDestroy(stream), "hipStreamDestroy");
        pass = pass && twoPartTables();
        std::printf("Sorting Finished: %s\n", pass ? "PASS" : "FAIL");
        return pass ? 0 : 1;
    }

    // Above 16384 tiles the scan kernels leave the prefix table in two parts: the same check of
    // partitions[g][r] through prefixScansBufferInfo + prefixGroupBasesBufferInfo.
    bool twoPartTables() {
        const size_t n = size_t(16392) * 4096 + 5;   // 16393 tiles
        auto sorter = std::make_shared<OpenSort>();
        sorter->initialize(device, program, n);
        auto buffer = std::make_shared<radx::VmaAllocatedBuffer>(device, 2 * (n + 3) * sizeof(uint32_t), 0, radx::MEMORY_USAGE_GPU_ONLY);
        const size_t swapOffset = ((n * sizeof(uint32_t) + 15) / 16) * 16;
        radx::Vector<uint32_t> keys(buffer, n, 0), swap(buffer, n, swapOffset);
        radx::check_abi(radx_hip_generate_u32(keys.deviceData(), n, 11, nullptr));
        auto input = std::make_shared<radx::InputInterface>(device);
        input->setElementCount(n).setKeysBufferInfo(keys).setSwapBufferInfo(swap).buildDescriptorSet();
        sorter->command(radx::CommandBuffer(nullptr), input);
        radx::check_hip(hipDeviceSynchronize(), "sync");
        uint64_t bad = 1;
        radx::check_abi(radx_hip_check_sorted(keys.deviceData(), nullptr, n, 4, &bad, nullptr));
        radx::InternalInterface& internal = sorter->internal();
        const size_t words = internal.histogramBufferInfo.range / sizeof(uint32_t), tiles = words / 16;
        if (bad || tiles != 16393 || !internal.prefixGroupBasesBufferInfo.buffer || !internal.prefixTilesPerGroup) return false;
        const size_t groups = internal.prefixGroupBasesBufferInfo.range / (16 * sizeof(uint32_t));
        const size_t tilesPerGroup = internal.prefixTilesPerGroup;
        std::vector<uint32_t> counts(words), prefix(words), bases(groups * 16);
        radx::check_hip(hipMemcpy(counts.data(), internal.histogramBufferInfo.address(), words * 4, hipMemcpyDeviceToHost), "counts");
        radx::check_hip(hipMemcpy(prefix.data(), internal.prefixScansBufferInfo.address(), words * 4, hipMemcpyDeviceToHost), "prefix");
        radx::check_hip(hipMemcpy(bases.data(), internal.prefixGroupBasesBufferInfo.address(), bases.size() * 4, hipMemcpyDeviceToHost), "bases");
        // partitions[g][r] in (digit, tile) order is the running sum of counts      partition.comp:29-68
        uint64_t run = 0;
        for (int r = 0; r < 16; ++r)
            for (size_t g = 0; g < tiles; ++g) {
                const uint64_t have = uint64_t(prefix[g * 16 + r]) + bases[size_t(r) * groups + g / tilesPerGroup];
                if (have != run) return false;
                run += counts[g * 16 + r];
            }
        return run == n;
    }
};

}  // namespace rad

int main() {
    try {
        return rad::TestSort().run();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "reference_sequence: %s\n", e.what());
        return 2;
    }
}
