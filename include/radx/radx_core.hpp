#pragma once
// radx_core.hpp — forward declarations and the HIP-backed stand-ins for the Vulkan handle types
// the reference's signatures mention (reference: include/radx/radx_core.hpp:15-35).
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>

#include <hip/hip_runtime_api.h>

#include "../radx_hip.h"

namespace radx {

    class Device;
    class PhysicalDeviceHelper;
    template <class T> class Sort;
    class Algorithm;
    class Radix;
    class VmaAllocatedBuffer;
    class InputInterface;
    class InternalInterface;
    using RadixSort = Sort<Radix>;  // reference: radx_core.hpp:33

    // The reference returns cvk::VkResult and always VK_SUCCESS (radx_internal.hpp:89-96);
    // here the value is 0 or a hipError_t.
    using Result = int;
    constexpr Result SUCCESS = 0;

    // vk::PhysicalDevice -> a HIP device ordinal that has not been made current yet.
    struct PhysicalDevice { int ordinal = 0; PhysicalDevice(int o = 0) : ordinal(o) {} };
    // vk::Device -> the HIP device ordinal work is issued to.
    struct LogicalDevice { int ordinal = 0; LogicalDevice(int o = 0) : ordinal(o) {} };
    // vk::CommandBuffer -> a stream: command() enqueues, the caller synchronises (record-then-submit).
    struct CommandBuffer { hipStream_t stream = nullptr; CommandBuffer(hipStream_t s = nullptr) : stream(s) {} };
    // vk::DescriptorBufferInfo{buffer, offset, range} -> device base pointer, byte offset, byte range.
    struct DescriptorBufferInfo {
        void* buffer = nullptr; std::size_t offset = 0; std::size_t range = 0;
        DescriptorBufferInfo() {}
        DescriptorBufferInfo(void* b, std::size_t o, std::size_t r) : buffer(b), offset(o), range(r) {}
        void* address() const { return buffer ? static_cast<char*>(buffer) + offset : nullptr; }
    };
    // vma::VmaMemoryUsage values the harness uses (src/test/sort.cpp:318-320).
    enum MemoryUsage { MEMORY_USAGE_GPU_ONLY = 1, MEMORY_USAGE_CPU_ONLY = 2, MEMORY_USAGE_CPU_TO_GPU = 3, MEMORY_USAGE_GPU_TO_CPU = 4 };

    // vulkan.hpp throws on failure; this layer throws for a failed HIP / C-ABI call.
    class Error : public std::runtime_error {
    public:
        Error(int code, const std::string& what) : std::runtime_error(what), code_(code) {}
        int code() const { return code_; }
    private:
        int code_;
    };
    inline void check_hip(hipError_t e, const char* what) {
        if (e != hipSuccess) throw Error(static_cast<int>(e), std::string(what) + ": " + hipGetErrorString(e));
    }
    inline void check_abi(int rc) {
        if (rc != 0) throw Error(rc, std::string("radx_hip: ") + radx_hip_error_string());
    }
}
